import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
n = int(sys.argv[1])
torch.set_num_threads(n)
from oracle import vid2seq_ref as R
from vidchapters_amd import synth
t0 = time.time()
a = torch.randn(1000, 768); w = torch.randn(3072, 768)
for _ in range(20): (a @ w.T).relu() @ w
print(f"threads={n} matmul chain {time.time() - t0:.2f}s", flush=True)
cfg = R.RefConfig()
P = {k: torch.randn(*s) * 0.02 for k, s in R.param_shapes(cfg).items()}
b = synth.make_batch(1, 100, 1000, 256, cfg.vocab, 99, 768)
t0 = time.time()
with torch.no_grad():
    out, _ = R.vid2seq_forward(P, cfg, b["video"], b["input_ids"], b["input_ids"] != 0, b["output_ids"], b["output_ids"] != 0)
print(f"threads={n} forward B=1: {time.time() - t0:.2f}s", flush=True)
for v in P.values(): v.requires_grad_(True)
t0 = time.time()
rec = R.train_step(P, {}, cfg, b, denoising=0.0)
print(f"threads={n} train_step B=1: {time.time() - t0:.2f}s", flush=True)
