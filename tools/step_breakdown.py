"""Per-shape GEMM/attention timing inside a real training step (HIP events around each launch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import SyntheticTokenizer, Vid2Seq, synth, lib as L
from vidchapters_amd.train import Trainer
dev = torch.device("cuda")
tok = SyntheticTokenizer(32100, 100)
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
model = Vid2Seq("t5-base", tokenizer=tok, vis_drop=p, enc_drop=p, dec_drop=p, init_seed=1234, device=dev).train()
tr = Trainer(model, denoising=0.0)
batch = {k: v.to(dev) for k, v in synth.make_batch(32, 100, 1000, 256, len(tok), 1234, 768).items()}
batch["video"] = batch["video"].to(torch.bfloat16)
batch["input_lens"] = (batch["input_ids"] != 0).sum(1).tolist()
print("valid encoder tokens:", sum(batch["input_lens"]), "of", batch["input_ids"].numel())
if len(sys.argv) > 2 and "dense" in sys.argv[2]:
    model.engine().pack = False
for kv in sys.argv[3:]:          # extra library options, e.g. gemm_dma=2
    k, v = kv.split("=")
    L.set_option(k, int(v))
    print("option", k, v)
for _ in range(2): tr.step(batch)
model.engine().overlap = False      # per-launch durations only mean something without concurrent kernels
with L.KernelTimer(detail=True) as kt:
    tr.step(batch)
summ = kt.summary()
tot = sum(v[1] for v in summ.values())
print(f"dropout {p}: timed launches total {tot:.2f} ms")
for k, (n, ms, w) in sorted(summ.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{k:58s} n={n:3d} {ms:7.3f} ms  {w / (ms / 1e3) / 1e12:7.1f} TF/s  avg {ms / n * 1e3:7.1f} us")
