"""How long does the main stream wait for the side streams at the joins of a training step?  (exposed weight-gradient / ViT tails)
usage: python tools/join_probe.py"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import SyntheticTokenizer, Vid2Seq, synth
from vidchapters_amd.train import Trainer
dev = torch.device("cuda", 0)
tok = SyntheticTokenizer(32100, 100)
model = Vid2Seq("t5-base", tokenizer=tok, vis_drop=0.1, enc_drop=0.1, dec_drop=0.1, init_seed=1234, device=dev).train()
eng = model.engine(); eng.pack = False
tr = Trainer(model, lr=3e-4, clip_max_norm=1.0, generative=1.0, denoising=0.0)
batch = {k: v.to(dev) for k, v in synth.make_batch(32, 100, 1000, 256, len(tok), 1234, 768).items()}
batch["video"] = batch["video"].to(torch.bfloat16)
marks = []
orig_join = eng.join_wgrads
def timed_join():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig_join(); e1.record(); marks.append(("join_wgrads", e0, e1))
eng.join_wgrads = timed_join
main = torch.cuda.current_stream()
orig_wait_stream = main.wait_stream
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
res = {}
for it in range(6):
    marks.clear()
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record(); tr.step(batch); s1.record(); torch.cuda.synchronize()
    res.setdefault("step", []).append(s0.elapsed_time(s1))
    for i, (name, e0, e1) in enumerate(marks):
        res.setdefault(f"{name}#{i}", []).append(e0.elapsed_time(e1))
for k, v in res.items():
    print(f"{k:20s} median {statistics.median(v):8.3f} ms")
