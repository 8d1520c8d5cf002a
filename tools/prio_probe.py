"""Does giving the step's main stream (forward / dgrad / attention chain) a higher HIP stream priority than the weight-gradient, ViT and
K|V side streams shorten the step?  Interleaved: the same Trainer.step() issued from the default stream and from a high-priority stream.
usage: python tools/prio_probe.py [steps]"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import SyntheticTokenizer, Vid2Seq, synth
from vidchapters_amd.train import Trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
tok = SyntheticTokenizer(32100, 100)
model = Vid2Seq("t5-base", tokenizer=tok, vis_drop=0.1, enc_drop=0.1, dec_drop=0.1, init_seed=1234, device=dev).train()
model.engine().pack = False
tr = Trainer(model, lr=3e-4, clip_max_norm=1.0, generative=1.0, denoising=0.0)
batch = {k: v.to(dev) for k, v in synth.make_batch(32, 100, 1000, 256, len(tok), 1234, 768).items()}
batch["video"] = batch["video"].to(torch.bfloat16)
hi = torch.cuda.Stream(priority=-1)
print("priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
def run(use_hi, n):
    t = []
    for _ in range(n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if use_hi:
            hi.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(hi):
                e0.record(); tr.step(batch); e1.record()
        else:
            e0.record(); tr.step(batch); e1.record()
        torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1))
    return t
run(False, 2); run(True, 2)
a, b = [], []
for _ in range(steps):
    a += run(False, 1); b += run(True, 1)
print(f"default-priority main stream: median {statistics.median(a):.2f} ms (min {min(a):.2f});  high-priority main stream: median {statistics.median(b):.2f} ms (min {min(b):.2f})")
