"""Captured training step (Trainer.step_graph) with and without the three-stream overlap inside the capture, against the eager step.
usage: python tools/graph_step_probe.py     (PROBE=captured | eager: only the overlapped captured / eager step -- for sweeps of the runtime's
graph knobs, e.g. DEBUG_HIP_FORCE_GRAPH_QUEUES=8 PROBE=captured python tools/graph_step_probe.py)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vidchapters_amd import SyntheticTokenizer, Vid2Seq, synth
from vidchapters_amd.train import Trainer

dev = torch.device("cuda", 0)
tok = SyntheticTokenizer(32100, 100)
batch = {k: v.to(dev) for k, v in synth.make_batch(32, 100, 1000, 256, len(tok), 1234, 768).items()}
batch["video"] = batch["video"].to(torch.bfloat16)


def measure(overlap, graph):
    model = Vid2Seq("t5-base", tokenizer=tok, vis_drop=0.1, enc_drop=0.1, dec_drop=0.1, init_seed=1234, device=dev).train()
    eng = model.engine()
    eng.pack = False
    eng.overlap = overlap
    tr = Trainer(model, lr=3e-4, clip_max_norm=1.0, generative=1.0, denoising=0.0)
    f = (lambda: tr.step_graph(batch)) if graph else (lambda: tr.step(batch))
    for _ in range(4):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(4):
            f()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 4 * 1e3)
    del tr, model
    torch.cuda.empty_cache()
    return sorted(ts)[1]


only = os.environ.get("PROBE", "")
if only:
    knobs = {k: v for k, v in os.environ.items() if k.startswith(("DEBUG_HIP_", "DEBUG_CLR_", "GPU_MAX_HW"))}
    print(f"overlap True  {only:8s}: {measure(True, only == 'captured'):7.2f} ms/step   {knobs}")
    sys.exit(0)
for overlap in (True, False):
    for graph in (False, True):
        print(f"overlap {overlap!s:5s} {'captured' if graph else 'eager   '}: {measure(overlap, graph):7.2f} ms/step")
