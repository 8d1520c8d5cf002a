"""Interleaved A/B of library options on the full cfg-2 train step: one model / trainer / batch, the option settings alternate step
by step, HIP-event time per step, median per setting (separate bench.py runs differ by +-2-3 % from box to box and run to run).
usage: python tools/step_ab.py "gemm_p8=0" "gemm_p8=1" [--steps 12] [--model t5-base]      (also "lib=path/to/other_build.so")"""
import sys, os, statistics, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import SyntheticTokenizer, Vid2Seq, synth
from vidchapters_amd import lib as L
from vidchapters_amd.train import Trainer

ap = argparse.ArgumentParser()
ap.add_argument("settings", nargs="+")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--model", default="t5-base")
ap.add_argument("--frames", type=int, default=100)
ap.add_argument("--asr", type=int, default=1000)
ap.add_argument("--block", type=int, default=1, help="time blocks of this many back-to-back steps (pipelined like bench.py) instead of single synchronised steps")
a = ap.parse_args()
dev = torch.device("cuda", 0)
tok = SyntheticTokenizer(32100, 100)
model = Vid2Seq(a.model, num_features=a.frames, tokenizer=tok, vis_drop=0.1, enc_drop=0.1, dec_drop=0.1, init_seed=1234, device=dev).train()
model.engine().pack = False
tr = Trainer(model, lr=3e-4, clip_max_norm=1.0, generative=1.0, denoising=0.0)
batch = {k: v.to(dev) for k, v in synth.make_batch(32, a.frames, a.asr, 256, len(tok), 1234, 768).items()}
batch["video"] = batch["video"].to(torch.bfloat16)
batch["input_lens"] = (batch["input_ids"] != 0).sum(1).tolist()         # host-side lengths, as a data loader knows them
batch["output_lens"] = (batch["output_ids"] != 0).sum(1).tolist()
defaults = {}
def apply(setting):
    """"gemm_p8=0,eng:fused_head=1": library options, and engine attributes with the eng: prefix"""
    for kv in filter(None, setting.split(",")):
        k, v = kv.split("=")
        if k.startswith("eng:"):
            setattr(model.engine(), k[4:], int(v))
            continue
        if k.startswith("tr:"):              # Trainer attribute, e.g. "tr:defer_adam=1"
            setattr(tr, k[3:], int(v))
            continue
        if k == "lib":                       # another build of the library (same ABI): "lib=tools/lib_head.so"
            L.LIB_PATH = os.path.abspath(v); L._LIB = None; L.lib()
            continue
        defaults.setdefault(k, L.get_option(k))
        L.set_option(k, int(v))
for s in a.settings:
    apply(s); tr.step(batch)
torch.cuda.synchronize()
times = {s: [] for s in a.settings}
clk = {s: [] for s in a.settings}     # effective shader clock of each timed block (v2s_clock_probe: s_memtime cycles / 100 MHz ticks)
for i in range(a.steps):
    for s in a.settings:
        apply(s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        p0 = torch.zeros(8, 4, dtype=torch.int64, device=dev); p1 = torch.zeros(8, 4, dtype=torch.int64, device=dev)
        e0.record(); L.clock_probe(p0)
        for _ in range(a.block):
            tr.step(batch)
        L.clock_probe(p1); e1.record(); torch.cuda.synchronize()
        times[s].append(e0.elapsed_time(e1) / a.block)
        clk[s].append(L.effective_sclk_mhz(p0.cpu(), p1.cpu()) or 0.0)
for s in a.settings:
    med = statistics.median(times[s])
    print(f"{s:40s} median {med:7.2f} ms  min {min(times[s]):7.2f}  -> {32 / med * 1e3:6.1f} samples/s   sclk {statistics.median(clk[s]):6.0f} MHz")
for k, v in defaults.items():
    L.set_option(k, v)
