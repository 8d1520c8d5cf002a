"""Ablations of gemm_a4_kernel's main loop (profiling build tools/libvid2seq_hip_abl.so: tools/build_a4_ablations.sh): what does the loop cost
without its LDS-DMA requests, without its fragment reads, with neither, without its barriers?  Main loop only (gemm_dbg = 2 and 11-14).
usage: python tools/gemm_a4_ablate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
PLAN = next((a[7:] for a in sys.argv[1:] if a.startswith("--plan=")), "spread")
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"libvid2seq_hip_abl_{PLAN}.so")
print(f"slot plan: {PLAN}")
dev = "cuda"
VAR = [("main loop", 2), ("no DMA (LDS stays zero: MFMAs on zeros)", 11), ("no frag reads (stale registers)", 12), ("MFMA + barriers only (stale registers)", 13), ("no barrier", 14), ("no vmcnt wait", 15)]
def timed(f, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
if "--persistent" in sys.argv:      # the persistent form, full kernel and stores dropped
    VAR = [("a4p", 0), ("a4p stores dropped", 1)]
L.set_option("gemm_a4", 2 if "--persistent" in sys.argv else 3)
for M, N, K in ((32000, 2304, 768), (8192, 8192, 8192), (32000, 768, 3072)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    res = {n: [] for n, _ in VAR}
    for rep in range(5):
        for n, d in VAR:
            L.set_option("gemm_dbg", d)
            f = lambda: L.gemm(A, B, C, M, N, K)
            f(); res[n].append(timed(f, 10))
    L.set_option("gemm_dbg", 0)
    fl = 2.0 * M * N * K
    print(f"{M}x{N}x{K}: " + "  ".join(f"{n} {sorted(v)[2]:.1f} us ({fl / sorted(v)[2] / 1e6:.0f})" for n, v in res.items()), flush=True)
L.set_option("gemm_a4", 1)
