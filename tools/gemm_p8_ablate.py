"""Ablation of the 8-phase GEMM kernel on the wide K=768 shapes of the step: full kernel / epilogue without the global store /
main loop only (option gemm_dbg).  Tells how much of a tile is store burst, LDS staging + post-ops, prologue + main loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"

def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for M, N, K, epi in [(32000, 2304, 768, ""), (32000, 3072, 768, "act"), (32000, 768, 3072, "res"), (8192, 8192, 8192, ""), (32000, 2304, 128, ""), (32000, 2304, 256, "")]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16); R = torch.randn(M, N, device=dev).to(torch.bfloat16)
    kw = dict(residual=R, dropout_p=0.1, dropout_seed=3) if epi == "res" else (dict(act=L.ACT_RELU, dropout_p=0.1, dropout_seed=3) if epi == "act" else {})
    run = lambda: L.gemm(A, B, C, M, N, K, **kw)
    line = f"NT {M}x{N}x{K} {epi:3s}:"
    for mode, name in ((2, "p8-256"), (3, "p8-128")):
        L.set_option("gemm_p8", mode)
        for dbg, dn in ((0, "full"), (1, "no-store"), (2, "main-loop")):
            L.set_option("gemm_dbg", dbg)
            line += f"  {name}/{dn} {min(t(run) for _ in range(3)):7.1f}"
        L.set_option("gemm_dbg", 0)
    L.set_option("gemm_p8", 0)
    line += f"  old {min(t(run) for _ in range(3)):7.1f} us"
    print(line, flush=True)
