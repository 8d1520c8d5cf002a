"""Interleaved A/B of the GEMM kernel variants on a few shapes: every round times every variant once (10 launches), rounds repeat;
reported: median and min over the rounds (sequential per-variant timing drifts with the chip's clock / thermal state).
Variants: old dispatch, 8-phase synchronous (256x256), its ablations (no global store / main loop only), deferred-epilogue persistent
form and its no-drain ablation."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"

def t(f, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

VARIANTS = [("old", 0, 0), ("p8-256", 2, 0), ("p8/no-store", 2, 1), ("p8/main-loop", 2, 2), ("p8d", 4, 0), ("p8d/no-drain", 4, 3)]
shapes = [(32000, 2304, 768, ""), (32000, 3072, 768, "act"), (32000, 3072, 768, ""), (8192, 8192, 8192, ""), (65536, 2048, 2048, "")]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) + ("",) for a in sys.argv[1:]]
for M, N, K, epi in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    kw = dict(act=L.ACT_RELU, dropout_p=0.1, dropout_seed=3) if epi == "act" else {}
    run = lambda: L.gemm(A, B, C, M, N, K, **kw)
    res = {v[0]: [] for v in VARIANTS}
    for rnd in range(7):
        for name, mode, dbg in VARIANTS:
            L.set_option("gemm_p8", mode); L.set_option("gemm_dbg", dbg)
            if rnd == 0:
                run(); run()
            res[name].append(t(run))
    L.set_option("gemm_p8", 1); L.set_option("gemm_dbg", 0)
    print(f"NT {M}x{N}x{K} {epi:3s}: " + "  ".join(f"{n} {statistics.median(v):7.1f}/{min(v):7.1f}" for n, v in res.items()), flush=True)
