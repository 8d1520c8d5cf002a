"""Ablation of the persistent write-out-wave GEMM and of the 128x128 LDS-DMA kernel it replaces (option gemm_dbg): where does a
K = 768 launch spend its time?  usage: python tools/gemm_ps_ablate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
SHAPES = [("NT", 32000, 768, 768), ("NN", 32000, 768, 768), ("NT", 32000, 2304, 768), ("NT", 32000, 768, 3072), ("NT", 8192, 2304, 768)]
VARIANTS = [("dma", dict(gemm_ps=0, gemm_p8=0, gemm_big=0)), ("dma-ml", dict(gemm_ps=0, gemm_p8=0, gemm_big=0, gemm_dbg=2)),
            ("ps2", dict(gemm_ps=2, gemm_ps_nst=2)), ("ps2-ml", dict(gemm_ps=2, gemm_ps_nst=2, gemm_dbg=3)),
            ("ps3", dict(gemm_ps=2, gemm_ps_nst=3)), ("ps3-ml", dict(gemm_ps=2, gemm_ps_nst=3, gemm_dbg=3)),
            ("ps4", dict(gemm_ps=2, gemm_ps_nst=4)), ("ps4-nostore", dict(gemm_ps=2, gemm_ps_nst=4, gemm_dbg=1)),
            ("ps4-bar", dict(gemm_ps=2, gemm_ps_nst=4, gemm_dbg=2)), ("ps4-ml", dict(gemm_ps=2, gemm_ps_nst=4, gemm_dbg=3)), ("default", dict())]
DEF = dict(gemm_ps=0, gemm_ps_nst=4, gemm_p8=1, gemm_big=1, gemm_dbg=0)
def timed(f, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for kind, M, N, K in SHAPES:
    A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    B = (torch.randn(*((K, N) if kind == "NN" else (N, K)), device=dev) * 0.5).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    kw = dict(transB=(kind == "NN"), ldb=N if kind == "NN" else K)
    res = {n: [] for n, _ in VARIANTS}
    kern = {}
    for rep in range(5):
        for name, opts in VARIANTS:
            for k, v in {**DEF, **opts}.items(): L.set_option(k, v)
            f = lambda: L.gemm(A, B, C, M, N, K, **kw)
            f(); f()
            kern[name] = L.lib().v2s_last_gemm_kernel().decode()
            res[name].append(timed(f, 20))
    for k, v in DEF.items(): L.set_option(k, v)
    print(f"{kind} {M}x{N}x{K}: " + "  ".join(f"{n} {sorted(res[n])[2]:.1f}" for n, _ in VARIANTS) + f"   [default = {kern['default']}]", flush=True)
