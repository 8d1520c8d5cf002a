"""The masked wo dgrad (dact = RELU, z, dropout scale) on the default dispatch (gemm_a4 = 0) and on the persistent asm kernel (gemm_a4 = 5), warm
(the same operands re-used) and COLD (a 1 GB buffer rewritten between launches: what the launch sees inside a train step).
usage: [V2S_LIB=...] python tools/gemm_a4_dact_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vidchapters_amd import lib as L


def timed(f, n, flush=None):
    tot = 0.0
    for _ in range(n):
        if flush is not None:
            flush.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3


flush = torch.zeros(256 << 20, device="cuda")            # 1 GB of fp32: evicts L2 and the memory-side cache
for M, N, K in ((32000, 3072, 768), (8192, 3072, 768)):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); B = torch.randn(K, N, device="cuda").to(torch.bfloat16)
    z = (torch.relu(torch.randn(M, N, device="cuda")) * (torch.rand(M, N, device="cuda") > 0.1)).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    res = {}
    for rep in range(3):
        for mode in (0, 5):
            L.set_option("gemm_a4", mode)
            f = lambda: L.gemm(A, B, C, M, N, K, transB=True, ldb=N, dact=L.ACT_RELU, z=z, dropout_p=0.1, dropout_seed=3)
            f(); k = L.lib().v2s_last_gemm_kernel().decode()
            res.setdefault(mode, []).append((timed(f, 10), timed(f, 6, flush), k))
    L.set_option("gemm_a4", 1)
    for mode in (0, 5):
        w, c = sorted(x[0] for x in res[mode])[1], sorted(x[1] for x in res[mode])[1]
        print(f"wo dgrad (ReLU mask + dropout scale) {M}x{N}x{K} gemm_a4={mode}: warm {w:.1f} us, cold {c:.1f} us  {res[mode][0][2]}", flush=True)
