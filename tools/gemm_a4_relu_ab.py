"""Same-process interleaved A/B of the forward ReLU (+ dropout) epilogue: the persistent gemm_a4p kernel (option gemm_a4_relu = 1) against the
kernels it replaces (0) on the FFN wi shapes of the cfg-2 step.  usage: python tools/gemm_a4_relu_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vidchapters_amd import lib as L

dev = "cuda"


def timed(f, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, N, K, what in ((32000, 3072, 768, "enc wi fwd"), (8192, 3072, 768, "dec wi fwd"), (27800, 3072, 768, "enc wi fwd, padding-free")):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for p in (0.0, 0.1):
        res, kern, outs = {0: [], 1: []}, {}, {}
        for rep in range(5):
            for v in (0, 1):
                L.set_option("gemm_a4_relu", v)
                f = lambda: L.gemm(A, B, C, M, N, K, act=L.ACT_RELU, dropout_p=p, dropout_seed=7)
                f(); kern[v] = L.lib().v2s_last_gemm_kernel().decode(); outs[v] = C.clone()
                res[v].append(timed(f, 20))
        L.set_option("gemm_a4_relu", 1)
        m0, m1 = sorted(res[0])[2], sorted(res[1])[2]
        fl = 2.0 * M * N * K
        print(f"{M}x{N}x{K} {what:26s} p={p}: {kern[0]} {m0:7.1f} us ({fl / m0 / 1e6:5.0f} TF/s)   {kern[1]} {m1:7.1f} us ({fl / m1 / 1e6:5.0f} TF/s)   "
              f"ratio {m0 / m1:.3f}  identical {bool(torch.equal(outs[0], outs[1]))}", flush=True)
