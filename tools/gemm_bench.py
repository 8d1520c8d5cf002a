"""Micro-benchmark of the GEMM kernel on the shapes of the Vid2Seq step (and a square reference shape)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L

dev = "cuda"
def bench(M, N, K, ta, tb, f32=False, acc=False, ws=None, iters=20):
    A = torch.randn((K, M) if ta else (M, K), device=dev).to(torch.bfloat16)
    B = torch.randn((K, N) if tb else (N, K), device=dev).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    kw = dict(transA=ta, transB=tb, accumulate=acc, workspace=ws)
    for _ in range(3): L.gemm(A, B, C, M, N, K, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): L.gemm(A, B, C, M, N, K, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2.0 * M * N * K / ms / 1e9, ms * 1e3

shapes = [("enc qkv fwd", 32000, 2304, 768, 0, 0), ("enc o fwd", 32000, 768, 768, 0, 0), ("enc wi fwd", 32000, 3072, 768, 0, 0),
          ("enc wo fwd", 32000, 768, 3072, 0, 0), ("square 4096", 4096, 4096, 4096, 0, 0), ("square 8192", 8192, 8192, 8192, 0, 0),
          ("enc qkv dgrad", 32000, 768, 2304, 0, 1), ("enc wi dgrad", 32000, 768, 3072, 0, 1), ("enc wo dgrad", 32000, 3072, 768, 0, 1),
          ("lm head fwd", 8192, 32200, 768, 0, 0)]
ws = torch.empty(64 * 1024 * 1024 // 4, device=dev)
for big in (1, 2, 0):
    L.set_option("gemm_big", big)
    print(f"--- gemm_big={big}")
    for name, M, N, K, ta, tb in shapes:
        tf, us = bench(M, N, K, bool(ta), bool(tb), f32=(name == "lm head fwd"))
        print(f"{name:16s} M={M:6d} N={N:6d} K={K:6d}: {tf:7.1f} TF/s  {us:8.1f} us")
    for name, M, N, K in (("wgrad qkv", 2304, 768, 32000), ("wgrad wi", 3072, 768, 32000), ("wgrad o", 768, 768, 32000), ("wgrad wo", 768, 3072, 32000)):
        tf, us = bench(M, N, K, True, True, f32=True, acc=True, ws=ws)
        print(f"{name:16s} M={M:6d} N={N:6d} K={K:6d}: {tf:7.1f} TF/s  {us:8.1f} us (split-K)")
