"""Back-to-back: default dispatch vs the 4-wave 256x128x32 kernel forced (gemm_big=3) on forward / dgrad shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
def bench(M, N, K, tb, iters=30):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn((K, N) if tb else (N, K), device=dev).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3): L.gemm(A, B, C, M, N, K, transB=tb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): L.gemm(A, B, C, M, N, K, transB=tb)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, L.lib().v2s_last_gemm_kernel().decode()
shapes = [(32000, 2304, 768, 0), (32000, 768, 768, 0), (32000, 768, 3072, 0), (35200, 1536, 768, 0), (32000, 3072, 768, 0), (8192, 2304, 768, 0),
          (8192, 768, 3072, 0), (8192, 3072, 768, 0), (32000, 768, 3072, 1), (32000, 768, 2304, 1), (32000, 3072, 768, 1), (35200, 768, 1536, 1), (8192, 768, 768, 1)]
for rep in range(2):
    for M, N, K, tb in shapes:
        r = []
        for big in (1, 3):
            L.set_option("gemm_big", big)
            r.append(bench(M, N, K, bool(tb)))
        print(f"{'dgrad' if tb else 'NT   '} {M}x{N}x{K}: default {r[0][0]:6.1f} us ({r[0][1][:32]})   w4 {r[1][0]:6.1f} us")
L.set_option("gemm_big", 1)
