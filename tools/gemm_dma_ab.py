import sys; sys.path.insert(0, "/root/repo")
import torch
from vidchapters_amd import lib as L
dev = "cuda"
def bench(M, N, K, tb, iters=30):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn((K, N) if tb else (N, K), device=dev).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16); R = torch.randn(M, N, device=dev).to(torch.bfloat16)
    for _ in range(3): L.gemm(A, B, C, M, N, K, transB=tb, residual=R, dropout_p=0.1, dropout_seed=3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): L.gemm(A, B, C, M, N, K, transB=tb, residual=R, dropout_p=0.1, dropout_seed=3)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, L.lib().v2s_last_gemm_kernel().decode()
shapes = [(8192, 768, 3072), (8192, 768, 768), (8192, 2304, 768), (8192, 3072, 768), (3200, 768, 2048), (3200, 2048, 768), (3200, 2304, 768), (3200, 768, 768), (32000, 768, 768), (32000, 768, 3072)]
for rep in range(2):
    for M, N, K in shapes:
        r = []
        for dma in (1, 2):
            L.set_option("gemm_dma", dma)
            r.append(bench(M, N, K, False))
        print(f"NT {M}x{N}x{K}: dma=1 {r[0][0]:7.1f} us ({r[0][1]})   dma=2 {r[1][0]:7.1f} us ({r[1][1]})")
L.set_option("gemm_dma", 1)
