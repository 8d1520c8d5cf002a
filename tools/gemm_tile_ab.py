import sys; sys.path.insert(0, "/root/repo")
import torch
from vidchapters_amd import lib as L
dev = "cuda"
def bench(M, N, K, epi, iters=30):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16); R = torch.randn(M, N, device=dev).to(torch.bfloat16)
    kw = dict(residual=R, dropout_p=0.1, dropout_seed=3) if epi == "res" else (dict(act=L.ACT_RELU, dropout_p=0.1, dropout_seed=3) if epi == "act" else {})
    for _ in range(3): L.gemm(A, B, C, M, N, K, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): L.gemm(A, B, C, M, N, K, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, L.lib().v2s_last_gemm_kernel().decode()
L.set_option("gemm_dma", 2)
shapes = [(32000, 768, 768, "res"), (32000, 768, 3072, "res"), (35200, 1536, 768, ""), (32000, 2304, 768, ""), (32000, 3072, 768, "act"), (8192, 2304, 768, ""), (8192, 3072, 768, "act"), (8192, 32200, 768, "")]
for rep in range(2):
    for M, N, K, epi in shapes:
        r = []
        for big in (1, 0, 2):
            L.set_option("gemm_big", big)
            r.append(bench(M, N, K, epi))
        print(f"NT {M}x{N}x{K} {epi:3s}: big=1 {r[0][0]:7.1f} us ({r[0][1][:28]})  big=0 {r[1][0]:7.1f} us  big=2 {r[2][0]:7.1f} us ({r[2][1][:28]})")
