import sys; sys.path.insert(0, "/root/repo")
import torch
from vidchapters_amd import lib as L
dev = "cuda"
def bench(M, N, K, epi, iters=30):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(K, N, device=dev).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16); Z = torch.randn(M, N, device=dev).to(torch.bfloat16)
    kw = dict(dact=L.ACT_RELU, z=Z, dropout_p=0.1, dropout_seed=3) if epi == "act" else {}
    for _ in range(3): L.gemm(A, B, C, M, N, K, transB=True, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): L.gemm(A, B, C, M, N, K, transB=True, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, L.lib().v2s_last_gemm_kernel().decode()
shapes = [(32000, 3072, 768, "act"), (8192, 3072, 768, "act"), (8192, 768, 32200, ""), (32000, 768, 3072, ""), (3200, 2048, 768, "act")]
for rep in range(2):
    for M, N, K, epi in shapes:
        r = []
        for big in (1, 0):
            L.set_option("gemm_big", big)
            r.append(bench(M, N, K, epi))
        print(f"dgrad {M}x{N}x{K} {epi:3s}: big=1 {r[0][0]:7.1f} us ({r[0][1][:30]})  big=0 {r[1][0]:7.1f} us")
