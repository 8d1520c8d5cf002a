"""The N = 768 dgrad / forward shapes of the cfg-2 step (M = 32000 / 35200 rows) under the default dispatch, the 256 x 256 and the 256 x 128
8-phase kernels (gemm_p8 = 1 / 2 / 3), interleaved.  usage: python tools/gemm_n768_ab.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
def mk(M, N, K, tb):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = (torch.randn(K, N, device=dev) if tb else torch.randn(N, K, device=dev)).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16); R = torch.randn(M, N, device=dev).to(torch.bfloat16)
    return A, B, C, R
def run(t, M, N, K, tb, res, iters=20):
    A, B, C, R = t
    kw = dict(residual=R) if res else {}
    for _ in range(2): L.gemm(A, B, C, M, N, K, transB=tb, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): L.gemm(A, B, C, M, N, K, transB=tb, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, L.lib().v2s_last_gemm_kernel().decode()
shapes = [(32000, 768, 3072, True, False), (32000, 768, 2304, True, False), (35200, 768, 1536, True, True), (32000, 768, 768, True, False),
          (32000, 768, 768, False, True), (32000, 768, 3072, False, True)]
for M, N, K, tb, res in shapes:
    t = mk(M, N, K, tb)
    out = {}
    for rep in range(3):
        for mode in (1, 2, 3):
            L.set_option("gemm_p8", mode)
            us, name = run(t, M, N, K, tb, res)
            out.setdefault(mode, []).append((us, name))
    L.set_option("gemm_p8", 1)
    gf = 2.0 * M * N * K / 1e9
    s = "  ".join(f"p8={m}: {min(x[0] for x in out[m]):7.1f} us {gf / min(x[0] for x in out[m]) * 1e3:6.0f} TF/s ({out[m][0][1][:22]})" for m in (1, 2, 3))
    print(f"{M}x{N}x{K} {'NT(dgrad)' if tb else 'NN(fwd)  '}{' +res' if res else '     '}: {s}")
