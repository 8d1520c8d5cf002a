"""Same-process A/B of the GEMM kernels of two builds of the library on the shapes of the cfg-2 train step (forward NT, dgrad NN,
wgrad TN).  usage: python tools/gemm_lib_ab.py libA.so libB.so"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
paths = sys.argv[1:] or [L.LIB_PATH]

def use(path):
    L.LIB_PATH = os.path.abspath(path); L._LIB = None; L.lib()

def make(kind, M, N, K):
    if kind == "NT":   # y[M,N] = x[M,K] w[N,K]^T
        A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16); kw = dict(transB=False)
    elif kind == "NN": # dx[M,N] = dy[M,K] w[K,N]
        A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(K, N, device=dev).to(torch.bfloat16); kw = dict(transB=True)
    else:              # dw[M,N] = dy[K,M]^T x[K,N]
        A = torch.randn(K, M, device=dev).to(torch.bfloat16); B = torch.randn(K, N, device=dev).to(torch.bfloat16); kw = dict(transA=True, transB=True)
    fp32 = kind == "TN"
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if fp32 else torch.bfloat16)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    def run(): L.gemm(A, B, C, M, N, K, workspace=ws, **kw)
    return run

def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

SHAPES = [("NT", 32000, 768, 768), ("NT", 32000, 2304, 768), ("NT", 32000, 3072, 768), ("NT", 32000, 768, 3072), ("NT", 8192, 768, 768),
          ("NT", 8192, 3072, 768), ("NT", 8192, 768, 3072), ("NT", 3200, 2304, 768), ("NT", 3200, 768, 2048), ("NT", 8192, 32200, 768),
          ("NN", 32000, 768, 768), ("NN", 32000, 768, 2304), ("NN", 32000, 768, 3072), ("NN", 32000, 3072, 768), ("NN", 8192, 768, 3072),
          ("NN", 8192, 768, 32200), ("TN", 768, 768, 32000), ("TN", 2304, 768, 32000), ("TN", 3072, 768, 32000), ("TN", 768, 3072, 32000),
          ("TN", 768, 768, 8192), ("TN", 3072, 768, 8192), ("TN", 32200, 768, 8192)]
tot = {p: 0.0 for p in paths}
for kind, M, N, K in SHAPES:
    use(paths[0]); run = make(kind, M, N, K)
    best = {p: 1e9 for p in paths}; names = {}
    for rep in range(3):
        for p in paths:
            use(p); best[p] = min(best[p], t(run)); names[p] = L.lib().v2s_last_gemm_kernel().decode()
    line = f"{kind} {M:6d}x{N:6d}x{K:6d}"
    for p in paths:
        tot[p] += best[p]
        line += f" | {os.path.basename(p)[:12]:12s} {best[p]:7.1f} us {2.0 * M * N * K / best[p] / 1e6:6.1f} TF/s {names[p][:34]:34s}"
    print(line)
print("sum:", {os.path.basename(p): round(v, 1) for p, v in tot.items()})
