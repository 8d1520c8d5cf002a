import os, sys
sys.path.insert(0, os.getcwd())
import torch
from vidchapters_amd import lib as L
dev = "cuda"
for (M, N) in ((3200, 2304), (3200, 768), (3200, 2048), (32000, 768)):
    X = torch.randn(M, N, device=dev).to(torch.bfloat16); out = torch.zeros(N, device=dev)
    ref = X.float().sum(0)
    L.colsum(X, M, N, out, accumulate=False); torch.cuda.synchronize()
    err = (out - ref).abs().max().item() / ref.abs().max().item()
    for _ in range(5): L.colsum(X, M, N, out, accumulate=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): L.colsum(X, M, N, out, accumulate=True)
    e1.record(); torch.cuda.synchronize()
    print(f"colsum {M}x{N}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us  rel err {err:.1e}")
