import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
M, d = 32000, 768
x = torch.randn(M, d, device=dev).to(torch.bfloat16); w = torch.ones(d, device=dev); rstd = torch.rand(M, device=dev) + 0.5
dy = torch.randn(M, d, device=dev).to(torch.bfloat16); dx = torch.empty_like(x); dadd = torch.randn(M, d, device=dev).to(torch.bfloat16); dw = torch.zeros(d, device=dev)
for name, add in (("with dx_add", dadd), ("no dx_add", None)):
    for _ in range(3): L.rmsnorm_bwd(x, w, rstd, dy, dx, add, dw, M, d)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): L.rmsnorm_bwd(x, w, rstd, dy, dx, add, dw, M, d)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    b = M * d * 2 * (4 if add is not None else 3)
    print(f"rmsnorm_bwd {name}: {us:.1f} us  {b / us / 1e6:.2f} TB/s")
