import sys; sys.path.insert(0, "/root/repo")
import torch
from vidchapters_amd import lib as L
dev = "cuda"
n, d, V = 32000, 768, 32200
ids = torch.randint(2, V, (n,), device=dev)
dy = torch.randn(n, d, device=dev).to(torch.bfloat16)
tab = torch.zeros(V, d, device=dev)
for p in (0.0, 0.1):
    for _ in range(2): L.embed_bwd(ids, dy, tab, n, d, V, p, 5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): L.embed_bwd(ids, dy, tab, n, d, V, p, 5)
    e1.record(); torch.cuda.synchronize()
    print(f"embed_bwd p={p}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
