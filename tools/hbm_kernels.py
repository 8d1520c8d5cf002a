"""Launch the HBM-bound kernels of the path a few times each (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes): the fused
clip+Adam+shadow step over a t5-base sized arena, RMSNorm forward/backward at the encoder shape, the cross-attention decode step at
cfg-4 (B=64, 1100 keys)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
n = 289_205_000 // 64 * 64
p = torch.randn(n, device=dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev); g = torch.randn(n, device=dev) * 1e-3
pb = torch.empty(n, dtype=torch.bfloat16, device=dev)
gn = torch.ones(1, device=dev)
for i in range(3):
    L.adam_step(p, m, v, g, pb, n, 3e-4, 0.9, 0.999, 1e-8, 0.0, i + 1, gnorm_sq=gn, max_norm=1.0)
M, d = 32000, 768
x = torch.randn(M, d, device=dev).to(torch.bfloat16); w = torch.ones(d, device=dev); y = torch.empty_like(x); rstd = torch.empty(M, device=dev)
dy = torch.randn(M, d, device=dev).to(torch.bfloat16); dx = torch.empty_like(x); dadd = torch.randn(M, d, device=dev).to(torch.bfloat16); dw = torch.zeros(d, device=dev)
for i in range(3):
    L.rmsnorm_fwd(x, w, y, rstd, M, d, 1e-6)
for i in range(3):
    L.rmsnorm_bwd(x, w, rstd, dy, dx, dadd, dw, M, d)
B, H, S = 64, 12, 1100
W = H * 64
q = torch.randn(B, W, device=dev).to(torch.bfloat16); kv = torch.randn(B, S, 2 * W, device=dev).to(torch.bfloat16); o = torch.empty(B, W, dtype=torch.bfloat16, device=dev)
mask = torch.ones(B, S, dtype=torch.uint8, device=dev)
for i in range(3):
    L.decode_attn(B, H, S, q, W, kv, kv[:, :, W:], S * 2 * W, 2 * W, o, W, key_mask=mask, mask_ld=S)
torch.cuda.synchronize()
print("algorithmic bytes: adam", n * 30, "rmsnorm_fwd", M * d * 4, "rmsnorm_bwd", M * d * 8, "decode_attn", B * S * 2 * W * 2)
