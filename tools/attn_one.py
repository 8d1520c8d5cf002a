"""Run the encoder-layer attention (B=32, H=12, N tokens, bias + key mask + dropout + dbias) forward and backward a few times (for
rocprofv3 --pmc passes).  usage: attn_one.py [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
B, H, N = 32, 12, int(sys.argv[1]) if len(sys.argv) > 1 else 1000
W = H * 64
torch.manual_seed(0)
q = (torch.randn(B, N, W, device=dev) * 0.5).to(torch.bfloat16); k = (torch.randn(B, N, W, device=dev) * 0.5).to(torch.bfloat16)
v = torch.randn(B, N, W, device=dev).to(torch.bfloat16); d_o = torch.randn(B, N, W, device=dev).to(torch.bfloat16)
o = torch.empty_like(q); ml = torch.empty(B, H, N, 2, device=dev); delta = torch.empty(B, H, N, 4, device=dev)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
diag = torch.randn(H, 2 * N - 1, device=dev); ddiag = torch.zeros(H, 2 * N - 1, device=dev)
lens = torch.randint(int(0.7 * N), N + 1, (B,), device=dev)
mask = (torch.arange(N, device=dev)[None, :] < lens[:, None]).to(torch.uint8).contiguous()
st = (N * W, W)
for _ in range(4):
    a = L.attn_args(B, H, N, N, q, k, v, o, st, st, st, st, ml=ml, scale=1.0, bias_diag=diag, key_mask=mask, dropout_p=0.1, dropout_seed=5)
    L.attn_fwd(a)
    L.attn_bwd(a, d_o, st, delta, dq, dk, dv, st, st, st, dbias_diag=ddiag, far=(-91, 91))
torch.cuda.synchronize()
