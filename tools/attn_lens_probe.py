"""Attention kernel times of the encoder layer when EVERY sequence of the dense, key-masked batch has the same valid length (B from the environment): how the
three kernels' time follows the work.  usage: [B=32] python tools/attn_lens_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
B, H, N = int(os.environ.get('B', '32')), 12, 1000
W = H * 64
torch.manual_seed(0)
qkv = (torch.randn(B, N, 3 * W, device=dev) * 0.5).to(torch.bfloat16); d_o = torch.randn(B, N, W, device=dev).to(torch.bfloat16)
o = torch.empty(B, N, W, dtype=torch.bfloat16, device=dev); ml = torch.empty(B, H, N, 2, device=dev)
dqkv = torch.empty_like(qkv); delta = torch.empty(B, H, N, 4, device=dev)
diag = torch.randn(H, 2 * N - 1, device=dev); ddiag = torch.zeros(H, 2 * N - 1, device=dev)
st = (N * 3 * W, 3 * W)
def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for ln in (1000, 896, 750, 640, 500, 250, 128):
    lens = torch.full((B,), ln, device=dev)
    mask = (torch.arange(N, device=dev)[None, :] < lens[:, None]).to(torch.uint8).contiguous()
    a = L.attn_args(B, H, N, N, qkv, qkv[..., W:], qkv[..., 2 * W:], o, st, st, st, (N * W, W), ml=ml, scale=1.0, bias_diag=diag, key_mask=mask, dropout_p=0.1, dropout_seed=5)
    f = t(lambda: L.attn_fwd(a))
    r = []
    for part in (1, 2):
        L.set_option("attn_bwd_part", part)
        r.append(t(lambda: L.attn_bwd(a, d_o, (N * W, W), delta, dqkv, dqkv[..., W:], dqkv[..., 2 * W:], st, st, st, dbias_diag=ddiag, far=(-91, 91))))
    L.set_option("attn_bwd_part", 0)
    print(f"all lens = {ln:4d}: fwd {f:7.1f}  dQ {r[0]:7.1f}  dK/dV {r[1]:7.1f} us")
