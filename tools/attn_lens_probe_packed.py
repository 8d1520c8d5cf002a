"""dK/dV kernel with packed sequences of length `ln` (seq_off): key blocks beyond a sequence's end return at the very top of the kernel -- what does a launched-and-returned block cost?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
B, H, N = 32, 12, 1000
W = H * 64
torch.manual_seed(0)
def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for ln in (1000, 500, 128):
    M = B * ln
    qkv = (torch.randn(M, 3 * W, device=dev) * 0.5).to(torch.bfloat16); d_o = torch.randn(M, W, device=dev).to(torch.bfloat16)
    o = torch.empty(M, W, dtype=torch.bfloat16, device=dev); ml = torch.zeros(B, H, N, 2, device=dev)
    dqkv = torch.empty_like(qkv); delta = torch.zeros(B, H, N, 4, device=dev)
    diag = torch.randn(H, 2 * N - 1, device=dev); ddiag = torch.zeros(H, 2 * N - 1, device=dev)
    off = (torch.arange(B + 1, device=dev) * ln).to(torch.int32)
    st = (0, 3 * W)
    a = L.attn_args(B, H, N, N, qkv, qkv[:, W:], qkv[:, 2 * W:], o, st, st, st, (0, W), ml=ml, scale=1.0, bias_diag=diag, dropout_p=0.1, dropout_seed=5, seq_off=off)
    f = t(lambda: L.attn_fwd(a))
    r = []
    for part in (1, 2):
        L.set_option("attn_bwd_part", part)
        r.append(t(lambda: L.attn_bwd(a, d_o, (0, W), delta, dqkv, dqkv[:, W:], dqkv[:, 2 * W:], st, st, st, dbias_diag=ddiag, far=(-91, 91))))
    L.set_option("attn_bwd_part", 0)
    print(f"packed, every sequence {ln:4d} long: fwd {f:7.1f}  dQ {r[0]:7.1f}  dK/dV {r[1]:7.1f} us")
