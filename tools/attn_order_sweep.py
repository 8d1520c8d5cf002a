"""dK/dV kernel time against the block dispatch order (library option attn_order: 1 = every XCD its own contiguous range of (sequence, head) groups, 0 = groups
dealt to the XCDs round-robin) and the length distribution of the batch.
usage: python tools/attn_order_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
B, H = 32, 12
W = H * 64


def run(N, lo, orders):
    torch.manual_seed(0)
    qkv = (torch.randn(B, N, 3 * W, device=dev) * 0.5).to(torch.bfloat16); d_o = torch.randn(B, N, W, device=dev).to(torch.bfloat16)
    o = torch.empty(B, N, W, dtype=torch.bfloat16, device=dev); ml = torch.empty(B, H, N, 2, device=dev)
    dqkv = torch.empty_like(qkv); delta = torch.empty(B, H, N, 4, device=dev)
    diag = torch.randn(H, 2 * N - 1, device=dev); ddiag = torch.zeros(H, 2 * N - 1, device=dev)
    lens = torch.randint(int(lo * N), N + 1, (B,), device=dev)
    mask = (torch.arange(N, device=dev)[None, :] < lens[:, None]).to(torch.uint8).contiguous()
    st = (N * 3 * W, 3 * W)
    out = []
    for order, klen in orders:
        L.set_option("attn_order", order)
        a = L.attn_args(B, H, N, N, qkv, qkv[..., W:], qkv[..., 2 * W:], o, st, st, st, (N * W, W), ml=ml, scale=1.0, bias_diag=diag, key_mask=mask,
                        dropout_p=0.1, dropout_seed=5)
        L.attn_fwd(a)
        L.set_option("attn_bwd_part", 0)
        bw = lambda: L.attn_bwd(a, d_o, (N * W, W), delta, dqkv, dqkv[..., W:], dqkv[..., 2 * W:], st, st, st, dbias_diag=ddiag, far=(-91, 91))
        bw()
        L.set_option("attn_bwd_part", 2)
        best = 1e9
        for rep in range(3):
            for _ in range(3): bw()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): bw()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        out.append(best)
    L.set_option("attn_bwd_part", 0); L.set_option("attn_order", 0)
    return out


orders = [(1, 0), (0, 0)] + [(400 + n, 0) for n in (5, 7, 8, 9, 11, 13, 15)]
print("order:                 " + " ".join(f"{('plain' if o == 1 else 'hint' if o == 0 else str(o)):>7s}" for o, _ in orders))
for N, lo in ((1000, 0.7), (1100, 0.7), (1100, 1.0), (2000, 0.7), (2000, 1.0), (1200, 0.7)):
    r = run(N, lo, orders)
    print(f"N={N:5d} lens>= {lo:.1f}N    " + " ".join(f"{x:7.1f}" for x in r))
