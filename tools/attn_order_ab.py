"""All three attention kernels of the encoder layer and the step's other attention shapes under the two dispatch orders (option attn_order: 1 = every XCD its own
contiguous range of (sequence, head) groups, rounds 1-5; 0 = groups dealt to the XCDs round-robin, round 6).  usage: python tools/attn_order_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
ORDERS = [int(x) for x in os.environ.get("ORDERS", "1,0,1,0").split(",")]


def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def case(name, B, H, Nq, Nk, lo, bias, causal, cross):
    torch.manual_seed(0)
    W = H * 64
    q = (torch.randn(B, Nq, W, device=dev) * 0.5).bfloat16(); k = (torch.randn(B, Nk, W, device=dev) * 0.5).bfloat16(); v = torch.randn(B, Nk, W, device=dev).bfloat16()
    d_o = torch.randn(B, Nq, W, device=dev).bfloat16(); o = torch.empty_like(q); ml = torch.empty(B, H, Nq, 2, device=dev); delta = torch.empty(B, H, Nq, 4, device=dev)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    diag = torch.randn(H, Nq + Nk - 1, device=dev) if bias else None
    ddiag = torch.zeros(H, Nq + Nk - 1, device=dev) if bias else None
    mask = None
    if lo < 1.0:
        lens = torch.randint(int(lo * Nk), Nk + 1, (B,), device=dev)
        mask = (torch.arange(Nk, device=dev)[None, :] < lens[:, None]).to(torch.uint8).contiguous()
    sq, sk = (Nq * W, W), (Nk * W, W)
    res = {}
    for order in ORDERS:
        L.set_option("attn_order", order)
        a = L.attn_args(B, H, Nq, Nk, q, k, v, o, sq, sk, sk, sq, ml=ml, scale=1.0, bias_diag=diag, key_mask=mask, causal=causal, dropout_p=0.1, dropout_seed=5)
        f = t(lambda: L.attn_fwd(a))
        r = []
        for part in (1, 2):
            L.set_option("attn_bwd_part", part)
            r.append(t(lambda: L.attn_bwd(a, d_o, sq, delta, dq, dk, dv, sq, sk, sk, dbias_diag=ddiag, far=(-91, 91) if bias else (0, 0))))
        L.set_option("attn_bwd_part", 0)
        cur = res.setdefault(order, [1e9] * 3)
        res[order] = [min(x, y) for x, y in zip(cur, [f] + r)]
    L.set_option("attn_order", 0)
    a_, b_ = res[ORDERS[0]], res[ORDERS[1]]
    print(f"{name:46s} fwd {a_[0]:7.1f} -> {b_[0]:7.1f}   dQ {a_[1]:7.1f} -> {b_[1]:7.1f}   dK/dV {a_[2]:7.1f} -> {b_[2]:7.1f}   sum {sum(a_):7.1f} -> {sum(b_):7.1f} us ({100 * (sum(b_) / sum(a_) - 1):+.1f} %)")


case("encoder self N=1000, lengths U[0.7 N, N]", 32, 12, 1000, 1000, 0.7, True, False, False)
case("encoder self N=1000, no padding", 32, 12, 1000, 1000, 1.0, True, False, False)
case("encoder self N=1000, lengths U[0.4 N, N]", 32, 12, 1000, 1000, 0.4, True, False, False)
case("encoder self N=2000 H=16 (cfg-5), U[0.7 N, N]", 32, 16, 2000, 2000, 0.7, True, False, False)
case("decoder self N=256 causal", 32, 12, 256, 256, 1.0, True, True, False)
case("decoder cross 256 x 1100, memory U[0.7, 1]", 32, 12, 256, 1100, 0.7, False, False, True)
case("ViT N=100", 32, 12, 100, 100, 1.0, False, False, False)
case("B = 2 (goldens) N=1000 U[0.7 N, N]", 2, 12, 1000, 1000, 0.7, True, False, False)
