"""Interleaved A/B of the library's attention kernels against torch's fused attention (scaled_dot_product_attention: the flash /
memory-efficient backends shipped with PyTorch-ROCm) at the encoder self-attention shape of the cfg-2 step (B = 32, 12 heads, 1100
keys, head_dim 64).  Measurement only: the product never calls it.  Rows: ours plain (no bias / mask / dropout), ours as the step runs
it (T5 bias + key mask + dropout 0.1), vendor plain, vendor with an additive [H, N, N] bias; forward, and forward+backward."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from vidchapters_amd import lib as L

dev = "cuda"
B, H, N = 32, 12, 1100
W = H * 64
torch.manual_seed(0)
q = (torch.randn(B, N, W, device=dev) * 0.5).to(torch.bfloat16); k = (torch.randn(B, N, W, device=dev) * 0.5).to(torch.bfloat16)
v = torch.randn(B, N, W, device=dev).to(torch.bfloat16); d_o = torch.randn(B, N, W, device=dev).to(torch.bfloat16)
o = torch.empty_like(q); ml = torch.empty(B, H, N, 2, device=dev); delta = torch.empty(B, H, N, 4, device=dev)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
diag = torch.randn(H, 2 * N - 1, device=dev)
ddiag = torch.zeros(H, 2 * N - 1, device=dev)
lens = torch.randint(int(0.7 * N), N + 1, (B,), device=dev)
mask = (torch.arange(N, device=dev)[None, :] < lens[:, None]).to(torch.uint8).contiguous()
sq = (N * W, W)


def ours(full):
    kw = dict(ml=ml, scale=1.0, bias_diag=diag if full else None, key_mask=mask if full else None, causal=False,
              dropout_p=0.1 if full else 0.0, dropout_seed=5)

    def fwd():
        L.attn_fwd(L.attn_args(B, H, N, N, q, k, v, o, sq, sq, sq, sq, **kw))

    def fb():
        a = L.attn_args(B, H, N, N, q, k, v, o, sq, sq, sq, sq, **kw)
        L.attn_fwd(a)
        L.attn_bwd(a, d_o, sq, delta, dq, dk, dv, sq, sq, sq, dbias_diag=ddiag if full else None, far=(-91, 91) if full else (0, 0))
    return fwd, fb


qh = q.view(B, N, H, 64).transpose(1, 2).detach().requires_grad_(True)      # [B, H, N, 64] views, no copy
kh = k.view(B, N, H, 64).transpose(1, 2).detach().requires_grad_(True)
vh = v.view(B, N, H, 64).transpose(1, 2).detach().requires_grad_(True)
doh = d_o.view(B, N, H, 64).transpose(1, 2)
idx = torch.arange(N, device=dev)
bias_full = diag[:, (idx[None, :] - idx[:, None]) + N - 1].to(torch.bfloat16).unsqueeze(0).contiguous()          # [1, H, N, N]


def vendor(with_bias, drop):
    kw = dict(attn_mask=bias_full if with_bias else None, dropout_p=0.1 if drop else 0.0, scale=1.0)

    def fwd():
        with torch.no_grad():
            F.scaled_dot_product_attention(qh, kh, vh, **kw)

    def fb():
        out = F.scaled_dot_product_attention(qh, kh, vh, **kw)
        out.backward(doh)
        qh.grad = kh.grad = vh.grad = None
    return fwd, fb


def timed(f, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


variants = {"ours plain": ours(False), "ours bias+mask+dropout (as in the step)": ours(True), "vendor plain": vendor(False, False),
            "vendor dropout": vendor(False, True), "vendor additive bias": vendor(True, False), "vendor bias+dropout": vendor(True, True)}
ok = {}
for name, (f, fb) in list(variants.items()):
    try:
        f(); fb(); torch.cuda.synchronize(); ok[name] = (f, fb)
    except Exception as e:      # a backend may refuse a combination
        print(f"{name}: not available ({type(e).__name__}: {str(e)[:100]})")
res = {n: ([], []) for n in ok}
for _ in range(4):
    for n, (f, fb) in ok.items():
        res[n][0].append(timed(f)); res[n][1].append(timed(fb))
fl = 4.0 * B * H * N * N * 64
for n in ok:
    f, fb = sorted(res[n][0])[1], sorted(res[n][1])[1]
    print(f"{n:42s} fwd {f:8.1f} us {fl / f / 1e6:6.0f} TF/s   fwd+bwd {fb:8.1f} us {3.5 * fl / fb / 1e6:6.0f} TF/s (3.5 x fwd flop)")
