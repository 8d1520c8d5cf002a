"""Micro-benchmark of the attention kernels on the encoder shape (B=32, H=12, L=1000) under option variants."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
B, H, N = 32, 12, 1000
W = H * 64
qkv = (torch.randn(B, N, 3 * W, device=dev) * 0.5).to(torch.bfloat16)
d_o = torch.randn(B, N, W, device=dev).to(torch.bfloat16)
o = torch.empty(B, N, W, dtype=torch.bfloat16, device=dev)
ml = torch.empty(B, H, N, 2, dtype=torch.float32, device=dev)
dqkv = torch.empty_like(qkv)
delta = torch.empty(B, H, N, 4, dtype=torch.float32, device=dev)
diag = torch.randn(H, 2 * N - 1, device=dev)
ddiag = torch.zeros(H, 2 * N - 1, device=dev)
lens = torch.randint(700, 1001, (B,), device=dev)
mask = (torch.arange(N, device=dev)[None, :] < lens[:, None]).to(torch.uint8).contiguous()
st = (N * 3 * W, 3 * W)

def run(bias, drop, dbias, masked, far):
    a = L.attn_args(B, H, N, N, qkv, qkv[..., W:], qkv[..., 2 * W:], o, st, st, st, (N * W, W), ml=ml, scale=1.0,
                    bias_diag=diag if bias else None, key_mask=mask if masked else None, dropout_p=0.1 if drop else 0.0, dropout_seed=5)
    def fwd(): L.attn_fwd(a)
    def bwd(): L.attn_bwd(a, d_o, (N * W, W), delta, dqkv, dqkv[..., W:], dqkv[..., 2 * W:], st, st, st,
                          dbias_diag=ddiag if (bias and dbias) else None, far=far)
    res = []
    for f in (fwd, bwd):
        for _ in range(2): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5 * 1e3)
    return res

for name, kw in [("plain", dict(bias=0, drop=0, dbias=0, masked=0, far=(0, 0))),
                 ("masked", dict(bias=0, drop=0, dbias=0, masked=1, far=(0, 0))),
                 ("bias", dict(bias=1, drop=0, dbias=0, masked=1, far=(0, 0))),
                 ("bias+dbias(far)", dict(bias=1, drop=0, dbias=1, masked=1, far=(-91, 91))),
                 ("bias+dbias(nofar)", dict(bias=1, drop=0, dbias=1, masked=1, far=(0, 0))),
                 ("bias+drop", dict(bias=1, drop=1, dbias=0, masked=1, far=(0, 0))),
                 ("bias+drop+dbias(far)", dict(bias=1, drop=1, dbias=1, masked=1, far=(-91, 91)))]:
    f, b = run(**kw)
    print(f"{name:24s} fwd {f:7.1f} us ({4*B*H*N*N*64/f/1e6:6.1f} TF/s)   bwd(delta+dq+dkv) {b:7.1f} us ({8*B*H*N*N*64/b/1e6:6.1f} TF/s)")
