"""Encoder-layer attention (B=32, H=12, N=1000, bias + mask + dropout + dbias): forward / dQ / dK|dV kernel times per library build,
interleaved in one process.  usage: python tools/attn_parts_ab.py libA.so libB.so ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
paths = sys.argv[1:] or [L.LIB_PATH]
N = int(os.environ.get("ATTN_N", "1000"))
DB = int(os.environ.get("ATTN_DBIAS", "1"))          # 0: the dQ kernel without the bias-gradient accumulation and its atomics
B, H = 32, 12
W = H * 64
torch.manual_seed(0)
qkv = (torch.randn(B, N, 3 * W, device=dev) * 0.5).to(torch.bfloat16); d_o = torch.randn(B, N, W, device=dev).to(torch.bfloat16)
o = torch.empty(B, N, W, dtype=torch.bfloat16, device=dev); ml = torch.empty(B, H, N, 2, device=dev)
dqkv = torch.empty_like(qkv); delta = torch.empty(B, H, N, 4, device=dev)
diag = torch.randn(H, 2 * N - 1, device=dev); ddiag = torch.zeros(H, 2 * N - 1, device=dev)
lens = torch.randint(int(0.7 * N), N + 1, (B,), device=dev)
mask = (torch.arange(N, device=dev)[None, :] < lens[:, None]).to(torch.uint8).contiguous()
st = (N * 3 * W, 3 * W)


def use(path):
    L.LIB_PATH = os.path.abspath(path); L._LIB = None; L.lib()


def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


res = {p: [[], [], []] for p in paths}
for rep in range(3):
    for p in paths:
        use(p)
        a = L.attn_args(B, H, N, N, qkv, qkv[..., W:], qkv[..., 2 * W:], o, st, st, st, (N * W, W), ml=ml, scale=1.0, bias_diag=diag, key_mask=mask,
                        dropout_p=0.1, dropout_seed=5)
        res[p][0].append(t(lambda: L.attn_fwd(a)))
        for part in (1, 2):
            L.set_option("attn_bwd_part", part)
            res[p][part].append(t(lambda: L.attn_bwd(a, d_o, (N * W, W), delta, dqkv, dqkv[..., W:], dqkv[..., 2 * W:], st, st, st, dbias_diag=(ddiag if DB else None), far=((-91, 91) if DB else (0, 0)))))
        L.set_option("attn_bwd_part", 0)
for p in paths:
    f, q, kv = (min(x) for x in res[p])
    print(f"{os.path.basename(p):40s} fwd {f:7.1f} us   dQ {q:7.1f} us   dK/dV {kv:7.1f} us   bwd sum {q + kv:7.1f} us")
