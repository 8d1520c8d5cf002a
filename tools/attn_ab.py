"""Same-process A/B of attention kernels from two builds of the library (box-to-box variance is +-5-15 %, so variants are only
comparable when interleaved on one GPU).  usage: python tools/attn_ab.py libA.so libB.so"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
paths = sys.argv[1:] or [L.LIB_PATH]

def use(path):
    L.LIB_PATH = os.path.abspath(path); L._LIB = None; L.lib()

def make(B, H, Nq, Nk, bias, masked, causal, drop, dbias):
    W = H * 64
    q = (torch.randn(B, Nq, W, device=dev) * 0.5).to(torch.bfloat16); k = (torch.randn(B, Nk, W, device=dev) * 0.5).to(torch.bfloat16)
    v = torch.randn(B, Nk, W, device=dev).to(torch.bfloat16); d_o = torch.randn(B, Nq, W, device=dev).to(torch.bfloat16)
    o = torch.empty_like(q); ml = torch.empty(B, H, Nq, 2, device=dev); delta = torch.empty(B, H, Nq, 4, device=dev)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    diag = torch.randn(H, Nq + Nk - 1, device=dev) if bias else None
    ddiag = torch.zeros(H, Nq + Nk - 1, device=dev) if (bias and dbias) else None
    mask = None
    if masked:
        lens = torch.randint(int(0.7 * Nk), Nk + 1, (B,), device=dev)
        mask = (torch.arange(Nk, device=dev)[None, :] < lens[:, None]).to(torch.uint8).contiguous()
        if os.environ.get("ZROWS") == "1" and Nq == Nk:      # dense self-attention: the pad rows' gradient is exactly zero (as in the train step)
            d_o = torch.where((torch.arange(Nq, device=dev)[None, :] < lens[:, None])[..., None], d_o, torch.zeros_like(d_o)).contiguous()
    sq, sk = (Nq * W, W), (Nk * W, W)
    def fwd():
        a = L.attn_args(B, H, Nq, Nk, q, k, v, o, sq, sk, sk, sq, ml=ml, scale=1.0, bias_diag=diag, key_mask=mask, causal=causal,
                        dropout_p=0.1 if drop else 0.0, dropout_seed=5)
        L.attn_fwd(a); return a
    def bwd():
        a = L.attn_args(B, H, Nq, Nk, q, k, v, o, sq, sk, sk, sq, ml=ml, scale=1.0, bias_diag=diag, key_mask=mask, causal=causal,
                        dropout_p=0.1 if drop else 0.0, dropout_seed=5)
        if "r2attn" in L.LIB_PATH:      # the round-2 kernels need delta = rowsum(dO * O) from a separate launch first
            import ctypes as C
            a.d_o = d_o.data_ptr(); a.do_bs, a.do_rs = sq
            L._check(L.lib().v2s_attn_delta(C.byref(a), delta.data_ptr(), L.stream_ptr()), "v2s_attn_delta")
        L.attn_bwd(a, d_o, sq, delta, dq, dk, dv, sq, sk, sk, dbias_diag=ddiag, far=(-91, 91) if ddiag is not None else (0, 0))
    return fwd, bwd, 4.0 * B * H * Nq * Nk * 64

def t(f, n=10):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

CASES = [("encoder self  B32 N1000 bias mask drop dbias", dict(B=32, H=12, Nq=1000, Nk=1000, bias=1, masked=1, causal=0, drop=1, dbias=1)),
         ("encoder self  B32 N1100 bias mask drop dbias", dict(B=32, H=12, Nq=1100, Nk=1100, bias=1, masked=1, causal=0, drop=1, dbias=1)),
         ("ViT           B32 N100  drop", dict(B=32, H=12, Nq=100, Nk=100, bias=0, masked=0, causal=0, drop=1, dbias=0)),
         ("decoder self  B32 N256  bias causal drop dbias", dict(B=32, H=12, Nq=256, Nk=256, bias=1, masked=0, causal=1, drop=1, dbias=1)),
         ("decoder cross B32 256x1100 mask drop", dict(B=32, H=12, Nq=256, Nk=1100, bias=0, masked=1, causal=0, drop=1, dbias=0)),
         ("encoder self  no dropout (eval)", dict(B=32, H=12, Nq=1000, Nk=1000, bias=1, masked=1, causal=0, drop=0, dbias=0))]
for name, kw in CASES:
    torch.manual_seed(0)
    use(paths[0])
    fwd, bwd, fl = make(**kw)
    res = {p: [[], []] for p in paths}
    for rep in range(3):
        for p in paths:
            use(p)
            res[p][0].append(t(fwd)); res[p][1].append(t(bwd))
    line = f"{name:48s}"
    for p in paths:
        f, b = min(res[p][0]), min(res[p][1])
        line += f" | {os.path.basename(p)[:14]:14s} fwd {f:7.1f} us {fl / f / 1e6:6.1f} TF/s  bwd {b:7.1f} us {2 * fl / b / 1e6:6.1f} TF/s"
    print(line)
