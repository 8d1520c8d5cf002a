"""Grouped cross-attention of a beam step (kv_group = beams of an entry against the entry's encoder K/V): packed-FMA kernel vs the
MFMA kernel at 4 / 8 / 16 waves per block (option gemm_skinny = 3 | 5 | 6 | 7 as A/B hook).  usage: python tools/decode_xattn_ab.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
H, W = 12, 768
def t(f, n=50):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, nb, S in ((16, 4, 1100), (64, 4, 1100), (16, 8, 1100), (16, 12, 1100), (4, 4, 1100), (16, 4, 2200)):
    R = B * nb
    g = torch.Generator(device=dev); g.manual_seed(1)
    q = (torch.randn(R, W, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    kvs = [(torch.randn(B, S, 2 * W, device=dev, generator=g) * 0.5).to(torch.bfloat16) for _ in range(12)]    # one K|V per decoder layer:
    kv = kvs[0]                                                                                               # 12 x 47 MB do not stay in the 256 MB L3
    lens = torch.randint(int(0.72 * S), S + 1, (B,), device=dev, generator=g)
    mask = (torch.arange(S, device=dev)[None] < lens[:, None]).to(torch.uint8).contiguous()
    outs = {}
    line = f"B={B} beams={nb} S={S} ({int(lens.sum()) * 2 * W * 2 / 1e6:.0f} MB of valid K/V): "
    for mode, name in ((3, "fma"), (5, "mfma4"), (6, "mfma8"), (7, "mfma16"), (1, "default")):
        L.set_option("gemm_skinny", mode)
        o = torch.empty(R, W, dtype=torch.bfloat16, device=dev)
        def f():
            for kv_ in kvs:
                L.decode_attn(R, H, S, q, W, kv_, kv_[:, :, W:], S * 2 * W, 2 * W, o, W, key_mask=mask, mask_ld=S, kv_group=nb)
        us = min(t(f, 10) for _ in range(3)) / 12
        L.decode_attn(R, H, S, q, W, kv, kv[:, :, W:], S * 2 * W, 2 * W, o, W, key_mask=mask, mask_ld=S, kv_group=nb)
        outs[name] = o.float()
        line += f"{name} {us:6.1f} us  "
    L.set_option("gemm_skinny", 1)
    err = max(float((outs[k] - outs["fma"]).abs().max()) for k in outs)
    print(line + f"| max |diff| vs fma {err:.4f}")
