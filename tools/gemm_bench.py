"""Micro-benchmark of the GEMM kernel on the shapes of the Vid2Seq step (and a square reference shape)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L

dev = "cuda"
def bench(M, N, K, ta, tb, f32=False, acc=False, ws=None, iters=20):
    A = torch.randn((K, M) if ta else (M, K), device=dev).to(torch.bfloat16)
    B = torch.randn((K, N) if tb else (N, K), device=dev).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    kw = dict(transA=ta, transB=tb, accumulate=acc, workspace=ws)
    for _ in range(3): L.gemm(A, B, C, M, N, K, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): L.gemm(A, B, C, M, N, K, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2.0 * M * N * K / ms / 1e9, ms * 1e3

shapes = [("enc qkv fwd", 32000, 2304, 768, 0, 0), ("enc o fwd", 32000, 768, 768, 0, 0), ("enc wi fwd", 32000, 3072, 768, 0, 0),
          ("enc wo fwd", 32000, 768, 3072, 0, 0), ("square 4096", 4096, 4096, 4096, 0, 0), ("square 8192", 8192, 8192, 8192, 0, 0),
          ("enc qkv dgrad", 32000, 768, 2304, 0, 1), ("enc wi dgrad", 32000, 768, 3072, 0, 1), ("enc wo dgrad", 32000, 3072, 768, 0, 1),
          ("lm head fwd", 8192, 32200, 768, 0, 0)]
ws = torch.empty(80 * 1024 * 1024 // 4, device=dev)
CHECK_MODE = 3
def check(M, N, K, ta, tb):
    A = torch.randn((K, M) if ta else (M, K), device=dev).to(torch.bfloat16)
    B = torch.randn((K, N) if tb else (N, K), device=dev).to(torch.bfloat16)
    outs = []
    for big in (0, CHECK_MODE):
        L.set_option("gemm_big", big)
        C = torch.zeros(M, N, device=dev, dtype=torch.float32)
        L.gemm(A, B, C, M, N, K, transA=ta, transB=tb)
        outs.append(C)
    ref = (A.float().t() if ta else A.float()) @ (B.float() if tb else B.float().t())
    print(f"check M={M} N={N} K={K} ta={ta} tb={tb}: |w4-ref|max={float((outs[1]-ref).abs().max()):.3e} |old-ref|max={float((outs[0]-ref).abs().max()):.3e}")
for args in ((512, 256, 96, False, False), (512, 1024, 32, False, False), (768, 2048, 64, False, True), (1024, 1280, 352, True, True), (256, 128, 128, False, False), (300, 200, 64, False, False), (512, 384, 160, False, True), (256, 128, 32, True, True), (1000, 520, 224, True, True)):
    check(*args)

vit = [("vit qkv fwd", 3200, 2304, 768, 0, 0), ("vit fc1 fwd", 3200, 2048, 768, 0, 0), ("vit fc2 fwd", 3200, 768, 2048, 0, 0), ("vit proj fwd", 3200, 768, 768, 0, 0),
       ("vit qkv dgrad", 3200, 768, 2304, 0, 1), ("vit fc2 dgrad", 3200, 2048, 768, 0, 1), ("dec o fwd", 8192, 768, 768, 0, 0), ("dec o dgrad", 8192, 768, 768, 0, 1)]
vitw = [("vit wgrad qkv", 2304, 768, 3200), ("vit wgrad fc1", 2048, 768, 3200), ("vit wgrad fc2", 768, 2048, 3200), ("vit wgrad proj", 768, 768, 3200),
        ("dec wgrad o", 768, 768, 8192), ("dec wgrad wi", 3072, 768, 8192), ("dec wgrad qkv", 2304, 768, 8192), ("cross kv wgrad", 1536, 768, 35200)]
import sys as _s
if len(_s.argv) > 1 and _s.argv[1] == "wgrad":
    allw = [("wgrad qkv", 2304, 768, 32000), ("wgrad wi", 3072, 768, 32000), ("wgrad o", 768, 768, 32000), ("wgrad wo", 768, 3072, 32000)] + vitw + [("vit wgrad fc2b", 768, 2048, 3200)]
    L.set_option("gemm_order", int(_s.argv[2]) if len(_s.argv) > 2 else 0)
    for big, split in ((1, 1), (3, 1), (0, 1), (2, 1), (1, 1), (3, 1), (0, 1), (2, 1)):
        L.set_option("gemm_big", big); L.set_option("gemm_split", split)
        print(f"--- gemm_big={big} gemm_split={split}")
        for name, M, N, K in allw:
            tf, us = bench(M, N, K, True, True, f32=True, acc=True, ws=ws)
            print(f"{name:16s} M={M:6d} N={N:6d} K={K:6d}: {tf:7.1f} TF/s  {us:8.1f} us (split-K)")
    _s.exit(0)
if len(_s.argv) > 1 and _s.argv[1] == "order":
    allw = [("wgrad qkv", 2304, 768, 32000), ("wgrad wi", 3072, 768, 32000), ("wgrad o", 768, 768, 32000), ("wgrad wo", 768, 3072, 32000)] + vitw
    for order in (0, 4, 8, 0, 4, 8):
        L.set_option("gemm_order", order)
        print(f"--- gemm_order={order}")
        for name, M, N, K, ta, tb in shapes:
            tf, us = bench(M, N, K, bool(ta), bool(tb), f32=(name == "lm head fwd"))
            print(f"{name:16s} M={M:6d} N={N:6d} K={K:6d}: {tf:7.1f} TF/s  {us:8.1f} us")
        for name, M, N, K in allw:
            tf, us = bench(M, N, K, True, True, f32=True, acc=True, ws=ws)
            print(f"{name:16s} M={M:6d} N={N:6d} K={K:6d}: {tf:7.1f} TF/s  {us:8.1f} us (split-K)")
    _s.exit(0)
if len(_s.argv) > 1 and _s.argv[1] == "small":
    for big in (1, 3, 0):
        L.set_option("gemm_big", big)
        print(f"--- gemm_big={big}")
        for name, M, N, K, ta, tb in vit:
            tf, us = bench(M, N, K, bool(ta), bool(tb))
            print(f"{name:16s} M={M:6d} N={N:6d} K={K:6d}: {tf:7.1f} TF/s  {us:8.1f} us")
        for name, M, N, K in vitw:
            tf, us = bench(M, N, K, True, True, f32=True, acc=True, ws=ws)
            print(f"{name:16s} M={M:6d} N={N:6d} K={K:6d}: {tf:7.1f} TF/s  {us:8.1f} us (split-K)")
    _s.exit(0)

for big, pers in ((1, 0), (3, 0), (0, 0)):
    L.set_option("gemm_big", big)
    print(f"--- gemm_big={big} gemm_pers={pers}")
    for name, M, N, K, ta, tb in shapes:
        tf, us = bench(M, N, K, bool(ta), bool(tb), f32=(name == "lm head fwd"))
        print(f"{name:16s} M={M:6d} N={N:6d} K={K:6d}: {tf:7.1f} TF/s  {us:8.1f} us")
    for name, M, N, K in (("wgrad qkv", 2304, 768, 32000), ("wgrad wi", 3072, 768, 32000), ("wgrad o", 768, 768, 32000), ("wgrad wo", 768, 3072, 32000)):
        tf, us = bench(M, N, K, True, True, f32=True, acc=True, ws=ws)
        print(f"{name:16s} M={M:6d} N={N:6d} K={K:6d}: {tf:7.1f} TF/s  {us:8.1f} us (split-K)")
