"""Grouped weight-gradient GEMM (v2s_gemm_grouped: the same projection of 12 layers in one launch, whole-K tiles) against 12 single
launches with split-K + reduce, on the decoder / ViT weight-gradient shapes of the cfg-2 step.  usage: python tools/gemm_group_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
SHAPES = [(768, 768, 8192, "dec o / cross q / cross o"), (2304, 768, 8192, "dec QKV"), (3072, 768, 8192, "dec wi"), (768, 3072, 8192, "dec wo"),
          (768, 768, 3200, "ViT proj"), (2304, 768, 3200, "ViT qkv"), (2048, 768, 3200, "ViT fc1"), (768, 2048, 3200, "ViT fc2")]
G = 12
def timed(f, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
tot = [0.0, 0.0]
for M, N, K, what in SHAPES:
    As = [(torch.randn(K, M, device=dev) * 0.5).to(torch.bfloat16) for _ in range(G)]
    Bs = [(torch.randn(K, N, device=dev) * 0.5).to(torch.bfloat16) for _ in range(G)]
    C1 = [torch.zeros(M, N, device=dev) for _ in range(G)]; C2 = [torch.zeros(M, N, device=dev) for _ in range(G)]
    def single():
        for a, b, c in zip(As, Bs, C1):
            L.gemm(a, b, c, M, N, K, transA=True, transB=True, workspace=ws)
    def grouped():
        L.gemm_grouped(As, Bs, C2, M, N, K)
    single(); grouped(); torch.cuda.synchronize()
    ref = [a.float().t() @ b.float() for a, b in zip(As[:2], Bs[:2])]
    err1 = max(float((c - r).abs().max() / r.abs().max()) for c, r in zip(C1, ref))
    err2 = max(float((c - r).abs().max() / r.abs().max()) for c, r in zip(C2, ref))
    same = all(torch.allclose(a, b, rtol=1e-4, atol=1e-3) for a, b in zip(C1, C2))
    t1, t2 = [], []
    for _ in range(5):
        t1.append(timed(single, 5)); t2.append(timed(grouped, 5))
    a, b = sorted(t1)[2], sorted(t2)[2]
    tot[0] += a; tot[1] += b
    fl = 2.0 * G * M * N * K
    print(f"{what:26s} {G} x {M}x{N}x{K}: single {a:7.1f} us ({fl / a / 1e6:4.0f} TF/s)  grouped {b:7.1f} us ({fl / b / 1e6:4.0f} TF/s)  x{a / b:.2f}   "
          f"rel err vs fp32 torch {err1:.1e} / {err2:.1e}" + ("" if same else "  MISMATCH"), flush=True)
print(f"sum single {tot[0]:.0f} us, grouped {tot[1]:.0f} us")
