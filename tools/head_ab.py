"""LM head + label-smoothed CE at the cfg-2 shape (8192 decoder rows, vocabulary 32200 padded to 32256, d_model 768): the round-2 chunked flow
(fp32 logits chunk -> v2s_ce_fwd -> v2s_ce_bwd) against the round-6 flow that never writes logits (v2s_lmhead_ce_fwd over all rows, v2s_lmhead_ce_bwd
per chunk), kernel by kernel (HIP events, 20 repetitions), without the two consumer GEMMs that both flows share.  usage: python tools/head_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
M, V, d, R = 8192, 32200, 768, 2048
Vp = (V + 63) // 64 * 64
torch.manual_seed(0)
h = torch.randn(M, d, device=dev).bfloat16()
E = torch.zeros(Vp, d, device=dev, dtype=torch.bfloat16); E[:V] = (torch.randn(V, d, device=dev) * 0.2).bfloat16()
labels = torch.randint(0, V, (M,), device=dev); labels[::9] = -100
alpha = d ** -0.5
row = torch.empty(M, 2, device=dev); acc = torch.zeros(2, device=dev); gs = torch.full((1,), 1e-4, device=dev)
lg = torch.empty(R, Vp, device=dev); dl = torch.empty(R, Vp, device=dev, dtype=torch.bfloat16)
part = torch.empty(L.lmhead_ce_workspace_floats(M, Vp), device=dev)
dh32 = torch.empty(R, d, device=dev); dhb = torch.empty(R, d, device=dev, dtype=torch.bfloat16); ws = torch.empty(16 * R * d, device=dev)
gE = torch.zeros(V, d, device=dev)
wsw = torch.empty(8 * 3072 * 768, device=dev)


def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


rows = []
rows.append(("old: logits GEMM fp32 out (per 2048-row chunk)", t(lambda: L.gemm(h[:R], E, lg, R, V, d, ldc=Vp, alpha=alpha)), 4))
rows.append(("old: v2s_ce_fwd (per chunk)", t(lambda: L.ce_fwd(lg, Vp, labels[:R], R, V, 0.1, row[:R], acc[0:1], acc[1:2])), 4))
rows.append(("old: v2s_ce_bwd (per chunk)", t(lambda: L.ce_bwd(lg, Vp, labels[:R], row[:R], R, V, 0.1, gs, dl, Vp)), 4))
rows.append(("old: d(hidden) GEMM split-K fp32 + cast (per chunk)", t(lambda: (L.gemm(dl, E, dh32, R, d, Vp, transB=True, lda=Vp, ldb=d, alpha=alpha, workspace=ws),
                                                                              L.cast_bf16(dh32.view(-1), dhb.view(-1), R * d))), 4))
rows.append(("new: v2s_lmhead_ce_fwd, all 8192 rows (stats GEMM + finish + reduce)", t(lambda: L.lmhead_ce_fwd(h, d, E, M, V, Vp, d, alpha, labels, 0.1, part, row, acc[0:1], acc[1:2])), 1))
rows.append(("new: v2s_lmhead_ce_bwd (per chunk)", t(lambda: L.lmhead_ce_bwd(h[:R], d, E, R, V, Vp, d, alpha, labels[:R], row[:R], 0.1, gs, dl, Vp)), 4))
rows.append(("new: d(hidden) GEMM split-K -> bf16 (per chunk)", t(lambda: L.gemm(dl, E, dhb, R, d, Vp, transB=True, lda=Vp, ldb=d, alpha=alpha, workspace=ws)), 4))
rows.append(("both: d(E) += d(logits)^T h (per chunk)", t(lambda: L.gemm(dl, h[:R], gE, V, d, R, transA=True, transB=True, lda=Vp, ldb=d, ldc=d, accumulate=True, alpha=alpha, workspace=wsw)), 4))
tot = {"old": 0.0, "new": 0.0}
for n, us, k in rows:
    print(f"{n:78s} {us:8.1f} us x {k} = {us * k / 1e3:6.3f} ms   ({2.0 * (M if k == 1 else R) * Vp * d / us / 1e6:6.0f} TF/s as a GEMM)")
    for key in tot:
        if n.startswith(key) or n.startswith("both"):
            tot[key] += us * k / 1e3
print(f"head forward + backward per step: old {tot['old']:.3f} ms, new {tot['new']:.3f} ms")
