"""A/B + bit-identity check of the 4-wave 128x128-wave-tile GEMM (option gemm_w128) against the default dispatch.
usage: python tools/gemm_w128_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
SHAPES = [("NT", 32000, 2304, 768), ("NT", 32000, 768, 768), ("NT", 32000, 3072, 768), ("NT", 32000, 768, 3072), ("NT", 35200, 1536, 768),
          ("NT", 8192, 2304, 768), ("NT", 8192, 3072, 768), ("NT", 2048, 32256, 768), ("NT", 1000, 520, 256), ("NT", 3200, 2304, 768), ("NT", 3200, 768, 2048), ("NT", 8192, 8192, 8192)]
VARIANTS = [("default", dict()), ("w128", dict(gemm_w128=2)), ("w128-nostore", dict(gemm_w128=2, gemm_dbg=1)), ("w128-ml", dict(gemm_w128=2, gemm_dbg=2))]
DEF = dict(gemm_w128=0, gemm_dbg=0)
def timed(f, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
bad = 0
for kind, M, N, K in SHAPES:
    torch.manual_seed(1)
    A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    B = (torch.randn(*((K, N) if kind == "NN" else (N, K)), device=dev) * 0.5).to(torch.bfloat16)
    kw = dict(transB=(kind == "NN"), ldb=N if kind == "NN" else K)
    outs, kern = {}, {}
    for name, opts in VARIANTS[:2]:
        for k, v in {**DEF, **opts}.items(): L.set_option(k, v)
        C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        L.gemm(A, B, C, M, N, K, **kw); torch.cuda.synchronize()
        outs[name] = C; kern[name] = L.lib().v2s_last_gemm_kernel().decode()
    same = torch.equal(outs["default"], outs["w128"]) and "gemm_wt" in kern["w128"]
    bad += 0 if same else 1
    res = {n: [] for n, _ in VARIANTS}
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    n_it = 10 if M * N * K > 2e11 else 20
    for rep in range(5):
        for name, opts in VARIANTS:
            for k, v in {**DEF, **opts}.items(): L.set_option(k, v)
            f = lambda: L.gemm(A, B, C, M, N, K, **kw)
            f(); res[name].append(timed(f, n_it))
    for k, v in DEF.items(): L.set_option(k, v)
    fl = 2.0 * M * N * K
    print(f"{kind} {M}x{N}x{K}: " + "  ".join(f"{n} {sorted(res[n])[2]:.1f} ({fl / sorted(res[n])[2] / 1e6:.0f})" for n, _ in VARIANTS)
          + f"   [{kern['default']} | {kern['w128']}]" + ("" if same else "  DIFFERENT"), flush=True)
print(f"{bad} shape(s) differ")
sys.exit(1 if bad else 0)
