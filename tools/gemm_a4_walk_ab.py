"""Tile walk of the persistent gemm_a4p kernel: row-major (gemm_a4_walk = 1) against groups of 2 / 4 / 8 tile rows, interleaved in one process, against
the vendor library behind torch.mm, on shapes with many tile columns.  usage: python tools/gemm_a4_walk_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vidchapters_amd import lib as L

dev = "cuda"
SHAPES = [("NT", 8192, 8192, 8192), ("NT", 16384, 8192, 2048), ("NT", 64000, 4096, 1024), ("NT", 64000, 3072, 1024), ("NN", 64000, 4096, 1024), ("NT", 32000, 3072, 768),
          ("NT", 32000, 2304, 768), ("NT", 2048, 32256, 768)]
WALKS = (1, 2, 4, 8)


def timed(f, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


L.set_option("gemm_a4", 2)
for kind, M, N, K in SHAPES:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = (torch.randn(K, N, device=dev) if kind == "NN" else torch.randn(N, K, device=dev)).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16); Cv = torch.empty_like(C)
    Bt = B if kind == "NN" else B.t()
    res, outs = {w: [] for w in WALKS}, {}
    res["vendor"] = []
    n_it = 5 if M * N * K > 2e11 else 15
    for rep in range(5):
        for w in WALKS:
            L.set_option("gemm_a4_walk", w)
            f = lambda: L.gemm(A, B, C, M, N, K, transB=(kind == "NN"), ldb=N if kind == "NN" else K)
            f(); outs[w] = C.clone(); kern = L.lib().v2s_last_gemm_kernel().decode()
            res[w].append(timed(f, n_it))
        fv = lambda: torch.mm(A, Bt, out=Cv)
        fv(); res["vendor"].append(timed(fv, n_it))
    L.set_option("gemm_a4_walk", 0)
    fl = 2.0 * M * N * K
    med = {k: sorted(v)[2] for k, v in res.items()}
    same = all(torch.equal(outs[w], outs[1]) for w in WALKS)
    print(f"{kind} {M}x{N}x{K} [{kern}] " + "  ".join(f"{'GM ' + str(k) if k != 'vendor' else k} {med[k]:7.1f} us ({fl / med[k] / 1e6:5.0f})" for k in med) + f"  identical {same}", flush=True)
L.set_option("gemm_a4", 1)
