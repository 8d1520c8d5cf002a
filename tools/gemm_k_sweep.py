"""Fixed per-launch cost of the 128x128 LDS-DMA GEMM: time vs K at M = 32000, N = 768 (NT and NN), back to back launches.
The intercept of the linear fit is what a K = 768 launch pays besides its K-steps (launch, first stage, epilogues, tail, output burst).
usage: python tools/gemm_k_sweep.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vidchapters_amd import lib as L
dev = "cuda"
M, N = 32000, 768
Ks = [256, 512, 768, 1536, 2304, 3072]
def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
dbg = int(sys.argv[1]) if len(sys.argv) > 1 else 0      # 1: no global store, 2: no epilogue at all (ablations: results invalid)
L.set_option("gemm_dbg", dbg)
print("gemm_dbg =", dbg)
for tb, name in ((False, "NT"), (True, "NN")):
    ts = []
    for K in Ks:
        A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        B = (torch.randn(*( (K, N) if tb else (N, K)), device=dev) * 0.5).to(torch.bfloat16)
        C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        best = min(t(lambda: L.gemm(A, B, C, M, N, K, transB=tb, ldb=N if tb else K)) for _ in range(3))
        ts.append(best)
    slope, icpt = np.polyfit(Ks, ts, 1)
    print(name, " ".join(f"K={k}: {x:.1f} us ({2 * M * N * k / x / 1e6:.0f} TF/s)" for k, x in zip(Ks, ts)))
    print(f"   fit: {icpt:.1f} us + {slope * 768:.1f} us per 768 of K  ->  main loop alone {2 * M * N * 768 / (slope * 768) / 1e6:.0f} TF/s; kernel {L.lib().v2s_last_gemm_kernel().decode()}")
