"""Interleaved A/B of dispatch options on the weight-gradient (TN, split-K, fp32 accumulate) shapes of the cfg-2 step.
usage: python tools/gemm_wgrad_ab.py "gemm_p8=1" "gemm_p8=2" "gemm_p8=3" ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
SHAPES = [(768, 768, 32000, "enc O"), (2304, 768, 32000, "enc QKV"), (3072, 768, 32000, "enc wi"), (768, 3072, 32000, "enc wo"), (1536, 768, 35200, "cross K|V"),
          (768, 768, 8192, "dec O/q"), (2304, 768, 8192, "dec QKV"), (3072, 768, 8192, "dec wi"), (768, 3072, 8192, "dec wo"),
          (2304, 768, 3200, "ViT qkv"), (2048, 768, 3200, "ViT fc1"), (768, 2048, 3200, "ViT fc2"), (32256, 768, 2048, "LM head chunk")]
settings = sys.argv[1:] or ["gemm_p8=1", "gemm_p8=2", "gemm_p8=3"]
DEF = dict(gemm_p8=1, gemm_big=1, gemm_split=1)
def apply(s):
    for k, v in DEF.items(): L.set_option(k, v)
    for kv in filter(None, s.split(",")):
        k, v = kv.split("="); L.set_option(k, int(v))
def timed(f, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
tot = {s: 0.0 for s in settings}
for M, N, K, what in SHAPES:
    A = (torch.randn(K, M, device=dev) * 0.5).to(torch.bfloat16); B = (torch.randn(K, N, device=dev) * 0.5).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev)
    res, kern, outs = {s: [] for s in settings}, {}, {}
    f = lambda: L.gemm(A, B, C, M, N, K, transA=True, transB=True, workspace=ws)
    for s in settings:
        apply(s); C.zero_(); f(); torch.cuda.synchronize(); outs[s] = C.clone(); kern[s] = L.lib().v2s_last_gemm_kernel().decode()
    for rep in range(5):
        for s in settings:
            apply(s); f(); res[s].append(timed(f, 20))
    fl = 2.0 * M * N * K
    ok = all(torch.allclose(outs[s], outs[settings[0]], rtol=1e-3, atol=1e-2) for s in settings)
    print(f"{what:14s} {M:6d}x{N:5d}x{K:6d} " + "  ".join(f"[{s}] {sorted(res[s])[2]:7.1f} us {fl / sorted(res[s])[2] / 1e6:5.0f} TF/s {kern[s].replace('gemm_', '').replace('_kernel', '')}" for s in settings)
          + ("" if ok else "  MISMATCH"), flush=True)
    for s in settings: tot[s] += sorted(res[s])[2]
apply("")
print("sum: " + "  ".join(f"[{s}] {tot[s]:.0f} us" for s in settings))
