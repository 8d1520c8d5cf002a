"""Correctness + same-process interleaved A/B of the persistent write-out-wave GEMM (gemm_ps_kernel, option gemm_ps) against the
dispatch without it, on the forward / dgrad shapes of the cfg-2 train step WITH the epilogues the engine gives them.  Outputs must be
bit-identical (same K order, same fp32 epilogue arithmetic); exits 1 on any difference.
usage: python tools/gemm_ps_ab.py [--quick] [--large]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vidchapters_amd import lib as L

dev = "cuda"
# (name, kind, M, N, K, epilogue)
CASES = [
    ("edge ragged rows/cols", "NT", 1000, 200, 64, "res"), ("edge ragged dgrad", "NN", 777, 136, 192, "dact"),
    ("edge 1 tile fp32", "NT", 100, 96, 128, "f32"), ("edge bias+gelu+pre", "NT", 3200, 2048, 768, "gelu"),
    ("enc QKV fwd", "NT", 32000, 2304, 768, ""), ("enc O fwd res+drop", "NT", 32000, 768, 768, "res"),
    ("enc wi fwd relu+drop", "NT", 32000, 3072, 768, "act"), ("enc wo fwd res+drop", "NT", 32000, 768, 3072, "res"),
    ("enc O dgrad", "NN", 32000, 768, 768, ""), ("enc QKV dgrad", "NN", 32000, 768, 2304, ""),
    ("enc wo dgrad relu-mask+drop", "NN", 32000, 3072, 768, "dact"), ("enc wi dgrad", "NN", 32000, 768, 3072, ""),
    ("cross K|V fwd", "NT", 35200, 1536, 768, ""), ("cross K|V dgrad", "NN", 35200, 768, 1536, ""),
    ("cross K|V x12 fwd", "NT", 35200, 18432, 768, ""), ("cross K|V x12 dgrad", "NN", 35200, 768, 18432, ""),
    ("dec QKV fwd", "NT", 8192, 2304, 768, ""), ("dec O fwd res+drop", "NT", 8192, 768, 768, "res"),
    ("dec wi fwd relu+drop", "NT", 8192, 3072, 768, "act"), ("dec wo fwd res+drop", "NT", 8192, 768, 3072, "res"),
    ("dec QKV dgrad", "NN", 8192, 768, 2304, ""), ("dec wo dgrad relu-mask+drop", "NN", 8192, 3072, 768, "dact"),
    ("dec wi dgrad", "NN", 8192, 768, 3072, ""),
    ("LM head chunk fwd fp32", "NT", 2048, 32256, 768, "f32"),
    ("ViT QKV fwd bias", "NT", 3200, 2304, 768, "bias"), ("ViT fc2 fwd bias+res", "NT", 3200, 768, 2048, "biasres"),
    ("ViT fc1 dgrad gelu'", "NN", 3200, 2048, 768, "dgelu"),
]
LARGE = [
    ("L enc QKV fwd", "NT", 64000, 3072, 1024, ""), ("L enc O fwd res+drop", "NT", 64000, 1024, 1024, "res"),
    ("L enc wi fwd relu+drop", "NT", 64000, 4096, 1024, "act"), ("L enc wo fwd res+drop", "NT", 64000, 1024, 4096, "res"),
    ("L enc O dgrad", "NN", 64000, 1024, 1024, ""), ("L enc QKV dgrad", "NN", 64000, 1024, 3072, ""),
    ("L enc wo dgrad relu-mask+drop", "NN", 64000, 4096, 1024, "dact"), ("L enc wi dgrad", "NN", 64000, 1024, 4096, ""),
]
if "--large" in sys.argv:
    CASES = CASES[:4] + LARGE
if "--quick" in sys.argv:
    CASES = CASES[:9]
NST = 4
for a_ in sys.argv:
    if a_.startswith("--nst="):
        NST = int(a_[6:])


def timed(f, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    bad = 0
    tot = {0: 0.0, 1: 0.0}
    print(f"{'case':30s} {'shape':24s} {'old us':>8s} {'TF/s':>6s} {'old kernel':34s} {'ps us':>8s} {'TF/s':>6s}  old/ps")
    for name, kind, M, N, K, ep in CASES:
        torch.manual_seed(M + 3 * N + 7 * K)
        A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        B = (torch.randn(*((K, N) if kind == "NN" else (N, K)), device=dev) * 0.5).to(torch.bfloat16)
        Z = torch.relu(torch.randn(M, N, device=dev)).to(torch.bfloat16) if ep in ("dact", "dgelu") else None
        R = torch.randn(M, N, device=dev).to(torch.bfloat16) if ep in ("res", "biasres") else None
        bias = torch.randn(N, device=dev) if ep in ("bias", "biasres", "gelu") else None
        kw = dict(transB=(kind == "NN"), ldb=N if kind == "NN" else K)
        if ep == "act": kw.update(act=L.ACT_RELU, dropout_p=0.1, dropout_seed=3)
        if ep == "dact": kw.update(dact=L.ACT_RELU, z=Z, dropout_p=0.1, dropout_seed=3)
        if ep == "dgelu": kw.update(dact=L.ACT_GELU, z=Z)
        if ep == "res": kw.update(residual=R, dropout_p=0.1, dropout_seed=3)
        if ep == "bias": kw.update(bias=bias)
        if ep == "biasres": kw.update(bias=bias, residual=R)
        odt = torch.float32 if ep == "f32" else torch.bfloat16
        outs, kern, fns = {}, {}, {}
        L.set_option("gemm_ps_nst", NST)
        for mode in (0, 2):
            L.set_option("gemm_ps", mode)
            Cc = torch.full((M, N), float("nan"), dtype=odt, device=dev)
            kw2 = dict(kw)
            if ep == "gelu":
                kw2.update(bias=bias, act=L.ACT_GELU, pre=torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev))
            L.gemm(A, B, Cc, M, N, K, **kw2)
            torch.cuda.synchronize()
            kern[mode] = L.lib().v2s_last_gemm_kernel().decode()
            outs[mode] = (Cc, kw2.get("pre"))
            fns[mode] = (lambda m=mode, C_=Cc, k_=kw2: (L.set_option("gemm_ps", m), L.gemm(A, B, C_, M, N, K, **k_)))
        same = torch.equal(outs[0][0], outs[2][0]) and not bool(torch.isnan(outs[2][0].float()).any())
        if outs[0][1] is not None:
            same = same and torch.equal(outs[0][1], outs[2][1])
        if "gemm_ps_kernel" not in kern[2]:
            same = False
        n = 10 if M * N * K > 2e11 else 30
        t = {0: [], 2: []}
        for _ in range(5):
            for mode in (0, 2):
                fns[mode](); t[mode].append(timed(fns[mode], n))
        o, s = sorted(t[0])[2], sorted(t[2])[2]
        fl = 2.0 * M * N * K
        if M >= 3200:
            tot[0] += o; tot[1] += s
        print(f"{name:30s} {kind} {M:6d}x{N:6d}x{K:6d} {o:8.1f} {fl / o / 1e6:6.0f} {kern[0]:34s} {s:8.1f} {fl / s / 1e6:6.0f}  {o / s:6.3f}"
              + ("" if same else f"   DIFFERENT ({kern[2]})"), flush=True)
        bad += 0 if same else 1
    L.set_option("gemm_ps", 0)
    print(f"sum (M >= 3200): old {tot[0]:.0f} us, ps {tot[1]:.0f} us, old/ps {tot[0] / max(tot[1], 1e-9):.3f};  {bad} case(s) differ")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
