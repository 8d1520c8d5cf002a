"""Same-process, interleaved A/B of the library's GEMM dispatch against the vendor library behind torch (hipBLASLt / rocBLAS through
torch.mm / torch.addmm) on the PLAIN GEMMs of the cfg-2 train step (no fused epilogue: the vendor call would need extra kernels for
ReLU / dropout / residual / dact, so those shapes are timed bare on both sides).  Test / measurement tool only: the product never calls
the vendor library.  usage: python tools/gemm_vendor_ab.py  ->  one row per shape, us and TF/s for both, ratio."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vidchapters_amd import lib as L

dev = "cuda"
SHAPES = [  # (kind, M, N, K, what)
    ("NT", 32000, 2304, 768, "enc QKV fwd"), ("NT", 32000, 768, 768, "enc O fwd"), ("NT", 32000, 3072, 768, "enc wi fwd"),
    ("NT", 32000, 768, 3072, "enc wo fwd"), ("NN", 32000, 768, 768, "enc O dgrad"), ("NN", 32000, 768, 2304, "enc QKV dgrad"),
    ("NN", 32000, 3072, 768, "enc wo dgrad"), ("NN", 32000, 768, 3072, "enc wi dgrad"), ("NT", 35200, 1536, 768, "cross K|V fwd"),
    ("NN", 35200, 768, 1536, "cross K|V dgrad"), ("NT", 8192, 2304, 768, "dec QKV fwd"), ("NT", 8192, 768, 768, "dec O fwd"),
    ("NT", 8192, 3072, 768, "dec wi fwd"), ("NT", 8192, 768, 3072, "dec wo fwd"), ("NN", 8192, 768, 2304, "dec QKV dgrad"),
    ("NN", 8192, 3072, 768, "dec wo dgrad"), ("NT", 2048, 32256, 768, "LM head chunk fwd (fp32 out)"), ("NN", 2048, 768, 32256, "LM head chunk dgrad (split-K fp32 + cast)"),
    ("NT", 3200, 2304, 768, "ViT QKV fwd"), ("NT", 3200, 2048, 768, "ViT fc1 fwd"), ("NT", 3200, 768, 2048, "ViT fc2 fwd"),
    ("TN", 768, 768, 32000, "enc O wgrad (fp32 out)"), ("TN", 2304, 768, 32000, "enc QKV wgrad"), ("TN", 3072, 768, 32000, "enc wi wgrad"),
    ("TN", 768, 3072, 32000, "enc wo wgrad"), ("TN", 1536, 768, 35200, "cross K|V wgrad"), ("TN", 2304, 768, 8192, "dec QKV wgrad"),
    ("TN", 32256, 768, 2048, "LM head chunk wgrad"),
    ("NT", 8192, 8192, 8192, "8192^3 (reference point)"),
]


# cfg-5 exact (t5-large: d = 1024, ff = 4096, 16 heads; B = 32, 200 frames, 2000 ASR tokens, 256 targets): `--large`
SHAPES_LARGE = [
    ("NT", 64000, 3072, 1024, "enc QKV fwd"), ("NT", 64000, 1024, 1024, "enc O fwd"), ("NT", 64000, 4096, 1024, "enc wi fwd"),
    ("NT", 64000, 1024, 4096, "enc wo fwd"), ("NN", 64000, 1024, 1024, "enc O dgrad"), ("NN", 64000, 1024, 3072, "enc QKV dgrad"),
    ("NN", 64000, 4096, 1024, "enc wo dgrad"), ("NN", 64000, 1024, 4096, "enc wi dgrad"), ("NT", 70400, 2048, 1024, "cross K|V fwd"),
    ("NN", 70400, 1024, 2048, "cross K|V dgrad"), ("NT", 8192, 3072, 1024, "dec QKV fwd"), ("NT", 8192, 1024, 1024, "dec O fwd"),
    ("NT", 8192, 4096, 1024, "dec wi fwd"), ("NT", 8192, 1024, 4096, "dec wo fwd"), ("NN", 8192, 1024, 3072, "dec QKV dgrad"),
    ("NN", 8192, 4096, 1024, "dec wo dgrad"), ("NN", 8192, 1024, 4096, "dec wi dgrad"),
    ("TN", 1024, 1024, 64000, "enc O wgrad (fp32 out)"), ("TN", 3072, 1024, 64000, "enc QKV wgrad"), ("TN", 4096, 1024, 64000, "enc wi wgrad"),
    ("TN", 1024, 4096, 64000, "enc wo wgrad"), ("TN", 2048, 1024, 70400, "cross K|V wgrad"), ("TN", 3072, 1024, 8192, "dec QKV wgrad"),
]
if "--large" in sys.argv:
    SHAPES = SHAPES_LARGE


def make(kind, M, N, K):
    g = torch.Generator(device=dev); g.manual_seed(M * 7 + N * 3 + K)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
    head_dgrad = kind == "NN" and K >= 32000          # the engine runs it split-K into fp32 + one cast to bf16 (Engine.t5_loss_forward)
    f32 = kind == "TN" or N >= 32000 or head_dgrad
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    if kind == "NT":
        A, B, kw = rn(M, K), rn(N, K), {}
        At, Bt = A, B.t()
    elif kind == "NN":
        A, B, kw = rn(M, K), rn(K, N), dict(transB=True)
        At, Bt = A, B
        if head_dgrad:
            kw["workspace"] = torch.empty(96 << 20, dtype=torch.uint8, device=dev)
    else:
        A, B, kw = rn(K, M), rn(K, N), dict(transA=True, transB=True)
        At, Bt = A.t(), B
        kw["workspace"] = torch.empty(96 << 20, dtype=torch.uint8, device=dev)
    Cv = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)       # the vendor side writes bf16 (torch.mm has no fp32-out bf16 GEMM)

    Cb = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if head_dgrad else None

    def ours():
        L.gemm(A, B, C, M, N, K, **kw)
        if head_dgrad:
            L.cast_bf16(C.view(-1), Cb.view(-1), M * N)

    def vendor():
        torch.mm(At, Bt, out=Cv)
    return ours, vendor, C, Cv


def timed(f, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    tot_o = tot_v = 0.0
    print(f"{'shape':42s} {'ours us':>9s} {'TF/s':>6s} {'kernel':28s} {'vendor us':>10s} {'TF/s':>6s}  vendor/ours")
    for kind, M, N, K, what in SHAPES:
        ours, vendor, C, Cv = make(kind, M, N, K)
        for _ in range(3):
            ours(); vendor()
        kern = L.lib().v2s_last_gemm_kernel().decode()
        torch.cuda.synchronize()
        err = float((C.float() - Cv.float()).abs().max() / (Cv.float().abs().max() + 1e-9))
        n = 20 if M * N * K > 4e12 else 50
        to, tv = [], []
        for _ in range(5):                      # interleaved rounds: drift hits both sides alike
            to.append(timed(ours, n)); tv.append(timed(vendor, n))
        o, v = sorted(to)[2], sorted(tv)[2]
        fl = 2.0 * M * N * K
        tot_o += o; tot_v += v
        print(f"{kind} {M:6d}x{N:6d}x{K:6d} {what:20s} {o:9.1f} {fl / o / 1e6:6.0f} {kern:28s} {v:10.1f} {fl / v / 1e6:6.0f}  {v / o:6.3f}   (max rel diff {err:.1e})")
    print(f"sum: ours {tot_o:.0f} us, vendor {tot_v:.0f} us, vendor/ours {tot_v / tot_o:.3f}")


if __name__ == "__main__":
    main()
