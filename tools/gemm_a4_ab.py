"""Correctness + same-process interleaved A/B of gemm_a4_kernel (option gemm_a4) against the default dispatch and the vendor library
behind torch.mm (measurement only) on the forward / dgrad GEMM shapes of the cfg-2 train step.
usage: python tools/gemm_a4_ab.py [--check-only] [--quick] [--large]   ->  one row per shape: us (TF/s) per variant"""
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
    ("NN", 8192, 3072, 768, "dec wo dgrad"), ("NT", 2048, 32256, 768, "LM head chunk fwd"),
    ("NT", 3200, 2304, 768, "ViT QKV fwd"), ("NT", 3200, 2048, 768, "ViT fc1 fwd"), ("NT", 3200, 768, 2048, "ViT fc2 fwd"),
    ("NT", 8192, 8192, 8192, "8192^3"),
]
if "--large" in sys.argv:
    SHAPES = [("NT", 64000, 3072, 1024, "enc QKV fwd"), ("NT", 64000, 1024, 1024, "enc O fwd"), ("NT", 64000, 4096, 1024, "enc wi fwd"),
              ("NT", 64000, 1024, 4096, "enc wo fwd"), ("NN", 64000, 1024, 3072, "enc QKV dgrad"), ("NN", 64000, 4096, 1024, "enc wo dgrad")]
if "--quick" in sys.argv:
    SHAPES = SHAPES[:4] + SHAPES[5:7] + SHAPES[8:10]
VARIANTS = [("default", dict()), ("a4", dict(gemm_a4=2)), ("a4-nostore", dict(gemm_a4=2, gemm_dbg=1)), ("a4v1", dict(gemm_a4=3)), ("a4v1-mainloop", dict(gemm_a4=3, gemm_dbg=2))]
# gemm_a4 = 2: the persistent deferred-write-out kernel where it is legal (plain bf16 epilogue, whole tiles), else the one-tile kernel; 3: one-tile kernel only
DEF = dict(gemm_a4=0, gemm_dbg=0)


def setopts(o):
    for k, v in {**DEF, **o}.items():
        L.set_option(k, v)


def relerr(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def check():
    """a4 against fp32 torch on bf16-rounded inputs, plain and fused epilogues, ragged edges, both operand layouts"""
    cases = [("NT", 256, 256, 128, ""), ("NT", 512, 512, 256, ""), ("NT", 1000, 520, 384, ""), ("NT", 2000, 768, 768, "res"), ("NT", 1300, 1544, 384, "act"),
             ("NT", 600, 520, 256, "f32"), ("NT", 3200, 768, 2048, "bias"), ("NN", 256, 256, 128, ""), ("NN", 777, 392, 256, "dact"), ("NN", 2048, 768, 2304, ""),
             ("NN", 1000, 520, 384, "res"), ("NT", 32000, 2304, 768, ""), ("NN", 32000, 768, 2304, ""), ("NT", 35200, 1536, 768, ""),
             ("NT", 512, 512, 384, ""), ("NN", 768, 512, 512, ""), ("NT", 8192, 768, 768, ""), ("NN", 32000, 3072, 768, ""), ("NT", 66048, 512, 384, ""), ("NT", 512, 66048, 384, ""), ("NT", 66048, 256, 384, "")]
    bad = 0
    for kind, M, N, K, ep in cases:
        g = torch.Generator(device=dev); g.manual_seed(M * 7 + N * 3 + K)
        rn = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).to(torch.bfloat16)
        A = rn(M, K)
        B = rn(K, N) if kind == "NN" else rn(N, K)
        kw = dict(transB=(kind == "NN"), ldb=N if kind == "NN" else K)
        ref = A.float() @ (B.float() if kind == "NN" else B.float().t())
        if ep == "res":
            r = rn(M, N); kw.update(residual=r); ref = ref + r.float()
        if ep == "act":
            kw.update(act=L.ACT_RELU); ref = torch.relu(ref)
        if ep == "bias":
            b = torch.randn(N, device=dev, generator=g); kw.update(bias=b, act=L.ACT_GELU); ref = torch.nn.functional.gelu(ref + b)
        if ep == "dact":
            z = torch.relu(rn(M, N)); kw.update(dact=L.ACT_RELU, z=z); ref = ref * (z.float() > 0)
        outs = {}
        for name, o in (("default", {}), ("a4", dict(gemm_a4=2)), ("a4v1", dict(gemm_a4=3))):
            setopts(o)
            C = torch.full((M, N), float("nan"), dtype=torch.float32 if ep == "f32" else torch.bfloat16, device=dev)
            L.gemm(A, B, C, M, N, K, **kw)
            torch.cuda.synchronize()
            outs[name] = (C, L.lib().v2s_last_gemm_kernel().decode())
        setopts({})
        e_a4, e_v1, e_def = relerr(outs["a4"][0], ref), relerr(outs["a4v1"][0], ref), relerr(outs["default"][0], ref)
        tol = 2e-5 if ep == "f32" else 3e-3
        ok = all("gemm_a4" in outs[n][1] and relerr(outs[n][0], ref) < tol and bool(torch.isfinite(outs[n][0]).all()) for n in ("a4", "a4v1"))
        ok = ok and bool(torch.equal(outs["a4"][0], outs["a4v1"][0]))       # same K order, same single rounding: the two forms agree bit for bit
        bad += 0 if ok else 1
        print(f"check {kind} {M}x{N}x{K} {ep or 'plain':5s}: a4 rel err {e_a4:.2e} / one-tile {e_v1:.2e} (default {e_def:.2e})  [{outs['a4'][1]} | {outs['a4v1'][1]}] {'ok' if ok else 'FAILED'}", flush=True)
    return bad


def timed(f, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    bad = check()
    print(f"{bad} correctness case(s) failed", flush=True)
    if "--check-only" in sys.argv or bad:
        sys.exit(1 if bad else 0)
    tot = {n: 0.0 for n, _ in VARIANTS}; tot["vendor"] = 0.0
    for kind, M, N, K, what in SHAPES:
        g = torch.Generator(device=dev); g.manual_seed(M * 7 + N * 3 + K)
        rn = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
        A = rn(M, K)
        B = rn(K, N) if kind == "NN" else rn(N, K)
        kw = dict(transB=(kind == "NN"), ldb=N if kind == "NN" else K)
        C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        Cv = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        Bt = B if kind == "NN" else B.t()
        res = {n: [] for n, _ in VARIANTS}; res["vendor"] = []
        kern = {}
        n_it = 10 if M * N * K > 2e11 else 20
        for rep in range(5):
            for name, o in VARIANTS:
                setopts(o)
                f = lambda: L.gemm(A, B, C, M, N, K, **kw)
                f(); kern[name] = L.lib().v2s_last_gemm_kernel().decode()
                res[name].append(timed(f, n_it))
            fv = lambda: torch.mm(A, Bt, out=Cv)
            fv(); res["vendor"].append(timed(fv, n_it))
        setopts({})
        fl = 2.0 * M * N * K
        med = {n: sorted(v)[2] for n, v in res.items()}
        for n in tot:
            tot[n] += med[n]
        print(f"{kind} {M:6d}x{N:6d}x{K:6d} {what:18s} " + "  ".join(f"{n} {med[n]:7.1f} ({fl / med[n] / 1e6:5.0f})" for n in med)
              + f"   a4/vendor {med['vendor'] / med['a4']:.3f}  a4/default {med['default'] / med['a4']:.3f}  [{kern['default']}]", flush=True)
    print("sum us: " + "  ".join(f"{n} {v:.0f}" for n, v in tot.items()) + f"   vendor/a4 {tot['vendor'] / tot['a4']:.3f}  default/a4 {tot['default'] / tot['a4']:.3f}")


if __name__ == "__main__":
    main()
