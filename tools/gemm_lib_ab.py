"""Same-process A/B of GEMM epilogue variants from two builds of the library (shapes of the step whose epilogue reads a global
operand: the ReLU-mask source of the wo dgrad, the residual of the O / wo projections, bias + GELU' of the ViT).
usage: python tools/gemm_lib_ab.py libA.so libB.so[@gemm_big=0]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
paths = sys.argv[1:] or [L.LIB_PATH]

def use(spec):
    """spec = lib.so[@option=value[,option=value]]"""
    path, _, opts = spec.partition("@")
    L.LIB_PATH = os.path.abspath(path); L._LIB = None; L.lib()
    for k, v in (("gemm_big", 1), ("gemm_p8", 1), ("gemm_dma", 2)):       # options live in the loaded library: back to the defaults first
        L.set_option(k, v)
    for kv in filter(None, opts.split(",")):
        k, v = kv.split("="); L.set_option(k, int(v))

def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

CASES = [("enc QKV fwd (deferred)", "NT", 32000, 2304, 768, ""), ("enc wi fwd relu+drop (deferred)", "NT", 32000, 3072, 768, "act"),
         ("enc wi fwd bare (deferred)", "NT", 32000, 3072, 768, ""), ("cross K|V fwd (deferred)", "NT", 35200, 1536, 768, ""),
         ("t5-large wi fwd relu+drop", "NT", 16000, 4096, 1024, "act"),
         ("enc wo dgrad relu-mask+drop", "NN", 32000, 3072, 768, "dact"), ("enc wo dgrad bare", "NN", 32000, 3072, 768, ""),
         ("enc O fwd residual+drop", "NT", 32000, 768, 768, "res"), ("enc O fwd bare", "NT", 32000, 768, 768, ""),
         ("enc wo fwd residual+drop", "NT", 32000, 768, 3072, "res"), ("enc wo fwd bare", "NT", 32000, 768, 3072, ""),
         ("dec wo dgrad relu-mask+drop", "NN", 8192, 3072, 768, "dact"), ("dec O fwd residual+drop", "NT", 8192, 768, 768, "res"),
         ("dec wo fwd residual+drop", "NT", 8192, 768, 3072, "res"), ("ViT fc2 fwd residual", "NT", 3200, 768, 2048, "res"),
         ("ViT fc1 dgrad gelu'", "NN", 3200, 2048, 768, "dgelu")]
for name, kind, M, N, K, ep in CASES:
    torch.manual_seed(0)
    A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    B = (torch.randn(*((K, N) if kind == "NN" else (N, K)), device=dev) * 0.5).to(torch.bfloat16)
    Z = torch.relu(torch.randn(M, N, device=dev)).to(torch.bfloat16) if ep in ("dact", "dgelu") else None
    R = torch.randn(M, N, device=dev).to(torch.bfloat16) if ep == "res" else None
    outs = {}
    res = {p: [] for p in paths}
    kern = ""
    for rep in range(3):
        for p in paths:
            use(p)
            Cc = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            kw = dict(transB=(kind == "NN"), ldb=N if kind == "NN" else K)
            if ep == "act": kw.update(act=L.ACT_RELU, dropout_p=0.1, dropout_seed=3)
            if ep == "dact": kw.update(dact=L.ACT_RELU, z=Z, dropout_p=0.1, dropout_seed=3)
            if ep == "dgelu": kw.update(dact=L.ACT_GELU, z=Z)
            if ep == "res": kw.update(residual=R, dropout_p=0.1 if M > 4000 else 0.0, dropout_seed=3)
            f = lambda: L.gemm(A, B, Cc, M, N, K, **kw)
            res[p].append(t(f)); outs[p] = Cc
            kern = L.lib().v2s_last_gemm_kernel().decode()
    same = all(torch.equal(outs[paths[0]], outs[p]) for p in paths)
    print(f"{name:30s} {kind} {M}x{N}x{K} [{kern}] " + "  ".join(f"{min(res[p]):7.1f} us" for p in paths) + ("  identical" if same else "  DIFFERENT"))
