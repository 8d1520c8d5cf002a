"""What a decode-step skinny GEMM costs as a link of a dependent chain, and why: a replayed graph of 240 dependent launches (each reads what
the previous one wrote) of the decoder's QKV projection (64 x 2304 x 768 with the fused RMSNorm) or O projection (64 x 768 x 768 + residual),
(a) always the same weight tensor (weights and code hot), (b) rotating over W weight tensors (W x 3.5 MB >> L2 / last-level cache: weights
from HBM, as in the decode step), (c) like (b) but with another kernel family (the wo-shaped skinny GEMM, the decode attention) between the
launches, as in the step (instruction cache shared by more code than it holds).  usage: python tools/skinny_chain_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L

dev = torch.device("cuda")
torch.manual_seed(0)
NW = int(os.environ.get("NW", "120"))
x0 = torch.randn(64, 768, device=dev).to(torch.bfloat16)
Wqkv = [torch.randn(2304, 768, device=dev).to(torch.bfloat16) * 0.03 for _ in range(NW)]
Wo = [torch.randn(768, 768, device=dev).to(torch.bfloat16) * 0.03 for _ in range(NW)]
Wi = [torch.randn(3072, 768, device=dev).to(torch.bfloat16) * 0.03 for _ in range(12)]
Wo2 = [torch.randn(768, 3072, device=dev).to(torch.bfloat16) * 0.03 for _ in range(12)]
qkv = torch.zeros(64, 2304, device=dev, dtype=torch.bfloat16)
h = torch.zeros(64, 3072, device=dev, dtype=torch.bfloat16)
xa, xb = x0.clone(), x0.clone()


def chain(kind, n=240):
    """n dependent launches; returns us per launch of the replayed graph"""
    def body():
        a, b = xa, xb
        for i in range(n):
            if kind == "qkv_hot":
                L.gemm(a, Wqkv[0], qkv, 64, 2304, 768, rms_eps=1e-6, decode=True)
                L.gemm(qkv, Wo[0], b, 64, 768, 768, lda=2304, residual=a, decode=True)
            elif kind == "qkv_cold":
                L.gemm(a, Wqkv[i % NW], qkv, 64, 2304, 768, rms_eps=1e-6, decode=True)
                L.gemm(qkv, Wo[i % NW], b, 64, 768, 768, lda=2304, residual=a, decode=True)
            elif kind == "ffn_mix":      # four kernel shapes alternate: QKV, O, wi (relu), wo
                L.gemm(a, Wqkv[i % NW], qkv, 64, 2304, 768, rms_eps=1e-6, decode=True)
                L.gemm(qkv, Wo[i % NW], b, 64, 768, 768, lda=2304, residual=a, decode=True)
                L.gemm(b, Wi[i % 12], h, 64, 3072, 768, rms_eps=1e-6, act=L.ACT_RELU, decode=True)
                L.gemm(h, Wo2[i % 12], a, 64, 768, 3072, residual=b, decode=True)
                continue
            a, b = b, a
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    per = {"qkv_hot": 2, "qkv_cold": 2, "ffn_mix": 4}[kind]
    return best * 1000.0 / (n * per)


libs = sys.argv[1:] or [L.LIB_PATH]          # other builds of the library (same ABI) as arguments: one table row each
for rep in range(2):
    for lp in libs:
        L.LIB_PATH = os.path.abspath(lp); L._LIB = None; L.lib()
        print(f"{os.path.basename(lp):32s} " + "  ".join(f"{kind} {chain(kind):5.2f}" for kind in ("qkv_hot", "qkv_cold", "ffn_mix")) + "   us per launch (node to node, replayed graph)")
