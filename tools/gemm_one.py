"""Run ONE GEMM shape a few times (for rocprofv3 --pmc passes).  usage: gemm_one.py M N K ta tb gemm_big [f32acc]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
M, N, K, ta, tb, big = (int(x) for x in sys.argv[1:7])
f32 = len(sys.argv) > 7
dev = "cuda"
L.set_option("gemm_big", big)
A = torch.randn((K, M) if ta else (M, K), device=dev).to(torch.bfloat16)
B = torch.randn((K, N) if tb else (N, K), device=dev).to(torch.bfloat16)
C = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
ws = torch.empty(64 * 1024 * 1024 // 4, device=dev) if f32 else None
kw = {}
if os.environ.get("GEMM_ONE_ACT") == "relu":            # forward epilogues: GEMM_ONE_ACT=relu [GEMM_ONE_DROP=0.1]
    kw = dict(act=L.ACT_RELU, dropout_p=float(os.environ.get("GEMM_ONE_DROP", "0")), dropout_seed=7)
for _ in range(6):
    L.gemm(A, B, C, M, N, K, transA=bool(ta), transB=bool(tb), accumulate=f32, workspace=ws, **kw)
torch.cuda.synchronize()
