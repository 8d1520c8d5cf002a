"""Run ONE vendor GEMM (torch.mm -> hipBLASLt) a few times, for rocprofv3 --pmc passes next to tools/gemm_one.py.  Measurement tool only."""
import sys, torch
M, N, K = (int(x) for x in sys.argv[1:4])
A = torch.randn(M, K, device="cuda").to(torch.bfloat16); B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(6):
    torch.mm(A, B.t(), out=C)
torch.cuda.synchronize()
