"""FETCH_SIZE of one dgrad shape under different tile walks (option gemm_order = tile rows per group): 3 launches per order, in order.
Run under rocprofv3 --pmc FETCH_SIZE --kernel-trace (tools/pmc_order_probe.sh); without the profiler it prints the timings."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L
dev = "cuda"
ORDERS = [1, 2, 4, 8, 16, 32]
SHAPES = [(32000, 768, 3072, True), (32000, 768, 768, True), (32000, 768, 3072, False)]
if __name__ == "__main__":
    for M, N, K, tb in SHAPES:
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        B = torch.randn((K, N) if tb else (N, K), device=dev).to(torch.bfloat16)
        C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        for o in ORDERS:
            L.set_option("gemm_order", o)
            for _ in range(3):
                L.gemm(A, B, C, M, N, K, transB=tb)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): L.gemm(A, B, C, M, N, K, transB=tb)
            e1.record(); torch.cuda.synchronize()
            print(f"{M}x{N}x{K} {'NN' if tb else 'NT'} order {o}: {e0.elapsed_time(e1) * 100:.1f} us")
