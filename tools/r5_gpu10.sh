cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or sum_n" 2>&1 | tail -3
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_a4p_dact_ab.txt
import torch, sys, os
sys.path.insert(0, os.getcwd())
from vidchapters_amd import lib as L
def timed(f, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N, K in ((32000, 3072, 768), (8192, 3072, 768)):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); B = torch.randn(K, N, device="cuda").to(torch.bfloat16)
    z = (torch.relu(torch.randn(M, N, device="cuda")) * (torch.rand(M, N, device="cuda") > 0.1)).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    res = {0: [], 1: []}
    for rep in range(5):
        for mode in (0, 5):
            L.set_option("gemm_a4", mode)
            f = lambda: L.gemm(A, B, C, M, N, K, transB=True, ldb=N, dact=L.ACT_RELU, z=z, dropout_p=0.1, dropout_seed=3)
            f(); k = L.lib().v2s_last_gemm_kernel().decode(); res[mode].append((timed(f, 10), k))
    L.set_option("gemm_a4", 1)
    fl = 2.0 * M * N * K
    for mode in (0, 5):
        t = sorted(x[0] for x in res[mode])[2]
        print(f"wo dgrad (ReLU mask + dropout scale) {M}x{N}x{K} gemm_a4={mode}: {t:.1f} us ({fl / t / 1e6:.0f} TF/s) {res[mode][0][1]}")
PY
timeout 900 python tools/step_ab.py "gemm_a4=0" "gemm_a4=1,eng:dmem_parts=0" "gemm_a4=1,eng:dmem_parts=1" --steps 10 --block 4 2>&1 | tail -4 | tee gpurun_out/r05_step_ab_a4.txt
