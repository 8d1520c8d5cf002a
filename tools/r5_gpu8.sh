cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for plan in spread early; do
  timeout 600 python tools/gemm_a4_ablate.py --plan=$plan 2>&1 | grep -v amdgpu.ids
  timeout 600 python tools/gemm_a4_ablate.py --plan=$plan --persistent 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r05_gemm_a4_plans.txt
