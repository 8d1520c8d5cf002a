# The round-end validation recipe as one gpurun command: full -m gpu suite, smoke, default bench line, rocprofv3 kernel stats of the same step,
# t5-large GEMM A/B.  usage: gpurun --timeout 4500 -- bash tools/final_gpu_run.sh   (outputs under gpurun_out/, copied to profiles/ by hand)
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r05_final_pytest.txt
cat gpurun_out/r05_final_pytest.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_final_smoke.txt 2>&1; tail -1 gpurun_out/r05_final_smoke.txt
timeout 900 python bench.py > gpurun_out/r05_c_bench.log 2> gpurun_out/r05_c_bench.err
tail -1 gpurun_out/r05_c_bench.log > gpurun_out/r05_c_bench_default_n1.json
cut -c1-300 gpurun_out/r05_c_bench_default_n1.json
bash tools/prof_bench.sh prof_r05c > /dev/null 2>&1
python tools/kernel_stats_summary.py gpurun_out/prof_r05c/trace gpurun_out/prof_r05c/trace.log gpurun_out/r05_c_kernel_stats_cfg2_nooverlap.txt 5 | head -3
timeout 600 python tools/gemm_a4_ab.py --large 2>&1 | grep -v "amdgpu.ids\|^check" > gpurun_out/r05_gemm_a4_ab_large.txt; cat gpurun_out/r05_gemm_a4_ab_large.txt | cut -c1-330
