cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r05_final_pytest.txt
cat gpurun_out/r05_final_pytest.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_final_smoke.txt 2>&1; tail -3 gpurun_out/r05_final_smoke.txt
timeout 900 python bench.py > gpurun_out/r05_b_bench.log 2> gpurun_out/r05_b_bench.err
tail -1 gpurun_out/r05_b_bench.log > gpurun_out/r05_b_bench_default_n1.json
cut -c1-400 gpurun_out/r05_b_bench_default_n1.json
bash tools/prof_bench.sh prof_r05b > /dev/null 2>&1
python tools/kernel_stats_summary.py gpurun_out/prof_r05b/trace gpurun_out/prof_r05b/trace.log gpurun_out/r05_b_kernel_stats_cfg2_nooverlap.txt 5 | head -5
