cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_a_bench_default_n1.json 2> gpurun_out/r05_a_bench.log; tail -c 1500 gpurun_out/r05_a_bench_default_n1.json | head -c 1500; echo
bash tools/prof_bench.sh prof_r05a > /dev/null 2>&1
python tools/kernel_stats_summary.py gpurun_out/prof_r05a/trace gpurun_out/prof_r05a/trace.log gpurun_out/r05_a_kernel_stats_cfg2_nooverlap.txt 5 | head -32
