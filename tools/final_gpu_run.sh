# The round-end validation recipe as one gpurun command: full -m gpu suite, smoke, default bench line, rocprofv3 kernel stats of the same step,
# PMC traffic of the dominant kernel, the vendor table.  usage: gpurun --timeout 4500 -- bash tools/final_gpu_run.sh <tag>   (outputs under gpurun_out/,
# copied to profiles/ by hand)
T=${1:-r06_a}
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q --tb=short -rf > gpurun_out/${T}_pytest_full.txt 2>&1; tail -6 gpurun_out/${T}_pytest_full.txt > gpurun_out/${T}_pytest.txt; grep -q failed gpurun_out/${T}_pytest.txt || rm -f gpurun_out/${T}_pytest_full.txt
cat gpurun_out/${T}_pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.txt 2>&1; tail -1 gpurun_out/${T}_smoke.txt
timeout 900 python bench.py > gpurun_out/${T}_bench.log 2> gpurun_out/${T}_bench.err
tail -1 gpurun_out/${T}_bench.log > gpurun_out/${T}_bench_default_n1.json
cut -c1-300 gpurun_out/${T}_bench_default_n1.json
timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-generate > gpurun_out/${T}_bench20.log 2>/dev/null; tail -1 gpurun_out/${T}_bench20.log > gpurun_out/${T}_bench_steps20_n1.json; cut -c1-200 gpurun_out/${T}_bench_steps20_n1.json
bash tools/prof_bench.sh prof_$T > /dev/null 2>&1
python tools/kernel_stats_summary.py gpurun_out/prof_$T/trace gpurun_out/prof_$T/trace.log gpurun_out/${T}_kernel_stats_cfg2_nooverlap.txt 5 | head -3
rm -rf gpurun_out/prof_$T
bash tools/pmc_traffic.sh pmc_$T dgrad > /dev/null 2>&1
python tools/pmc_traffic_summary.py gpurun_out/pmc_$T gpurun_out/${T}_pmc_traffic_gemm_dma_dgrad.json dgrad | head -8
rm -rf gpurun_out/pmc_$T
