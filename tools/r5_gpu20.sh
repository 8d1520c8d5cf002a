cd $GRAFT_REPO_ROOT
timeout 900 python tools/step_ab.py "gemm_a4=0,gemm_a4_relu=0" "gemm_a4=1,gemm_a4_relu=0" "gemm_a4=1,gemm_a4_relu=1" --model t5-large --frames 200 --asr 2000 --steps 6 --block 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_step_ab_cfg5_a4.txt
timeout 900 python bench.py --model t5-large --frames 200 --asr-tokens 2000 --steps 10 --warmup 2 --no-cpu-baseline --no-generate > gpurun_out/r05_cfg5_bench.log 2> gpurun_out/r05_cfg5_bench.err
tail -1 gpurun_out/r05_cfg5_bench.log > gpurun_out/r05_bench_cfg5_exact_n1.json; cut -c1-300 gpurun_out/r05_bench_cfg5_exact_n1.json
