# SQ counter pass (round 5): gemm_a4_kernel (main loop only and full) next to the default dispatch; summary: tools/pmc_sq_summary.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_r5
i=0
for spec in "32000 2304 768 0 0 1|gemm_a4=3,gemm_dbg=2" "32000 2304 768 0 0 1|gemm_a4=2" "32000 2304 768 0 0 1|gemm_a4=2,gemm_dbg=1" "32000 2304 768 0 0 1|" "32000 768 2304 0 1 1|gemm_a4=2" "8192 8192 8192 0 0 1|gemm_a4=2"; do
  cfg=${spec%%|*}; opts=${spec##*|}
  i=$((i+1))
  tag=$(echo $cfg | awk '{print $1"_"$2"_"$3}')_v$i
  V2S_OPTIONS=$opts timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $R/gpurun_out/pmc_r5/pmc_$tag -o pmc --output-format csv -- python $R/tools/gemm_one.py $cfg > $R/gpurun_out/pmc_r5/pmc_$tag.log 2>&1
  echo "== $cfg [$opts]"
done
python $R/tools/pmc_sq_summary.py $R/gpurun_out/pmc_r5
