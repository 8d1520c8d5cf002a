# SQ counter pass (round 4): the 4-wave 256x192 kernel (gemm_wt, main loop only and full), the write-out-wave kernel and the default
# dispatch on the same shapes; summary: tools/pmc_sq_summary.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_r4
i=0
for spec in "32000 2304 768 0 0 1|gemm_w128=2,gemm_dbg=2" "32000 2304 768 0 0 1|gemm_w128=2" "32000 2304 768 0 0 1|" "8192 8192 8192 0 0 1|gemm_w128=2,gemm_dbg=2" "8192 8192 8192 0 0 1|" "32000 768 768 0 0 1|gemm_ps=2,gemm_ps_nst=2" "32000 768 768 0 0 1|"; do
  cfg=${spec%%|*}; opts=${spec##*|}
  i=$((i+1))
  tag=$(echo $cfg | awk '{print $1"_"$2"_"$3}')_v$i
  V2S_OPTIONS=$opts rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $R/gpurun_out/pmc_r4/pmc_$tag -o pmc --output-format csv -- python $R/tools/gemm_one.py $cfg > $R/gpurun_out/pmc_r4/pmc_$tag.log 2>&1
  echo "== $cfg [$opts]"
done
python $R/tools/pmc_sq_summary.py $R/gpurun_out/pmc_r4
