#!/bin/bash
# Profiling build: libvid2seq_hip_abl.so = the library with gemm_a4_kernel's ablated main loops compiled in (gemm_dbg = 11 nodma, 12 noread,
# 13 neither, 14 no barrier; results invalid).  usage: bash tools/build_a4_ablations.sh   then   tools/gemm_a4_ablate.py
set -e
cd "$(dirname "$0")/../vidchapters_amd/csrc"
bash build.sh > /dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -munsafe-fp-atomics"
OBJS=""; for s in v2s_api v2s_norm v2s_attn v2s_misc v2s_optim v2s_decode v2s_memattn v2s_data; do OBJS="$OBJS build/$s.o"; done
# one library per slot plan of the generator (A4_PLAN): the committed one ("spread") and the experimental ones
for plan in ${A4_PLANS:-spread early}; do
  mkdir -p build/plan_$plan
  A4_PLAN=$plan python3 gen_gemm_a4.py build/plan_$plan/v2s_gemm_a4.inc > /dev/null
  $HIPCC $FLAGS -DV2S_A4_ABLATIONS -DA4_INC="\"build/plan_$plan/v2s_gemm_a4.inc\"" -c v2s_gemm.hip -o build/v2s_gemm_abl_$plan.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../tools/libvid2seq_hip_abl_$plan.so build/v2s_gemm_abl_$plan.o $OBJS
  echo "built tools/libvid2seq_hip_abl_$plan.so"
done
