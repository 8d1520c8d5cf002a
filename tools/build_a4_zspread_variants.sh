#!/bin/bash
# Experiment builds: the masked-dgrad epilogue of gemm_a4p with its z loads spread over two iterations (generator env A4_ZSPREAD).
# usage: bash tools/build_a4_zspread_variants.sh  ->  tools/libvid2seq_hip_z<n>.so  (only for K >= 640 problems; step_ab.py "lib=...")
set -e
cd "$(dirname "$0")/../vidchapters_amd/csrc"
bash build.sh > /dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -munsafe-fp-atomics"
OBJS=""; for s in v2s_api v2s_norm v2s_attn v2s_misc v2s_optim v2s_decode v2s_memattn v2s_data; do OBJS="$OBJS build/$s.o"; done
pids=()
for z in 3 5; do
  mkdir -p build/z$z
  A4_ZSPREAD=$z python3 gen_gemm_a4.py build/z$z/v2s_gemm_a4.inc > /dev/null
  ( $HIPCC $FLAGS -DA4_INC="\"build/z$z/v2s_gemm_a4.inc\"" -c v2s_gemm.hip -o build/v2s_gemm_z$z.o &&
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../tools/libvid2seq_hip_z$z.so build/v2s_gemm_z$z.o $OBJS && echo "built tools/libvid2seq_hip_z$z.so" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
