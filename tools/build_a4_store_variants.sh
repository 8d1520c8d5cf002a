#!/bin/bash
# Experiment builds: the library with another cache policy on gemm_a4p's output stores (generator env A4_STORE_MOD).
# usage: bash tools/build_a4_store_variants.sh   ->  tools/libvid2seq_hip_st_<tag>.so   (step_ab.py / gemm_a4_ab.py "lib=...")
set -e
cd "$(dirname "$0")/../vidchapters_amd/csrc"
bash build.sh > /dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -munsafe-fp-atomics"
OBJS=""; for s in v2s_api v2s_norm v2s_attn v2s_misc v2s_optim v2s_decode v2s_memattn v2s_data; do OBJS="$OBJS build/$s.o"; done
pids=()
for tag in sc1 nt sc0sc1; do
  mod=$(echo $tag | sed 's/sc0sc1/sc0 sc1/')
  mkdir -p build/st_$tag
  A4_STORE_MOD="$mod" python3 gen_gemm_a4.py build/st_$tag/v2s_gemm_a4.inc > /dev/null
  ( $HIPCC $FLAGS -DA4_INC="\"build/st_$tag/v2s_gemm_a4.inc\"" -c v2s_gemm.hip -o build/v2s_gemm_st_$tag.o &&
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../tools/libvid2seq_hip_st_$tag.so build/v2s_gemm_st_$tag.o $OBJS && echo "built tools/libvid2seq_hip_st_$tag.so" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
