#!/bin/bash
# Diagnostic build of the library for tools/skinny_stamps.py / tools/decode_chain_study.sh: v2s_gemm.hip with -DSKINNY_STAMPS=1 (the skinny GEMM's blocks stamp
# the 100 MHz reference clock at their phase boundaries into the caller's workspace), every other object from the product build.  Run build.sh first.
set -e
cd "$(dirname "$0")/../vidchapters_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -munsafe-fp-atomics"
mkdir -p build/stamps
/opt/rocm/bin/hipcc $FLAGS -DSKINNY_STAMPS=1 -c v2s_gemm.hip -o build/stamps/v2s_gemm.o
OBJS="build/stamps/v2s_gemm.o"; for s in v2s_api v2s_norm v2s_attn v2s_misc v2s_optim v2s_decode v2s_memattn v2s_data; do OBJS="$OBJS build/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libvid2seq_hip_stamps.so $OBJS
echo "built tools/libvid2seq_hip_stamps.so"
