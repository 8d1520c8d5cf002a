#!/bin/bash
# Experiment builds of the attention kernels (same ABI; tools/attn_ab.py libA.so libB.so ...): -D flags after the tag.
# usage: bash tools/build_attn_variants.sh occ3:-DATTN_BWD_OCC=3 prio:-DATTN_PRIO=1   ->  tools/libvid2seq_hip_attn_<tag>.so
set -e
cd "$(dirname "$0")/../vidchapters_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -munsafe-fp-atomics"
OBJS=""; for s in v2s_api v2s_gemm v2s_norm v2s_misc v2s_optim v2s_decode v2s_memattn v2s_data; do OBJS="$OBJS build/$s.o"; done
pids=()
for spec in "$@"; do
  tag=${spec%%:*}; defs=${spec#*:}; defs=${defs//,/ }
  ( $HIPCC $FLAGS $defs -c v2s_attn.hip -o build/v2s_attn_$tag.o -Rpass-analysis=kernel-resource-usage 2> build/v2s_attn_$tag.usage.raw;
    grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" build/v2s_attn_$tag.usage.raw | paste - - - - | sed 's/remark: [^ ]* //g; s/\[-Rpass[^]]*\]//g' | grep -E "bwd_d(q|kv)_kernelILb1ELb[01]ELb[01]ELb1ELi[24]" | sed 's/v2s_attn.hip:[0-9]*:[0-9]*: *//g' > build/v2s_attn_$tag.usage;
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../tools/libvid2seq_hip_attn_$tag.so build/v2s_attn_$tag.o $OBJS && echo "built tools/libvid2seq_hip_attn_$tag.so" && cat build/v2s_attn_$tag.usage ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
