#!/bin/bash
# Build libvid2seq_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [-j N]
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -munsafe-fp-atomics: hardware float atomics (global_atomic_add_f32 / ds_add_f32) instead of CAS loops; all our
# atomics target ordinary (coarse-grained) device memory, where they are exact fp32 adds
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -munsafe-fp-atomics"
SRCS="v2s_api v2s_gemm v2s_norm v2s_attn v2s_misc v2s_optim v2s_decode v2s_memattn v2s_data"
mkdir -p build
# the hand-scheduled K loop of gemm_a4_kernel is generated (scoreboarded s_waitcnt counts, MFMA-gap placement): see gen_gemm_a4.py
if [ ! -f v2s_gemm_a4.inc ] || [ gen_gemm_a4.py -nt v2s_gemm_a4.inc ]; then
  python3 gen_gemm_a4.py v2s_gemm_a4.inc
fi
pids=()
for s in $SRCS; do
  if [ ! -f build/$s.o ] || [ $s.hip -nt build/$s.o ] || [ v2s_common.h -nt build/$s.o ] || [ ../../include/vid2seq_hip.h -nt build/$s.o ] ||
     { [ $s = v2s_gemm ] && { [ v2s_gemm_a4.h -nt build/$s.o ] || [ v2s_gemm_a4.inc -nt build/$s.o ]; }; }; then
    ( $HIPCC $FLAGS -c $s.hip -o build/$s.o ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
# The persistent deferred-epilogue GEMM counts its VMEM operations by hand (s_waitcnt vmcnt(N) across barriers): a register spill would
# add scratch loads/stores to that queue.  Fail the build if the compiler ever needs scratch for it.
if [ build/v2s_gemm.o -nt build/v2s_gemm.usage ] || [ ! -f build/v2s_gemm.usage ]; then
  $HIPCC $FLAGS -c v2s_gemm.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2> build/v2s_gemm.usage.raw || true
  grep -A8 "Function Name: .*gemm_p8d_kernel" build/v2s_gemm.usage.raw | grep -E "Function Name|ScratchSize" > build/v2s_gemm.usage || true
  if grep -E "ScratchSize \[bytes/lane\]: [1-9]" build/v2s_gemm.usage; then
    echo "ERROR: gemm_p8d_kernel spills to scratch (see build/v2s_gemm.usage): its hand-counted vmcnt waits would be wrong" >&2
    rm -f build/v2s_gemm.o; exit 1
  fi
  # no GEMM kernel is meant to touch scratch (an accumulator array indexed by a rolled loop, a pointer select between a register
  # value and memory: both have happened) -- say so loudly, the kernels stay correct but lose 10-30 %
  grep -E "Function Name|ScratchSize" build/v2s_gemm.usage.raw | paste - - | grep -vE "lane\]: 0 " | sed 's/.*Function Name: \([^ ]*\).*lane\]: \([0-9]*\).*/WARNING: \1 uses \2 bytes of scratch per lane/' >&2 || true
fi
# gemm_a4_kernel keeps its 256 accumulators in LITERAL AGPRs across asm statements: any v_accvgpr_* / scratch access the compiler
# emits on its own inside that kernel could only be a spill into them.  Audit the ISA (everything outside ;;#ASMSTART .. ;;#ASMEND).
if [ build/v2s_gemm.o -nt build/v2s_gemm_a4.audit ] || [ ! -f build/v2s_gemm_a4.audit ]; then
  ( cd build && $HIPCC $FLAGS -S --cuda-device-only ../v2s_gemm.hip -o v2s_gemm.s 2>/dev/null ) || true
  python3 - <<'PYEOF' > build/v2s_gemm_a4.audit
import re, sys
txt = open("build/v2s_gemm.s").read()
bad = nk = 0
for m in re.finditer(r"^(_ZN\S*gemm_a4p?_kernel\S*):[^\n]*\n(.*?)\n\s*s_endpgm", txt, re.S | re.M):
    body = re.sub(r";;#ASMSTART.*?;;#ASMEND", "", m.group(2), flags=re.S)
    hits = [l for l in body.splitlines() if re.search(r"v_accvgpr|scratch_|buffer_(load|store).*offen.*s\[0:3\]", l)]
    print(m.group(1), "compiler-emitted AGPR/scratch instructions:", len(hits))
    bad += len(hits); nk += 1
print("BAD" if bad else ("CLEAN" if nk == 8 else "NOT-FOUND"))
PYEOF
  cat build/v2s_gemm_a4.audit
  if grep -q BAD build/v2s_gemm_a4.audit; then echo "ERROR: the compiler touched AGPRs / scratch inside gemm_a4_kernel" >&2; rm -f build/v2s_gemm.o; exit 1; fi
  grep -q CLEAN build/v2s_gemm_a4.audit || { echo "ERROR: gemm_a4 ISA audit did not run" >&2; exit 1; }
fi
# The decode memory-attention kernel keeps 192 accumulators in AGPRs and ~200 VGPRs: a spill puts scratch reloads (and their vmcnt(0)
# waits) into every iteration of its loop -- measured 5 us per 16-key group instead of 0.7.  Say so loudly.
if [ build/v2s_memattn.o -nt build/v2s_memattn.usage ] || [ ! -f build/v2s_memattn.usage ]; then
  $HIPCC $FLAGS -c v2s_memattn.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2> build/v2s_memattn.usage.raw || true
  grep -E "Function Name|ScratchSize" build/v2s_memattn.usage.raw | paste - - > build/v2s_memattn.usage || true
  grep -vE "lane\]: 0 " build/v2s_memattn.usage | sed 's/.*Function Name: \([^ ]*\).*lane\]: \([0-9]*\).*/WARNING: \1 uses \2 bytes of scratch per lane/' >&2 || true
fi
OBJS=""; for s in $SRCS; do OBJS="$OBJS build/$s.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libvid2seq_hip.so $OBJS
echo "built $(realpath ../libvid2seq_hip.so)"
