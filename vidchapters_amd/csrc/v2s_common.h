// Shared device/host helpers for libvid2seq_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vid2seq_hip.h"

typedef unsigned short bf16_t;  // raw bf16 bits in memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define V2S_LDS __attribute__((address_space(3)))

// ---------------------------------------------------------------- error plumbing (host)
void v2s_set_error(const char* fmt, ...);
#define V2S_CHECK(cond, code, ...)            \
  do {                                        \
    if (!(cond)) {                            \
      v2s_set_error(__VA_ARGS__);             \
      return (code);                          \
    }                                         \
  } while (0)
#define V2S_LAUNCH_CHECK()                                                    \
  do {                                                                        \
    hipError_t e__ = hipGetLastError();                                       \
    if (e__ != hipSuccess) {                                                  \
      v2s_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,            \
                    hipGetErrorString(e__));                                  \
      return V2S_ERR_LAUNCH;                                                  \
    }                                                                         \
  } while (0)

const uint32_t* v2s_seed_salt();  // device word XOR-ed into every dropout seed by the kernels (v2s_set_seed_salt), or NULL
int v2s_opt_tr_read();   // 1 = use ds_read_b64_tr_b16 for transposed operand fragments
int v2s_opt_gemm_dma();  // 2 = LDS-DMA (global_load_lds) 128x128 main loop for every variant where K % 64 == 0 (default), 1 = transposed only, 0 = never
int v2s_opt_attn_bwd_part(); // profiling aid for v2s_attn_bwd: 0 = both kernels (default), 1 = dQ only, 2 = dK/dV only
int v2s_opt_gemm_skinny(); // 1 = weight-streaming kernels for cached decoding (default; other values: A/B block shapes, see the header), 0 = general tiles
int v2s_opt_gemm_order(); // tile walk of the tiled GEMM kernels: GM > 0 = grouped GM tile rows deep with tile-major split-K (default 4), 0 = row-major
int v2s_opt_gemm_split(); // 1 = split-K slice count from the rounds x length cost model (default), 0 = fixed block-count target
int v2s_opt_gemm_p8();   // 8-phase ping-pong 256-row kernel: 0 = never, 1 = where it measured faster (default), 2 = 256x256 wherever legal, 3 = 256x128 wherever legal
int v2s_opt_ce_fused();  // reserved
int v2s_opt_gemm_dbg();  // profiling aid for the 8-phase kernel: 1 = epilogue without the global store, 2 = no epilogue (results invalid)
int v2s_opt_fp32_io();    // debug: 1 = the norm / cross-entropy / attention entry points take and return FP32 activations (attention: an fp32-arithmetic
                          // reference kernel); parity tests against fp32 references at <= 1e-4, never set by the product path
int v2s_opt_gemm_a4();    // 4-wave asm-scheduled 256x256 kernels (128x128 wave tiles in AGPRs, 32x32x16 MFMA): 0 = never, 1 = where they measured faster (default),
                          // 2 = wherever legal (persistent deferred-write-out form where its epilogue allows, else the one-tile form), 3 = one-tile form wherever legal, 4 = like 1 plus the split-K weight gradients
                          // with a long contraction, 5 = like 1 plus the ReLU-mask dgrad epilogue (both faster alone, slower in the step:
                          // DESIGN.md 8a-r5)
int v2s_opt_gemm_a4_grid(); // blocks of the persistent a4p kernel: 0 = one per CU (default), n = at most n (leaves CUs to concurrent streams: A/B knob), -1 = the fewest blocks with the same number of rounds
int v2s_opt_gemm_a4_relu(); // 1 (default): the persistent a4p kernel also takes forward GEMMs with a ReLU (+ dropout) epilogue (the FFN's wi); 0: plain epilogues only
int v2s_opt_gemm_a4_walk(); // tile walk of the persistent a4p kernel: 0 = auto (row-major below 16 tile columns, groups of 4 tile rows from there), n = groups of n tile rows
int v2s_opt_attn_order();   // dispatch order of the attention dK / dV kernel (experiment knob, see v2s_attn.hip)
int v2s_opt_gemm_big();  // 0 = never, 1 = 256x256/256x128 tiles where they pay (default), 2 = 256x128 only, 3 = 4-wave 256x128x32 ring kernel for every variant

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {  // round-to-nearest-even, NaN preserved
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
  v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// counter-based dropout RNG.  Activations are always processed in aligned chunks of 8 consecutive elements (linear index
// e0 = multiple of 8): one 32-bit mix of (seed, chunk index) and four 24-bit multiplies give eight 16-bit draws; element e0+j is
// kept iff its draw >= p16.  Every kernel that applies or re-applies a mask (GEMM epilogue, elementwise dropout, embedding)
// calls this one function with the same linear index, so forward and backward masks agree by construction.
__device__ __forceinline__ uint32_t v2s_hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t v2s_mix32(uint32_t x) {
  x *= 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13;
  return x;
}
// seed of a launch: the by-value seed, XOR the device-resident salt if one is set (captured graphs: new masks per replay)
__device__ __forceinline__ uint32_t v2s_salted(uint32_t seed, const uint32_t* salt) { return salt ? seed ^ *salt : seed; }
// bit j of the result = keep element e0 + j.  One full 32-bit mix per chunk of 8 elements, then one rotate + 24-bit multiply per PAIR
// (two 16-bit draws each): 14 integer instructions per chunk instead of the 24 of four full mixes -- the mask is regenerated in the
// write-out phases of the deferred GEMM epilogue, which have no slack (round 3).
__device__ __forceinline__ uint32_t v2s_keep8(unsigned long long e0, uint32_t seed, uint32_t p16) {
  const uint32_t base = v2s_mix32(seed * 0x9E3779B1u + (uint32_t)(e0 >> 3) + (uint32_t)(e0 >> 35) * 0x85EBCA6Bu);
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // pair i: the 24-bit window of the mix that starts at bit 8 i (a rotate) times its own odd multiplier -- xor-ing pair constants
    // into ONE window left 4 % correlation between some slots of a chunk; this form measures < 0.5 % (noise) over 160 000 chunks
    const uint32_t h = __umul24(i ? __builtin_amdgcn_alignbit(base, base, 8 * i) : base, 0x00EBCA77u + 0x2468u * (uint32_t)i);
    m |= ((h & 0xffffu) >= p16 ? 1u : 0u) << (2 * i);
    m |= ((h >> 16) >= p16 ? 1u : 0u) << (2 * i + 1);
  }
  return m;
}
__device__ __forceinline__ void v2s_drop8(float (&v)[8], unsigned long long e0, uint32_t seed, uint32_t p16, float inv_keep) {
  const uint32_t m = v2s_keep8(e0, seed, p16);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = ((m >> j) & 1u) ? v[j] * inv_keep : 0.f;
}

// exact-erf GELU and its derivative (torch nn.GELU default, vit.py:9)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
  float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// XCD-aware bijective block remap: consecutive logical ids land on the same XCD (8 XCDs, bid%8 -> XCD)
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, loc = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + loc;
}
