// Issue rates of the VALU instructions that make up the attention softmax / dropout code on gfx950, and whether MFMA and VALU
// work of DIFFERENT waves on one SIMD overlaps.  One block of 256 (or 512) threads per CU = 1 (or 2) waves per SIMD; every lane
// runs 8 independent dependency chains, 64 instructions per loop trip, cycles read with s_memtime by wave 0.
// build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

enum { OP_ADD, OP_PK_ADD, OP_FMA, OP_PK_FMA, OP_PK_MUL, OP_EXP, OP_CVT_PK, OP_MUL24, OP_MUL_LO, OP_XOR, OP_CNDMASK, OP_CMP_CND, OP_PERM,
       OP_MAX3, OP_ADD_F64, OP_CVT_F64, OP_NOPS };
static const char* NAMES[] = {"v_add_f32", "v_pk_add_f32", "v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_exp_f32", "v_cvt_pk_bf16_f32",
                              "v_mul_u32_u24", "v_mul_lo_u32", "v_xor_b32", "v_cndmask_b32", "v_cmp+v_cndmask", "v_perm_b32", "v_max3_f32",
                              "v_add_f64", "v_cvt_f64_f32"};

template <int OP>
__global__ __launch_bounds__(512) void rate_kernel(long long* out, float seed, int trips) {
  float a[8]; f32x2 p[8]; double d[8]; uint32_t u[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f32x2{a[i], a[i] + 1.f}; d[i] = a[i]; u[i] = (uint32_t)a[i]; }
  const f32x2 c2 = {seed, seed * 0.5f};
  const long long t0 = clock64();
  for (int t = 0; t < trips; ++t) {
#define X_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
#define X_PK_ADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
#define X_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed));
#define X_PK_FMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(c2));
#define X_PK_MUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
#define X_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define X_CVT_PK(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
#define X_MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
#define X_MUL_LO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
#define X_XOR(i) asm volatile("v_xor_b32 %0, 0x6ef362, %0" : "+v"(u[i]));
#define X_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(seed));
#define X_CMP_CND(i) asm volatile("v_cmp_le_u32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]), "v"(seed) : "vcc");
#define X_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
#define X_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(seed));
#define X_ADD_F64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
#define X_CVT_F64(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
    if (OP == OP_ADD) { BODY64(X_ADD) }
    if (OP == OP_PK_ADD) { BODY64(X_PK_ADD) }
    if (OP == OP_FMA) { BODY64(X_FMA) }
    if (OP == OP_PK_FMA) { BODY64(X_PK_FMA) }
    if (OP == OP_PK_MUL) { BODY64(X_PK_MUL) }
    if (OP == OP_EXP) { BODY64(X_EXP) }
    if (OP == OP_CVT_PK) { BODY64(X_CVT_PK) }
    if (OP == OP_MUL24) { BODY64(X_MUL24) }
    if (OP == OP_MUL_LO) { BODY64(X_MUL_LO) }
    if (OP == OP_XOR) { BODY64(X_XOR) }
    if (OP == OP_CNDMASK) { BODY64(X_CNDMASK) }
    if (OP == OP_CMP_CND) { BODY64(X_CMP_CND) }
    if (OP == OP_PERM) { BODY64(X_PERM) }
    if (OP == OP_MAX3) { BODY64(X_MAX3) }
    if (OP == OP_ADD_F64) { BODY64(X_ADD_F64) }
    if (OP == OP_CVT_F64) { BODY64(X_CVT_F64) }
  }
  const long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1] + (float)d[i] + (float)u[i];
  if (s == 12345.678f) out[1] = 1;                // keep the chains alive
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}

// mode 0: every wave does MFMA; 1: every wave does VALU; 2: waves 0-3 MFMA, waves 4-7 VALU (one of each per SIMD)
__global__ __launch_bounds__(512) void overlap_kernel(long long* out, float seed, int trips, int mode) {
  const int wave = threadIdx.x >> 6;
  f32x4 acc[4];
  float a[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{seed, seed, seed, seed};
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = seed + i;
  bf16x8 fa, fb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed + i); fb[i] = (__bf16)(seed - i); }
  const bool do_mfma = mode == 0 || (mode == 2 && wave < 4);
  __syncthreads();
  const long long t0 = clock64();
  if (do_mfma) {
    for (int t = 0; t < trips; ++t) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[j & 3], 0, 0, 0);   // 8 MFMA = 128 issue cycles
    }
  } else {
    for (int t = 0; t < trips; ++t) {
      REP8(X_FMA) REP8(X_FMA) REP8(X_FMA) REP8(X_FMA)                                                                  // 32 VALU = 128 issue cycles
    }
  }
  const long long t1 = clock64();
  __syncthreads();
  const long long t2 = clock64();
  float s = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7] + acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  if (s == 12345.678f) out[3] = 1;
  if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) out[threadIdx.x ? 1 : 0] = t1 - t0;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[2] = t2 - t0;
}

template <int OP>
void run_one(long long* d_out, int threads) {
  const int trips = 2000;
  long long h[2];
  hipLaunchKernelGGL(rate_kernel<OP>, dim3(256), dim3(threads), 0, 0, d_out, 1.0001f, 10);
  hipLaunchKernelGGL(rate_kernel<OP>, dim3(256), dim3(threads), 0, 0, d_out, 1.0001f, trips);
  hipDeviceSynchronize();
  hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  const int per_trip = OP == OP_CMP_CND ? 128 : 64;
  printf("  %-20s %d wave/SIMD: %6.2f cycles per instruction per wave  (%5.2f per SIMD)\n", NAMES[OP], threads / 256,
         (double)h[0] / trips / per_trip, (double)h[0] / trips / per_trip / (threads / 256));
}
template <int OP>
void run_all(long long* d) {
  if constexpr (OP < OP_NOPS) { run_one<OP>(d, 256); run_one<OP>(d, 512); run_all<OP + 1>(d); }
}

int main() {
  long long* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  printf("VALU issue rates (s_memtime cycles, wave 0 of block 0; 256 blocks):\n");
  run_all<0>(d);
  printf("MFMA / VALU overlap across waves of one SIMD (8 waves per CU; each wave: trips x 128 issue cycles of its own kind):\n");
  for (int mode = 0; mode < 3; ++mode) {
    const int trips = 4000;
    long long h[3];
    hipLaunchKernelGGL(overlap_kernel, dim3(256), dim3(512), 0, 0, d, 1.0001f, 10, mode);
    hipLaunchKernelGGL(overlap_kernel, dim3(256), dim3(512), 0, 0, d, 1.0001f, trips, mode);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("  mode %d (%s): wave0 %6.1f cycles/trip, wave4 %6.1f cycles/trip, block %6.1f cycles/trip\n", mode,
           mode == 0 ? "all MFMA" : mode == 1 ? "all VALU" : "4 MFMA waves + 4 VALU waves", (double)h[0] / trips, (double)h[1] / trips, (double)h[2] / trips);
  }
  return 0;
}
