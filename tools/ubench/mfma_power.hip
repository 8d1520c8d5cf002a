// Sustained (power-limited) MFMA throughput on RANDOM bf16 operands, one wave per SIMD, 256 accumulator registers per lane:
//   32x32x16 (16 blocks of 16 accumulators, 8 passes)  vs  16x16x32 (64 blocks of 4 accumulators, 4 passes).
// The 32x32 form moves 25 % more register bytes per flop (accumulators in + out dominate); which one the chip sustains longer at its power
// limit decides the MFMA shape of a GEMM whose main loop is power-bound (gemm_a4's ablations).  ~30 ms per variant.
// build: hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ unsigned h32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ bf16x8 rnd8(unsigned s) {      // 8 bf16 values, roughly uniform in [-2, 2): random sign, exponent and mantissa bits
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned r = h32(s * 8 + i);
    const unsigned short bits = (unsigned short)(((r & 0x8000u)) | ((0x3e + ((r >> 20) & 3)) << 8 >> 1 << 1) | (r & 0xff));   // sign | exponent ~ 2^-2..2^1 | mantissa
    v[i] = __builtin_bit_cast(__bf16, bits);
  }
  return v;
}

template <int ZERO>
__global__ __launch_bounds__(256, 1) void k32(float* sink, int trips) {
  f32x16 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = rnd8(threadIdx.x * 16 + i + blockIdx.x * 7919); b[i] = rnd8(threadIdx.x * 16 + 8 + i + blockIdx.x * 104729); }
  if (ZERO) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)0.f; b[i][j] = (__bf16)0.f; }
  }
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[i], acc[i * 4 + j], 0, 0, 0);
  }
  f32x16 s = acc[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) s += acc[i];
  if (s[0] == 12345.f) sink[0] = s[1];
}
template <int ZERO>
__global__ __launch_bounds__(256, 1) void k16(float* sink, int trips) {
  f32x4 acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = rnd8(threadIdx.x * 16 + i + blockIdx.x * 7919); b[i] = rnd8(threadIdx.x * 16 + 8 + i + blockIdx.x * 104729); }
  if (ZERO) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)0.f; b[i][j] = (__bf16)0.f; }
  }
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i * 8 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b[i], acc[i * 8 + j], 0, 0, 0);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 64; ++i) s += acc[i];
  if (s[0] == 12345.f) sink[0] = s[1];
}
template <typename F> void run(const char* name, F launch, double flop_per_trip_per_wave) {
  float* sink; hipMalloc(&sink, 4);
  launch(sink, 2000);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    const int trips = 60000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch(sink, trips);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.2f ms  %8.1f TF/s chip-wide\n", name, ms, flop_per_trip_per_wave * trips * 256.0 * 4 / (ms * 1e-3) / 1e12);
  }
}
int main() {
  run("32x32x16, 16 blocks, random operands", [](float* s, int t) { hipLaunchKernelGGL((k32<0>), dim3(256), dim3(256), 0, 0, s, t); }, 16 * 32768.0);
  run("16x16x32, 64 blocks, random operands", [](float* s, int t) { hipLaunchKernelGGL((k16<0>), dim3(256), dim3(256), 0, 0, s, t); }, 64 * 16384.0);
  run("32x32x16, 16 blocks, zero operands", [](float* s, int t) { hipLaunchKernelGGL((k32<1>), dim3(256), dim3(256), 0, 0, s, t); }, 16 * 32768.0);
  run("16x16x32, 64 blocks, zero operands", [](float* s, int t) { hipLaunchKernelGGL((k16<1>), dim3(256), dim3(256), 0, 0, s, t); }, 64 * 16384.0);
  run("32x32x16, 16 blocks, random operands (again)", [](float* s, int t) { hipLaunchKernelGGL((k32<0>), dim3(256), dim3(256), 0, 0, s, t); }, 16 * 32768.0);
  run("16x16x32, 64 blocks, random operands (again)", [](float* s, int t) { hipLaunchKernelGGL((k16<0>), dim3(256), dim3(256), 0, 0, s, t); }, 64 * 16384.0);
  return 0;
}
