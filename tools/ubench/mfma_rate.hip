// How fast can ONE wave per SIMD (and two) issue independent bf16 MFMAs on gfx950?  16x16x32 (4 passes) vs 32x32x16 (8 passes), accumulators
// in VGPRs or AGPRs, with and without an LDS read between groups of MFMAs (the shape of a GEMM main loop).  Cycles by clock64() of wave 0.
// build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int NACC, bool LDS, int TH>
__global__ __launch_bounds__(TH, TH / 256) void k16(long long* out, float* sink, int trips) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)i; }
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const long long t0 = clock64();
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      if (LDS && (i % 6) == 5) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(lds + ((threadIdx.x * 4 + i * 64) & 4095));
        asm volatile("" ::"v"(v));
      }
    }
  }
  const long long t1 = clock64();
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += acc[i];
  if (s[0] == 12345.f) sink[0] = s[1];
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int NACC, int TH>
__global__ __launch_bounds__(TH, TH / 256) void k32(long long* out, float* sink, int trips) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)i; }
  const long long t0 = clock64();
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  const long long t1 = clock64();
  f32x16 s = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += acc[i];
  if (s[0] == 12345.f) sink[0] = s[1];
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <typename F> void run(const char* name, F launch, int nmfma, double flop) {
  long long* d; float* sink; hipMalloc(&d, 8); hipMalloc(&sink, 4);
  const int trips = 2000;
  launch(d, sink, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  launch(d, sink, trips);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
  printf("%-52s %7.1f clock64 ticks per MFMA (100 MHz ticks x24 = shader clocks at 2.4 GHz), %8.1f TF/s chip-wide\n", name, (double)c / trips / nmfma,
         flop * nmfma * trips * 256.0 * 4 / (ms * 1e-3) / 1e12);
}
int main() {
#define L16(NACC, LDS, TH, tag) run(tag, [](long long* d, float* s, int t) { hipLaunchKernelGGL((k16<NACC, LDS, TH>), dim3(256), dim3(TH), 0, 0, d, s, t); }, NACC, 16384.0 * (TH / 256))
#define L32(NACC, TH, tag) run(tag, [](long long* d, float* s, int t) { hipLaunchKernelGGL((k32<NACC, TH>), dim3(256), dim3(TH), 0, 0, d, s, t); }, NACC, 32768.0 * (TH / 256))
  L16(8, false, 256, "16x16x32, 8 acc (VGPR), 1 wave/SIMD");
  L16(48, false, 256, "16x16x32, 48 acc, 1 wave/SIMD");
  L16(48, true, 256, "16x16x32, 48 acc + ds_read per 6 MFMAs, 1 wave/SIMD");
  L16(48, false, 512, "16x16x32, 48 acc, 2 waves/SIMD");
  L16(24, false, 1024, "16x16x32, 24 acc, 4 waves/SIMD");
  L16(48, true, 512, "16x16x32, 48 acc + ds_read per 6 MFMAs, 2 waves/SIMD");
  L32(4, 256, "32x32x16, 4 acc, 1 wave/SIMD");
  L32(12, 256, "32x32x16, 12 acc, 1 wave/SIMD");
  L32(12, 512, "32x32x16, 12 acc, 2 waves/SIMD");
  return 0;
}
