// micro-benchmark: LDS atomic add throughput, float vs int, with the 4-lanes-per-address pattern of the dbias window
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float wf[2048];
  __shared__ int wi[2048];
  __shared__ unsigned long long wl[2048];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  for (int i = tid; i < 2048; i += 256) { wf[i] = 0.f; wi[i] = 0; wl[i] = 0; }
  __syncthreads();
  float v = 1.0f + tid * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int a = 64 + 4 * g - li + r + (it & 63) * 16 + (tid >> 6) * 0;
      if (MODE == 0) atomicAdd(&wf[a], v);
      else if (MODE == 1) atomicAdd(&wi[a], (int)(v * 1024.f));
      else if (MODE == 2) atomicAdd(&wf[(a * 4 + lane) & 2047], v);      // conflict-free float
      else if (MODE == 3) atomicAdd(&wi[(a * 4 + lane) & 2047], (int)(v * 1024.f));    // conflict-free int
      else atomicAdd(&wl[a], (unsigned long long)(long long)(v * 1099511627776.f));   // i64 4-way
    }
  }
  __syncthreads();
  if (tid < 4) out[blockIdx.x * 4 + tid] = wf[64 + tid] + wi[64 + tid] + (float)wl[64 + tid];
}
int main() {
  float* out; hipMalloc(&out, 4096 * 4 * 4);
  const int iters = 2000, blocks = 1024;
  for (int mode = 0; mode < 5; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters);
      if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * iters * 16;
    printf("mode %d (%s): %.3f ms, %.1f G lane-atomics/s, %.3f per clk per CU (2.4GHz, 256 CU)\n", mode,
           mode == 0 ? "f32 4-way same addr" : mode == 1 ? "i32 4-way same addr" : mode == 2 ? "f32 distinct" : mode == 3 ? "i32 distinct" : "i64 4-way same addr",
           ms, ops / ms / 1e6, ops / (ms * 1e-3) / 2.4e9 / 256);
  }
  return 0;
}
