// Throughput of coalesced fp32 global atomic adds (fire-and-forget) by memory scope, gfx950.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_rate.hip -o atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int SCOPE>
__global__ __launch_bounds__(256) void k(float* dst, long n, int passes) {
  const long stride = (long)gridDim.x * 256;
  for (int p = 0; p < passes; ++p)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
      if (SCOPE == 0) __hip_atomic_fetch_add(dst + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (SCOPE == 1) __hip_atomic_fetch_add(dst + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else dst[i] += 1.0f;
    }
}
template <int SCOPE>
float run(float* d, long n, int passes, int blocks) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<SCOPE>, dim3(blocks), dim3(256), 0, 0, d, n, 1);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<SCOPE>, dim3(blocks), dim3(256), 0, 0, d, n, passes);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  const long n = 24576000;  // 98 MB of fp32 (32000 x 768)
  float* d; hipMalloc(&d, n * 4); hipMemset(d, 0, n * 4);
  for (int blocks : {2048, 8192}) {
    const int passes = 8;
    float t0 = run<0>(d, n, passes, blocks), t1 = run<1>(d, n, passes, blocks), t2 = run<2>(d, n, passes, blocks);
    printf("blocks %5d: agent-scope atomics %7.1f us/pass (%5.2f TB/s of operand bytes), workgroup-scope %7.1f us (%5.2f), plain RMW %7.1f us (%5.2f)\n",
           blocks, t0 * 1e3 / passes, n * 4.0 / (t0 / passes * 1e-3) / 1e12, t1 * 1e3 / passes, n * 4.0 / (t1 / passes * 1e-3) / 1e12,
           t2 * 1e3 / passes, n * 4.0 / (t2 / passes * 1e-3) / 1e12);
  }
  return 0;
}
