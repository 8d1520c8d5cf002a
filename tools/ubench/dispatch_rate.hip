// How fast do the workgroups of ONE short kernel get onto the chip?  Every block records the 100 MHz reference clock at entry, then holds its CU for
// `hold` us (a decode-step skinny GEMM block lives ~3 us).  Printed per (grid, block size): when the n-th block (by entry order) started, relative
// to the first one -- i.e. whether a grid larger than the CU count is dispatched at once or the blocks past one-per-CU wait.
// build: hipcc --offload-arch=gfx950 -O3 dispatch_rate.hip -o dispatch_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int LDS, int V = 0>
__global__ __launch_bounds__(512) void k(unsigned long long* t, int hold_ticks, const float* src = nullptr, float* dst = nullptr) {
  __shared__ float pad[LDS / 4 > 0 ? LDS / 4 : 1];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  float r[V > 0 ? V : 1];
  if (V > 0) {          // V live registers across the hold (what a real kernel's operands in flight look like to the occupancy calculation)
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = src[threadIdx.x + 512 * i];
#pragma unroll
    for (int i = 0; i < V; ++i) asm volatile("" : "+v"(r[i]));
  }
  const int bid = blockIdx.y * gridDim.x + blockIdx.x, nblk = gridDim.x * gridDim.y;
  if (threadIdx.x == 0) { t[bid] = t0; pad[0] = 1.f; }
  __syncthreads();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)hold_ticks) __builtin_amdgcn_s_sleep(2);
  if (V > 0) {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) { asm volatile("" : "+v"(r[i])); a += r[i]; }
    if (a == 123.456f) dst[threadIdx.x] = a;
  }
  if (threadIdx.x == 0) t[nblk + bid] = __builtin_amdgcn_s_memrealtime();
}
int main() {
  unsigned long long* d; CK(hipMalloc(&d, 2 * 4096 * 8));
  std::vector<unsigned long long> h(2 * 4096);
  for (int threads : {256, 512}) for (int grid : {192, 256, 320, 512, 576, 768, 1024}) for (int hold : {100, 300}) {
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k<9216>, dim3(grid), dim3(threads), 0, 0, d, hold); }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, 2 * grid * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> e(h.begin(), h.begin() + grid); std::sort(e.begin(), e.end());
    unsigned long long last_exit = 0; for (int i = 0; i < grid; ++i) last_exit = std::max(last_exit, h[grid + i]);
    auto at = [&](int i) { return (e[std::min(i, grid - 1)] - e[0]) * 0.01; };
    printf("threads %3d grid %4d hold %.0f us: entry of block #64 %5.2f  #128 %5.2f  #192 %5.2f  #256 %5.2f  #320 %5.2f  #512 %5.2f  last %5.2f us after the first;  first entry -> last exit %5.2f us\n",
           threads, grid, hold * 0.01, at(63), at(127), at(191), at(255), at(319), at(511), at(grid - 1), (last_exit - e[0]) * 0.01);
  }
  // the same with 48 live registers per lane (a decode skinny GEMM has 57): do 576 blocks of 512 threads still start together?
  float* src; CK(hipMalloc(&src, 512 * 64 * 4)); CK(hipMemset(src, 0, 512 * 64 * 4));
  for (int grid : {512, 576, 768, 1024}) {
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL((k<9216, 48>), dim3(grid), dim3(512), 0, 0, d, 300, src, src); }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, 2 * grid * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> e(h.begin(), h.begin() + grid); std::sort(e.begin(), e.end());
    auto at = [&](int i) { return (e[std::min(i, grid - 1)] - e[0]) * 0.01; };
    printf("threads 512 grid %4d hold 3 us, 48 live VGPRs: entry of block #256 %5.2f  #512 %5.2f  #513 %5.2f  #576 %5.2f  #768 %5.2f  last %5.2f us after the first\n", grid, at(255), at(511), at(512), at(575), at(767), at(grid - 1));
  }
  for (int gy : {1, 4}) {       // a 2-D grid like the skinny GEMM's (N / 16 column blocks x 4 row fragments)
    const int grid = 576;
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL((k<9216, 48>), dim3(grid / gy, gy), dim3(512), 0, 0, d, 300, src, src); }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, 2 * grid * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> e(h.begin(), h.begin() + grid); std::sort(e.begin(), e.end());
    auto at = [&](int i) { return (e[std::min(i, grid - 1)] - e[0]) * 0.01; };
    printf("grid (%d, %d) x 512 threads, barrier, 48 live VGPRs: entry of block #256 %5.2f  #512 %5.2f  #513 %5.2f  #576 %5.2f us after the first\n", grid / gy, gy, at(255), at(511), at(512), at(575));
  }
  return 0;
}
