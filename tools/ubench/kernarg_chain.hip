// Does it matter HOW a short dependent kernel gets its arguments?  A replayed hipGraph of 1000 dependent launches of a decode-sized kernel
// (192 blocks x 512 threads; every thread reads 16 bytes through a pointer argument, the block reduces and writes one value), with the arguments
// (a) in one 128-byte struct passed by value (what the library's GEMM / attention / decode kernels do: the compiler passes it by reference and
// s_loads the fields), (b) as leading scalar parameters.  Build the file twice -- plain, and with -mllvm -amdgpu-kernarg-preload-count=16 (the
// first 16 dwords of (b) then arrive in SGPRs with the wave, (a) is not preloadable) -- and compare the device time per node.
// build: hipcc --offload-arch=gfx950 -O3 kernarg_chain.hip -o kernarg_chain ; hipcc ... -mllvm -amdgpu-kernarg-preload-count=16 -o kernarg_chain_preload
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
struct P { const uint4* A; const uint4* B; float* C; const float* R; int M, N, K, lda, ldb, ldc, ldr, flags; float alpha, eps; long pad[8]; };
__device__ __forceinline__ void body(const uint4* A, const uint4* B, float* C, const float* R, int K, float alpha) {
  const uint4 a = A[(blockIdx.x * 512 + threadIdx.x) % K], b = B[(blockIdx.x * 512 + threadIdx.x)];
  float v = __uint_as_float(a.x ^ b.x) * alpha + R[blockIdx.x];
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
  __shared__ float s[8];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) C[blockIdx.x] = s[0] + s[1] + s[2] + s[3] + s[4] + s[5] + s[6] + s[7];
}
__global__ __launch_bounds__(512) void k_struct(const P p) { body(p.A, p.B, p.C, p.R, p.K, p.alpha); }
__global__ __launch_bounds__(512) void k_flat(const uint4* A, const uint4* B, float* C, const float* R, int K, float alpha, int M, int N, int lda, int ldb, int ldc, int ldr) {
  body(A, B, C, R, K, alpha);
}
int main() {
  const int blocks = 192, nodes = 1000;
  uint4 *A, *B; float *C0, *C1;
  CK(hipMalloc(&A, 6144 * 16)); CK(hipMalloc(&B, (size_t)blocks * 512 * 16)); CK(hipMalloc(&C0, blocks * 4)); CK(hipMalloc(&C1, blocks * 4));
  CK(hipMemset(A, 0, 6144 * 16)); CK(hipMemset(B, 0, (size_t)blocks * 512 * 16)); CK(hipMemset(C0, 0, blocks * 4)); CK(hipMemset(C1, 0, blocks * 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int variant = 0; variant < 2; ++variant) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < nodes; ++i) {
      float* src = (i & 1) ? C1 : C0; float* dst = (i & 1) ? C0 : C1;        // node i reads what node i - 1 wrote
      if (variant == 0) { P p{}; p.A = A; p.B = B; p.C = dst; p.R = src; p.K = 6144; p.alpha = 0.5f; hipLaunchKernelGGL(k_struct, dim3(blocks), dim3(512), 0, s, p); }
      else hipLaunchKernelGGL(k_flat, dim3(blocks), dim3(512), 0, s, A, B, dst, src, 6144, 0.5f, 64, 768, 768, 768, 768, 768);
    }
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    float best = 1e9f;
    for (int r = 0; r < 10; ++r) {
      CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%s: %.3f us per dependent node (best of 10 replays of %d nodes)\n", variant == 0 ? "struct by value " : "leading scalars ", best * 1000.f / nodes, nodes);
  }
  return 0;
}
