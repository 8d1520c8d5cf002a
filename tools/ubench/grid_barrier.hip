// What a decode step's ~110 dependent launches could cost as ONE persistent kernel, gfx950: (1) a grid-wide barrier (monotonic counter in
// device memory, agent scope) at 64..512 co-resident blocks; (2) a chain of dependent phases -- every block reads 4 KB of "weights" that no
// cache holds, reduces them with a value another block wrote in the previous phase, writes one value, barrier -- with the next phase's
// weights requested after or BEFORE the barrier; (3) the same chain as separate launches: eager, and as a replayed hipGraph (device time per
// node, host time per hipGraphLaunch).  Every spin is bounded (err flag) so that a lost block cannot hang the GPU.
// build: hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void grid_arrive(unsigned* ctr) {
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void grid_wait(unsigned* ctr, unsigned target, int* err) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1u << 21)) { *err = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void grid_sync(unsigned* ctr, unsigned target, int* err) { grid_arrive(ctr); grid_wait(ctr, target, err); }

__global__ __launch_bounds__(256) void barrier_only(unsigned* ctr, int iters, int* err) {
  for (int i = 0; i < iters; ++i) grid_sync(ctr, (unsigned)(i + 1) * gridDim.x, err);
}

// phase p: block b sums its 1024 floats of W[p][b] with out[p - 1][(b + 1) % G] and writes out[p][b]
template <int PREFETCH>
__global__ __launch_bounds__(256) void chain(const float4* __restrict__ W, float* out, unsigned* ctr, int phases, int* err) {
  __shared__ float red[4];
  const int G = gridDim.x, b = blockIdx.x, t = threadIdx.x;
  float4 w = W[((long)0 * G + b) * 256 + t];
  for (int p = 0; p < phases; ++p) {
    float4 wn = w;
    if (p > 0) grid_arrive(ctr);
    if (PREFETCH && p + 1 < phases) wn = W[((long)(p + 1) * G + b) * 256 + t];        // requested between arriving at the barrier and waiting on it
    if (p > 0) grid_wait(ctr, (unsigned)p * G, err);
    if (!PREFETCH && p > 0) w = W[((long)p * G + b) * 256 + t];                       // requested after the barrier
    float s = w.x + w.y + w.z + w.w;
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((t & 63) == 0) red[t >> 6] = s;
    __syncthreads();
    if (t == 0) {
      const float prev = p > 0 ? __hip_atomic_load(out + (long)(p - 1) * G + (b + 1) % G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
      __hip_atomic_store(out + (long)p * G + b, red[0] + red[1] + red[2] + red[3] + prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (PREFETCH) w = wn;
  }
}

// the same phase as its own launch
__global__ __launch_bounds__(256) void phase_kernel(const float4* __restrict__ W, float* out, int p) {
  __shared__ float red[4];
  const int G = gridDim.x, b = blockIdx.x, t = threadIdx.x;
  const float4 w = W[((long)p * G + b) * 256 + t];
  float s = w.x + w.y + w.z + w.w;
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  if (t == 0) out[(long)p * G + b] = red[0] + red[1] + red[2] + red[3] + (p > 0 ? out[(long)(p - 1) * G + (b + 1) % G] : 0.f);
}

int main() {
  const int PH = 110;                  // phases = launches of a decode step
  const int GMAX = 512;
  unsigned* ctr; int* err; float4* W; float* out;
  CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  CK(hipMalloc(&W, (size_t)PH * GMAX * 4096)); CK(hipMemset(W, 0, (size_t)PH * GMAX * 4096));      // 230 MB
  CK(hipMalloc(&out, (size_t)PH * GMAX * 4));
  float* flush; const size_t FL = 512u << 20; CK(hipMalloc(&flush, FL));                           // evicts W from L2 / MALL between runs
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipStream_t s; CK(hipStreamCreate(&s));
  float ms;
  for (int G : {64, 128, 256, 512}) {
    const int iters = 2000;
    CK(hipMemsetAsync(ctr, 0, 4, s));
    hipLaunchKernelGGL(barrier_only, dim3(G), dim3(256), 0, s, ctr, 10, err);
    CK(hipMemsetAsync(ctr, 0, 4, s));
    CK(hipEventRecord(a, s));
    hipLaunchKernelGGL(barrier_only, dim3(G), dim3(256), 0, s, ctr, iters, err);
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    printf("grid barrier, %3d blocks x 256 threads: %6.2f us per barrier\n", G, ms * 1e3 / iters);
  }
  for (int G : {64, 256}) {
    float t[2];
    for (int pre = 0; pre < 2; ++pre) {
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(flush, rep, FL, s));
        CK(hipMemsetAsync(ctr, 0, 4, s));
        CK(hipEventRecord(a, s));
        if (pre) hipLaunchKernelGGL(chain<1>, dim3(G), dim3(256), 0, s, W, out, ctr, PH, err);
        else hipLaunchKernelGGL(chain<0>, dim3(G), dim3(256), 0, s, W, out, ctr, PH, err);
        CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
      }
      t[pre] = best;
    }
    // separate launches, eager
    float eager = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemsetAsync(flush, rep, FL, s));
      CK(hipEventRecord(a, s));
      for (int p = 0; p < PH; ++p) hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(256), 0, s, W, out, p);
      CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
      eager = ms < eager ? ms : eager;
    }
    // the same launches as a replayed graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int p = 0; p < PH; ++p) hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(256), 0, s, W, out, p);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    float graph = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemsetAsync(flush, rep, FL, s));
      CK(hipEventRecord(a, s));
      CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
      graph = ms < graph ? ms : graph;
    }
    CK(hipStreamSynchronize(s));
    const int NL = 20;
    auto h0 = std::chrono::steady_clock::now();
    for (int i = 0; i < NL; ++i) CK(hipGraphLaunch(ge, s));
    auto h1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    auto h2 = std::chrono::steady_clock::now();
    const double host_us = std::chrono::duration<double, std::micro>(h1 - h0).count() / NL;
    const double all_us = std::chrono::duration<double, std::micro>(h2 - h0).count() / NL;
    printf("%3d blocks, %d dependent phases (4 KB of cold weights per block and phase): persistent kernel %6.2f us per phase (weights requested after "
           "the barrier), %6.2f (before); separate launches %6.2f eager, %6.2f in a replayed graph; hipGraphLaunch host time %6.1f us per replay = "
           "%4.2f us per node, %d replays back to back %6.1f us each\n", G, PH, t[0] * 1e3 / PH, t[1] * 1e3 / PH, eager * 1e3 / PH, graph * 1e3 / PH,
           host_us, host_us / PH, NL, all_us);
    // two independent chains: two graphs replayed on two streams at once, and the same launches interleaved eagerly on the two streams
    {
      hipStream_t s2; CK(hipStreamCreate(&s2));
      float* out2; CK(hipMalloc(&out2, (size_t)PH * GMAX * 4));
      hipGraph_t g2; hipGraphExec_t ge2;
      CK(hipStreamBeginCapture(s2, hipStreamCaptureModeThreadLocal));
      for (int p = 0; p < PH; ++p) hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(256), 0, s2, W + (size_t)PH * G * 256, out2, p);
      CK(hipStreamEndCapture(s2, &g2));
      CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge2, s2)); CK(hipStreamSynchronize(s2));
      hipEvent_t c; CK(hipEventCreate(&c));
      float two_graph = 1e9, two_eager = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(flush, rep, FL, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s)); CK(hipStreamWaitEvent(s2, a, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipGraphLaunch(ge2, s2));
        CK(hipEventRecord(c, s2)); CK(hipStreamWaitEvent(s, c, 0));
        CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        two_graph = ms < two_graph ? ms : two_graph;
      }
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(flush, rep, FL, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s)); CK(hipStreamWaitEvent(s2, a, 0));
        for (int p = 0; p < PH; ++p) {
          hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(256), 0, s, W, out, p);
          hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(256), 0, s2, W + (size_t)PH * G * 256, out2, p);
        }
        CK(hipEventRecord(c, s2)); CK(hipStreamWaitEvent(s, c, 0));
        CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        two_eager = ms < two_eager ? ms : two_eager;
      }
      printf("%3d blocks: TWO independent chains of %d launches on two streams: %6.1f us as two replayed graphs, %6.1f us as interleaved eager launches "
             "(one chain alone: %6.1f us in a graph, %6.1f eager)\n", G, PH, two_graph * 1e3, two_eager * 1e3, graph * 1e3, eager * 1e3);
      CK(hipGraphExecDestroy(ge2)); CK(hipGraphDestroy(g2)); CK(hipFree(out2)); CK(hipStreamDestroy(s2));
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  printf("spin limit hit: %s\n", herr ? "YES (numbers invalid)" : "no");
  return 0;
}
