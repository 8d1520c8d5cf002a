// Optimiser kernels over the flat fp32 parameter arena (gfx950, HBM-bound, float4 per lane):
//   global grad-norm (torch clip_grad_norm_, dvc.py:114-115), Adam (torch.optim.Adam, dvc.py:346-351,116)
//   with the clip coefficient folded in and the bf16 shadow weights refreshed in the same pass,
//   fp32->bf16 cast, and the time-token embedding renormalisation of dvc.py:118-126.
// Algorithmic bytes per Adam step: read g,p,m,v + write p,m,v (7 x 4 B) + 2 B bf16 shadow per parameter.
#include <math.h>
#include "v2s_common.h"

namespace {

constexpr int SQ_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ partial) {
  __shared__ float red[4];
  const long n4 = n >> 2;
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0) {
    for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(1024) void sqnorm_final_kernel(const float* __restrict__ partial, int nb, float* __restrict__ out) {
  __shared__ float a[1024];
  a[threadIdx.x] = threadIdx.x < nb ? partial[threadIdx.x] : 0.f;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) a[threadIdx.x] += a[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out += a[0];
}

struct AdamP {
  float* p; float* m; float* v; const float* g; bf16_t* pb;
  long n;
  float lr_bc1, beta1, beta2, eps, wd, inv_sqrt_bc2;
  const float* gnorm_sq; float max_norm, grad_scale;
  const float* hyper;    // device {lr / bias-correction-1, 1 / sqrt(bias-correction-2)} overriding the by-value pair (graph replay) or NULL
};

__device__ __forceinline__ void adam1(float& p, float& m, float& v, float g, const AdamP& a, float coef) {
  g = g * coef;
  if (a.wd != 0.f) g += a.wd * p;
  m = a.beta1 * m + (1.f - a.beta1) * g;
  v = a.beta2 * v + (1.f - a.beta2) * g * g;
  const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
  p -= a.lr_bc1 * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(AdamP a) {
  if (a.hyper) { a.lr_bc1 = a.hyper[0]; a.inv_sqrt_bc2 = a.hyper[1]; }
  float coef = a.grad_scale;
  if (a.gnorm_sq && a.max_norm > 0.f) {
    const float gn = sqrtf(a.gnorm_sq[0]) * a.grad_scale;          // norm of the scaled gradient
    coef *= fminf(1.f, a.max_norm / (gn + 1e-6f));                  // torch clip_grad_norm_
  }
  const long n4 = a.n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 p = reinterpret_cast<float4*>(a.p)[i], m = reinterpret_cast<float4*>(a.m)[i], v = reinterpret_cast<float4*>(a.v)[i];
    const float4 g = reinterpret_cast<const float4*>(a.g)[i];
    adam1(p.x, m.x, v.x, g.x, a, coef); adam1(p.y, m.y, v.y, g.y, a, coef);
    adam1(p.z, m.z, v.z, g.z, a, coef); adam1(p.w, m.w, v.w, g.w, a, coef);
    reinterpret_cast<float4*>(a.p)[i] = p; reinterpret_cast<float4*>(a.m)[i] = m; reinterpret_cast<float4*>(a.v)[i] = v;
    if (a.pb) {
      uint2 w; w.x = pack2bf(p.x, p.y); w.y = pack2bf(p.z, p.w);
      reinterpret_cast<uint2*>(a.pb)[i] = w;
    }
  }
  if (blockIdx.x == 0) {
    for (long i = n4 * 4 + threadIdx.x; i < a.n; i += 256) {
      float p = a.p[i], m = a.m[i], v = a.v[i];
      adam1(p, m, v, a.g[i], a, coef);
      a.p[i] = p; a.m[i] = m; a.v[i] = v;
      if (a.pb) a.pb[i] = f2bf(p);
    }
  }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 p = reinterpret_cast<const float4*>(src)[i];
    uint2 w; w.x = pack2bf(p.x, p.y); w.y = pack2bf(p.z, p.w);
    reinterpret_cast<uint2*>(dst)[i] = w;
  }
  if (blockIdx.x == 0)
    for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) dst[i] = f2bf(src[i]);
}

// one wavefront per embedding row: ws[2 + row] = ||row||_2
__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ emb, int V, int d, float* __restrict__ ws) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= V) return;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) { const float x = emb[(long)row * d + c]; s += x * x; }
  s = wave_sum(s);
  if (lane == 0) ws[2 + row] = sqrtf(s);
}
// one wavefront per embedding row: sumsq[row] += sum of squares of the row's elements whose FLAT index (row * d + c) lies in [f0, f1).
// Sharded optimizer: a rank's fp32 master values are current only inside the stripes it owns, so the frozen-row norms of the
// time-token renorm are assembled from per-rank partial sums (all-reduced by the caller) instead of read from stale masters.
__global__ __launch_bounds__(256) void rowsumsq_range_kernel(const float* __restrict__ emb, int row_lo, int row_hi, int d, long f0, long f1,
                                                            float* __restrict__ sumsq) {
  const int row = row_lo + blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= row_hi) return;
  const long base = (long)row * d;
  const int c0 = (int)max(0L, f0 - base), c1 = (int)min((long)d, f1 - base);
  float s = 0.f;
  for (int c = c0 + lane; c < c1; c += 64) { const float x = emb[base + c]; s += x * x; }
  s = wave_sum(s);
  if (lane == 0 && c1 > c0) sumsq[row] += s;
}
// ws[2 + row] = sqrt(text_sumsq[row]) for the frozen rows (the time-token rows are whole on every rank: rownorm_kernel on them)
__global__ __launch_bounds__(256) void rownorm_from_sumsq_kernel(const float* __restrict__ text_sumsq, int n, float* __restrict__ ws) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < n) ws[2 + r] = sqrtf(text_sumsq[r]);
}
__global__ __launch_bounds__(256) void rownorm_rows_kernel(const float* __restrict__ emb, int row_lo, int row_hi, int d, float* __restrict__ ws) {
  const int row = row_lo + blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= row_hi) return;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) { const float x = emb[(long)row * d + c]; s += x * x; }
  s = wave_sum(s);
  if (lane == 0) ws[2 + row] = sqrtf(s);
}
__global__ __launch_bounds__(1024) void renorm_means_kernel(float* __restrict__ ws, int V, int nb) {
  __shared__ float a[1024], b[1024];
  float sa = 0.f, sb = 0.f;
  for (int r = threadIdx.x; r < V; r += 1024) { if (r < V - nb) sa += ws[2 + r]; else sb += ws[2 + r]; }
  a[threadIdx.x] = sa; b[threadIdx.x] = sb;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) { a[threadIdx.x] += a[threadIdx.x + o]; b[threadIdx.x] += b[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { ws[0] = a[0] / (V - nb); ws[1] = b[0] / nb; }
}
__global__ __launch_bounds__(256) void renorm_scale_kernel(float* __restrict__ emb, bf16_t* __restrict__ embb, int V, int d, int nb,
                                                           const float* __restrict__ ws) {
  const float div = ws[1] / ws[0];   // trainable mean norm / frozen mean norm (dvc.py:122)
  const long base = (long)(V - nb) * d, n = (long)nb * d;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float x = emb[base + i] / div;
    emb[base + i] = x;
    if (embb) embb[base + i] = f2bf(x);
  }
}

}  // namespace

extern "C" int v2s_sqnorm(const float* g, int64_t n, float* partial_ws, float* out_sum, void* stream) {
  V2S_CHECK(g && partial_ws && out_sum && n > 0, V2S_ERR_ARG, "v2s_sqnorm: bad args");
  V2S_CHECK(((uintptr_t)g & 15) == 0, V2S_ERR_ALIGN, "v2s_sqnorm: g must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(SQ_BLOCKS), dim3(256), 0, s, g, (long)n, partial_ws);
  V2S_LAUNCH_CHECK();
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(1024), 0, s, partial_ws, SQ_BLOCKS, out_sum);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_adam_step(const v2s_adam_args* a, void* stream) {
  V2S_CHECK(a && a->p && a->m && a->v && a->g && a->n > 0 && a->step >= 1, V2S_ERR_ARG, "v2s_adam_step: bad args");
  V2S_CHECK((((uintptr_t)a->p | (uintptr_t)a->m | (uintptr_t)a->v | (uintptr_t)a->g) & 15) == 0 && (((uintptr_t)a->p_bf16) & 7) == 0,
            V2S_ERR_ALIGN, "v2s_adam_step: buffers must be 16-byte aligned");
  AdamP p;
  p.p = a->p; p.m = a->m; p.v = a->v; p.g = a->g; p.pb = (bf16_t*)a->p_bf16; p.n = a->n;
  const double bc1 = 1.0 - pow((double)a->beta1, (double)a->step), bc2 = 1.0 - pow((double)a->beta2, (double)a->step);
  p.lr_bc1 = (float)((double)a->lr / bc1);
  p.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  p.beta1 = a->beta1; p.beta2 = a->beta2; p.eps = a->eps; p.wd = a->weight_decay;
  p.gnorm_sq = a->gnorm_sq; p.max_norm = a->max_norm; p.grad_scale = a->grad_scale == 0.f ? 1.f : a->grad_scale;
  p.hyper = a->hyper_dev;
  hipLaunchKernelGGL(adam_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, p);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_cast_bf16(const float* src, void* dst, int64_t n, void* stream) {
  V2S_CHECK(src && dst && n > 0, V2S_ERR_ARG, "v2s_cast_bf16: bad args");
  V2S_CHECK(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0, V2S_ERR_ALIGN, "v2s_cast_bf16: misaligned");
  long blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cast_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, (long)n);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_timetoken_renorm(float* emb, void* emb_bf16, int32_t V, int32_t d, int32_t num_bins, float* ws,
                                    void* stream) {
  V2S_CHECK(emb && ws && V > num_bins && num_bins > 0 && d > 0, V2S_ERR_ARG, "v2s_timetoken_renorm: bad args V=%d bins=%d", V, num_bins);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(rownorm_kernel, dim3((V + 3) / 4), dim3(256), 0, s, emb, V, d, ws);
  V2S_LAUNCH_CHECK();
  hipLaunchKernelGGL(renorm_means_kernel, dim3(1), dim3(1024), 0, s, ws, V, num_bins);
  V2S_LAUNCH_CHECK();
  hipLaunchKernelGGL(renorm_scale_kernel, dim3((num_bins * d + 255) / 256), dim3(256), 0, s, emb, (bf16_t*)emb_bf16, V, d, num_bins, ws);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_rowsumsq_range(const float* emb, int32_t V, int32_t d, int64_t f0, int64_t f1, float* sumsq, void* stream) {
  V2S_CHECK(emb && sumsq && V > 0 && d > 0 && f0 >= 0 && f1 <= (int64_t)V * d, V2S_ERR_ARG, "v2s_rowsumsq_range: bad args V=%d d=%d range [%ld, %ld)", V, d, (long)f0, (long)f1);
  if (f1 <= f0) return V2S_OK;
  const int row_lo = (int)(f0 / d), row_hi = (int)((f1 - 1) / d) + 1;
  hipLaunchKernelGGL(rowsumsq_range_kernel, dim3((row_hi - row_lo + 3) / 4), dim3(256), 0, (hipStream_t)stream, emb, row_lo, row_hi, d, (long)f0, (long)f1, sumsq);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_timetoken_renorm_sq(float* emb, void* emb_bf16, int32_t V, int32_t d, int32_t num_bins, const float* text_sumsq, float* ws,
                                       void* stream) {
  V2S_CHECK(emb && ws && text_sumsq && V > num_bins && num_bins > 0 && d > 0, V2S_ERR_ARG, "v2s_timetoken_renorm_sq: bad args V=%d bins=%d", V, num_bins);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(rownorm_from_sumsq_kernel, dim3((V - num_bins + 255) / 256), dim3(256), 0, s, text_sumsq, V - num_bins, ws);
  V2S_LAUNCH_CHECK();
  hipLaunchKernelGGL(rownorm_rows_kernel, dim3((num_bins + 3) / 4), dim3(256), 0, s, emb, V - num_bins, V, d, ws);
  V2S_LAUNCH_CHECK();
  hipLaunchKernelGGL(renorm_means_kernel, dim3(1), dim3(1024), 0, s, ws, V, num_bins);
  V2S_LAUNCH_CHECK();
  hipLaunchKernelGGL(renorm_scale_kernel, dim3((num_bins * d + 255) / 256), dim3(256), 0, s, emb, (bf16_t*)emb_bf16, V, d, num_bins, ws);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}
