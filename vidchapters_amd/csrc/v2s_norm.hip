// Row normalisations for gfx950: one 64-lane wavefront per row, 16-byte bf16 loads, shuffle reductions,
// fp32 statistics.  HBM-bound (reads x once, writes y once).
//   RMSNorm  : model/modeling_t5.py:263-277 (T5LayerNorm.forward)
//   LayerNorm: torch nn.LayerNorm as used at model/vit.py:64,69,99 (eps 1e-5, affine)
// Backward produces dx (+ fused residual-gradient add); dw/db are reduced per wave in registers, per block in LDS and
// added to the caller's fp32 gradient with one hardware float atomic per column per block.  Optionally it also writes
// dropout(dx) -- the operand the NEXT sublayer of the backward pass starts from (the reference drops the sublayer output before the
// residual add, modeling_t5.py:618,654,353; vit.py:54,21) -- so that the residual-stream gradient is not re-read by a separate
// elementwise launch: same mask and bit-identical values as v2s_dropout applied to the stored dx.
#include "v2s_common.h"

namespace {

constexpr int MAXC = 4;  // chunks of 8 columns per lane -> cols <= 2048
// Grid of the backward kernels (grid-stride over rows).  Every block ends with one float atomic per column, and those are what the launch waits for at its
// end: 1024 blocks of 4 waves ran a 32000 x 768 RMSNorm backward in 47.9 us, 2048 in 62, 384 blocks of 8 waves (as many rows in flight, a third of the
// atomics) in 32.5 us = 6.05 TB/s of its four tensors (profiles/r06_norm_bwd_grid.txt; -DNORM_BWD_BLOCKS / -DNORM_BWD_WAVES for A/B builds)
#ifndef NORM_BWD_BLOCKS
#define NORM_BWD_BLOCKS 384
#endif
constexpr int BWD_BLOCKS = NORM_BWD_BLOCKS;
#ifndef NORM_BWD_WAVES
#define NORM_BWD_WAVES 8          // waves (= rows in flight) per backward block
#endif
constexpr int BW = NORM_BWD_WAVES;

// activation I/O type: bf16 (the product path) or fp32 (option "fp32_io": debug mode that takes the bf16 rounding of the
// activations out of the comparison with an fp32 reference -- SURVEY 8c asks for <= 1e-4 there)
__device__ __forceinline__ void ld8(const bf16_t* p, float* f) { unpack8(*reinterpret_cast<const uint4*>(p), f); }
__device__ __forceinline__ void ld8(const float* p, float* f) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
__device__ __forceinline__ void st8(bf16_t* p, const float* f) { *reinterpret_cast<uint4*>(p) = pack8(f); }
__device__ __forceinline__ void st8(float* p, const float* f) {
  *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
}
// a row chunk as loaded (unpacked after every load of the row has been issued)
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
  uint4 v = make_uint4(0, 0, 0, 0);
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void unpack(float* f) const { unpack8(v, f); }
};
template <> struct Raw8<float> {
  float4 a = make_float4(0, 0, 0, 0), b = make_float4(0, 0, 0, 0);
  __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
  __device__ __forceinline__ void unpack(float* f) const { f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w; }
};
__device__ __forceinline__ void round8(bf16_t*, float* f) { const uint4 pk = pack8(f); unpack8(pk, f); }   // the value the bf16 store holds
__device__ __forceinline__ void round8(float*, float*) {}

template <bool LN, typename T = bf16_t>
__global__ __launch_bounds__(256) void norm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, T* __restrict__ y,
                                                       float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                       int rows, int cols, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = cols >> 3;
  const T* xr = x + (long)row * cols;
  float v[MAXC][8];
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
      ld8(xr + c * 8, v[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s += v[i][j]; ss += v[i][j] * v[i][j]; }
    }
  }
  float mean = 0.f, rstd;
  if (LN) {
    mean = wave_sum(s) / cols;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + i * 64;
      if (c < nch) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; var += d * d; }
      }
    }
    rstd = rsqrtf(wave_sum(var) / cols + eps);
  } else {
    rstd = rsqrtf(wave_sum(ss) / cols + eps);
  }
  if (lane == 0) {
    rstd_out[row] = rstd;
    if (LN) mean_out[row] = mean;
  }
  T* yr = y + (long)row * cols;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
      float o[8];
      const float4 w0 = *reinterpret_cast<const float4*>(w + c * 8), w1 = *reinterpret_cast<const float4*>(w + c * 8 + 4);
      const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
      if (LN) {
        const float4 b0 = *reinterpret_cast<const float4*>(b + c * 8), b1 = *reinterpret_cast<const float4*>(b + c * 8 + 4);
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * wv[j] + bv[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = wv[j] * (v[i][j] * rstd);
      }
      st8(yr + c * 8, o);
    }
  }
}

template <bool LN, int NCH, typename T = bf16_t>
__global__ __launch_bounds__(BW * 64) void norm_bwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                       const T* __restrict__ dy, T* __restrict__ dx,
                                                       const T* __restrict__ dx_add, float* __restrict__ dw_out,
                                                       float* __restrict__ db_out, int rows, int cols, T* __restrict__ dx_drop,
                                                       uint32_t p16, float inv_keep, uint32_t seed, const uint32_t* __restrict__ salt) {
  __shared__ float red[BW][NCH * 512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = cols >> 3;
  float dwacc[NCH][8], dbacc[NCH][8];
  float wv[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
#pragma unroll
    for (int j = 0; j < 8; ++j) { dwacc[i][j] = 0.f; dbacc[i][j] = 0.f; wv[i][j] = 0.f; }
    if (c < nch) {
      const float4 w0 = *reinterpret_cast<const float4*>(w + c * 8), w1 = *reinterpret_cast<const float4*>(w + c * 8 + 4);
      wv[i][0] = w0.x; wv[i][1] = w0.y; wv[i][2] = w0.z; wv[i][3] = w0.w;
      wv[i][4] = w1.x; wv[i][5] = w1.y; wv[i][6] = w1.z; wv[i][7] = w1.w;
    }
  }
  for (int row = blockIdx.x * BW + wave; row < rows; row += gridDim.x * BW) {
    const float rstd = rstd_in[row];
    const float mean = LN ? mean_in[row] : 0.f;
    float xh[NCH][8], g[NCH][8];
    float sg = 0.f, sgx = 0.f;
    // all global loads of the row -- including the residual-gradient operand that is only needed after the reduction -- are issued
    // up front: serialised behind the wave reduction they cost 11 % (54 -> 48 us at 32000 x 768); prefetching the next row on top
    // of that gained nothing (not latency-bound any more)
    Raw8<T> xr[NCH], dr[NCH], ar[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nch) {
        xr[i].load(x + (long)row * cols + c * 8);
        dr[i].load(dy + (long)row * cols + c * 8);
        if (dx_add) ar[i].load(dx_add + (long)row * cols + c * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nch) {
        float xv[8], dv[8];
        xr[i].unpack(xv);
        dr[i].unpack(dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          xh[i][j] = (xv[j] - mean) * rstd;
          g[i][j] = dv[j] * wv[i][j];
          sg += g[i][j];
          sgx += g[i][j] * xh[i][j];
          dwacc[i][j] += dv[j] * xh[i][j];
          dbacc[i][j] += dv[j];
        }
      }
    }
    sgx = wave_sum(sgx) / cols;
    sg = LN ? wave_sum(sg) / cols : 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nch) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rstd * (g[i][j] - sg - xh[i][j] * sgx);
        if (dx_add) {
          float a[8];
          ar[i].unpack(a);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += a[j];
        }
        st8(dx + (long)row * cols + c * 8, o);
        if (dx_drop) {                      // dropout of the STORED (rounded) value: identical to v2s_dropout(dx)
          round8(dx, o);
          v2s_drop8(o, (unsigned long long)row * (unsigned long long)cols + (unsigned long long)(c * 8), v2s_salted(seed, salt), p16, inv_keep);
          st8(dx_drop + (long)row * cols + c * 8, o);
        }
      }
    }
  }
  // block reduction of dw (and db) over the 4 waves, then one partial row per block
  for (int pass = 0; pass < (LN ? 2 : 1); ++pass) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nch) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wave][c * 8 + j] = pass ? dbacc[i][j] : dwacc[i][j];
      }
    }
    __syncthreads();
    float* dst = pass ? db_out : dw_out;
    for (int c = threadIdx.x; c < cols; c += BW * 64) {
      float t = red[0][c];
#pragma unroll
      for (int w = 1; w < BW; ++w) t += red[w][c];
      atomicAdd(dst + c, t);
    }
  }
}

int bwd_blocks(int rows) { return (rows + BW - 1) / BW < BWD_BLOCKS ? (rows + BW - 1) / BW : BWD_BLOCKS; }

int check_shape(const char* who, int rows, int cols) {
  if (rows <= 0 || cols <= 0 || (cols % 8) != 0 || cols > 8 * 64 * MAXC) {
    v2s_set_error("%s: unsupported shape rows=%d cols=%d (cols must be a multiple of 8, <= %d)", who, rows, cols, 8 * 64 * MAXC);
    return V2S_ERR_SHAPE;
  }
  return V2S_OK;
}

}  // namespace

extern "C" int v2s_rmsnorm_fwd(const void* x, const float* w, void* y, float* rstd, int32_t rows, int32_t cols,
                               float eps, void* stream) {
  if (int e = check_shape("v2s_rmsnorm_fwd", rows, cols)) return e;
  if (v2s_opt_fp32_io())
    hipLaunchKernelGGL((norm_fwd_kernel<false, float>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)x, w,
                       (const float*)nullptr, (float*)y, (float*)nullptr, rstd, rows, cols, eps);
  else
  hipLaunchKernelGGL((norm_fwd_kernel<false>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, w,
                     (const float*)nullptr, (bf16_t*)y, (float*)nullptr, rstd, rows, cols, eps);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_layernorm_fwd(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                                 int32_t rows, int32_t cols, float eps, void* stream) {
  if (int e = check_shape("v2s_layernorm_fwd", rows, cols)) return e;
  if (v2s_opt_fp32_io())
    hipLaunchKernelGGL((norm_fwd_kernel<true, float>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)x, w, b,
                       (float*)y, mean, rstd, rows, cols, eps);
  else
  hipLaunchKernelGGL((norm_fwd_kernel<true>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, w, b,
                     (bf16_t*)y, mean, rstd, rows, cols, eps);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

// shared launcher of the four backward entry points
template <bool LN>
static int norm_bwd_launch(const char* who, const void* x, const float* w, const float* mean, const float* rstd, const void* dy, void* dx,
                           const void* dx_add, float* dw, float* db, int32_t rows, int32_t cols, void* dx_drop, float dropout_p,
                           uint32_t dropout_seed, void* stream) {
  if (int e = check_shape(who, rows, cols)) return e;
  V2S_CHECK(dropout_p >= 0.f && dropout_p < 1.f, V2S_ERR_ARG, "%s: dropout_p out of range", who);
  const int nb = bwd_blocks(rows);
  hipStream_t s = (hipStream_t)stream;
  const uint32_t p16 = (uint32_t)(dropout_p * 65536.0f + 0.5f);
  const float inv_keep = p16 ? 1.0f / (1.0f - (float)p16 / 65536.0f) : 1.0f;
  bf16_t* dd = (dx_drop && p16) ? (bf16_t*)dx_drop : nullptr;
  V2S_CHECK(!(dx_drop && !p16), V2S_ERR_ARG, "%s: dx_drop needs dropout_p > 0", who);
  if (v2s_opt_fp32_io()) {          // debug mode: fp32 activations in and out
    if (cols <= 1024)
      hipLaunchKernelGGL((norm_bwd_kernel<LN, 2, float>), dim3(nb), dim3(BW * 64), 0, s, (const float*)x, w, mean, rstd, (const float*)dy, (float*)dx,
                         (const float*)dx_add, dw, db, rows, cols, (float*)dd, p16, inv_keep, dropout_seed, v2s_seed_salt());
    else
      hipLaunchKernelGGL((norm_bwd_kernel<LN, 4, float>), dim3(nb), dim3(BW * 64), 0, s, (const float*)x, w, mean, rstd, (const float*)dy, (float*)dx,
                         (const float*)dx_add, dw, db, rows, cols, (float*)dd, p16, inv_keep, dropout_seed, v2s_seed_salt());
    V2S_LAUNCH_CHECK();
    return V2S_OK;
  }
  if (cols <= 1024)
    hipLaunchKernelGGL((norm_bwd_kernel<LN, 2>), dim3(nb), dim3(BW * 64), 0, s, (const bf16_t*)x, w, mean, rstd, (const bf16_t*)dy, (bf16_t*)dx,
                       (const bf16_t*)dx_add, dw, db, rows, cols, dd, p16, inv_keep, dropout_seed, v2s_seed_salt());
  else
    hipLaunchKernelGGL((norm_bwd_kernel<LN, 4>), dim3(nb), dim3(BW * 64), 0, s, (const bf16_t*)x, w, mean, rstd, (const bf16_t*)dy, (bf16_t*)dx,
                       (const bf16_t*)dx_add, dw, db, rows, cols, dd, p16, inv_keep, dropout_seed, v2s_seed_salt());
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_rmsnorm_bwd(const void* x, const float* w, const float* rstd, const void* dy, void* dx,
                               const void* dx_add, float* dw, int32_t rows, int32_t cols, void* stream) {
  return norm_bwd_launch<false>("v2s_rmsnorm_bwd", x, w, nullptr, rstd, dy, dx, dx_add, dw, nullptr, rows, cols, nullptr, 0.f, 0u, stream);
}

extern "C" int v2s_layernorm_bwd(const void* x, const float* w, const float* mean, const float* rstd, const void* dy,
                                 void* dx, const void* dx_add, float* dw, float* db, int32_t rows, int32_t cols, void* stream) {
  return norm_bwd_launch<true>("v2s_layernorm_bwd", x, w, mean, rstd, dy, dx, dx_add, dw, db, rows, cols, nullptr, 0.f, 0u, stream);
}

extern "C" int v2s_rmsnorm_bwd_drop(const void* x, const float* w, const float* rstd, const void* dy, void* dx, const void* dx_add,
                                    float* dw, int32_t rows, int32_t cols, void* dx_drop, float dropout_p, uint32_t dropout_seed,
                                    void* stream) {
  return norm_bwd_launch<false>("v2s_rmsnorm_bwd_drop", x, w, nullptr, rstd, dy, dx, dx_add, dw, nullptr, rows, cols, dx_drop, dropout_p,
                                dropout_seed, stream);
}

extern "C" int v2s_layernorm_bwd_drop(const void* x, const float* w, const float* mean, const float* rstd, const void* dy, void* dx,
                                      const void* dx_add, float* dw, float* db, int32_t rows, int32_t cols, void* dx_drop,
                                      float dropout_p, uint32_t dropout_seed, void* stream) {
  return norm_bwd_launch<true>("v2s_layernorm_bwd_drop", x, w, mean, rstd, dy, dx, dx_add, dw, db, rows, cols, dx_drop, dropout_p,
                               dropout_seed, stream);
}
