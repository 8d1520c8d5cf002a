// HBM-bound helper kernels of the Vid2Seq path (gfx950): embedding gather / scatter-add, broadcast add,
// dropout, label-smoothed cross entropy.  All use 16-byte accesses per lane and grid-stride loops.
//   embedding : model/modeling_t5.py:972, model/vid2seq.py:71 (nn.Embedding on the tied `shared` table)
//   pos add   : model/vit.py:119-126
//   CE        : model/modeling_t5.py:1721 (F.cross_entropy(ignore_index=-100, label_smoothing=eps))
#include <math.h>
#include "v2s_common.h"

namespace {

inline int grid_for(long work_items, int block = 256, int cap = 256 * 8) {
  long g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

__global__ __launch_bounds__(256) void embed_fwd_kernel(const long* __restrict__ ids, const bf16_t* __restrict__ table,
                                                        bf16_t* __restrict__ out, long n, int d, int vocab, uint32_t p16,
                                                        float inv_keep, uint32_t seed0, const uint32_t* __restrict__ salt) {
  const uint32_t seed = v2s_salted(seed0, salt);
  const int cpr = d >> 3;
  const long total = n * cpr;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const long row = t / cpr;
    const int c = (int)(t - row * cpr);
    long id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    uint4 v = *reinterpret_cast<const uint4*>(table + id * d + c * 8);
    if (p16) {
      float f[8];
      unpack8(v, f);
      v2s_drop8(f, (unsigned long long)row * d + c * 8, seed, p16, inv_keep);
      v = pack8(f);
    }
    *reinterpret_cast<uint4*>(out + row * d + c * 8) = v;
  }
}

// one ELEMENT per lane: a wave's atomic instruction then covers 64 consecutive floats of one table row (4 full 64-byte lines).
// With an 8-element chunk per lane each of the 8 atomic instructions touched 64 lines two floats at a time and the scatter ran at a
// quarter of the atomic rate (tools/ubench/atomic_rate.hip: 1.3 TB/s of operand bytes).
__global__ __launch_bounds__(256) void embed_bwd_kernel(const long* __restrict__ ids, const bf16_t* __restrict__ dy,
                                                        float* __restrict__ dtable, long n, int d, int vocab, uint32_t p16,
                                                        float inv_keep, uint32_t seed0, const uint32_t* __restrict__ salt) {
  const uint32_t seed = v2s_salted(seed0, salt);
  const long total = n * d;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const long row = t / d;
    const int e = (int)(t - row * d);
    long id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    float f = bf2f(dy[t]);
    if (p16) {          // same mask as the forward: bit (e & 7) of the keep byte of the element's 8-chunk
      const uint32_t keep = v2s_keep8((unsigned long long)(t & ~7L), seed, p16);
      f = ((keep >> (e & 7)) & 1u) ? f * inv_keep : 0.f;
    }
    if (f != 0.f) atomicAdd(dtable + id * d + e, f);
  }
}

// mode 0: y = x + add[i mod add_n]; mode 1: y = dropout(x); mode 2: y = x + add (same length)
__global__ __launch_bounds__(256) void ew_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ add,
                                                 bf16_t* __restrict__ y, long n8, long add_n8, int mode, uint32_t p16,
                                                 float inv_keep, uint32_t seed0, const uint32_t* __restrict__ salt) {
  const uint32_t seed = (mode == 1) ? v2s_salted(seed0, salt) : 0u;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n8; t += (long)gridDim.x * 256) {
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(x + t * 8), f);
    if (mode == 0 || mode == 2) {
      float a[8];
      const long ai = (mode == 0) ? (t % add_n8) : t;
      unpack8(*reinterpret_cast<const uint4*>(add + ai * 8), a);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += a[j];
    } else {
      v2s_drop8(f, (unsigned long long)t * 8, seed, p16, inv_keep);
    }
    *reinterpret_cast<uint4*>(y + t * 8) = pack8(f);
  }
}

// y = sum over p < nparts of parts[p * stride ..]: fp32 sum of bf16 terms, ONE rounding (the twelve cross-attention memory gradients of the
// decoder layers, each written by a plain GEMM: engine._cross_attn_bwd)
__global__ __launch_bounds__(256) void sum_n_kernel(const bf16_t* __restrict__ parts, long stride, int nparts, bf16_t* __restrict__ y, long n8) {
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n8; t += (long)gridDim.x * 256) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < nparts; ++p) {
      float f[8];
      unpack8(*reinterpret_cast<const uint4*>(parts + (long)p * stride + t * 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += f[j];
    }
    *reinterpret_cast<uint4*>(y + t * 8) = pack8(acc);
  }
}

__global__ __launch_bounds__(256) void bcast_grad_kernel(const bf16_t* __restrict__ dy, float* __restrict__ out, long n8,
                                                         long add_n8) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= add_n8) return;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long i = t; i < n8; i += add_n8) {
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(dy + i * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] += f[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) out[t * 8 + j] += s[j];
}

// ------------------------------------------------------------------------------------ cross entropy
struct MS { float m, s; };
__device__ __forceinline__ MS ms_merge(MS a, MS b) {
  const float m = fmaxf(a.m, b.m);
  MS r;
  r.m = m;
  r.s = (a.m == -INFINITY ? 0.f : a.s * __expf(a.m - m)) + (b.m == -INFINITY ? 0.f : b.s * __expf(b.m - m));
  return r;
}

__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, long ld, const long* __restrict__ labels,
                                                     int V, float eps, float* __restrict__ row_out) {
  __shared__ float sm[4], ss[4], st[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long y = labels[row];
  if (y < 0) {  // ignore_index (-100): no contribution
    if (tid == 0) { row_out[row * 2] = 0.f; row_out[row * 2 + 1] = 0.f; }
    return;
  }
  const float* z = logits + (long)row * ld;
  MS acc; acc.m = -INFINITY; acc.s = 0.f;
  float tot = 0.f;
  const int n4 = V >> 2;
  for (int i = tid; i < n4; i += 256) {
    const float4 v = *reinterpret_cast<const float4*>(z + i * 4);
    const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    const float m = fmaxf(acc.m, mx);
    acc.s = (acc.m == -INFINITY ? 0.f : acc.s * __expf(acc.m - m)) + __expf(v.x - m) + __expf(v.y - m) + __expf(v.z - m) + __expf(v.w - m);
    acc.m = m;
    tot += (v.x + v.y) + (v.z + v.w);
  }
  for (int i = n4 * 4 + tid; i < V; i += 256) {
    const float v = z[i];
    const float m = fmaxf(acc.m, v);
    acc.s = (acc.m == -INFINITY ? 0.f : acc.s * __expf(acc.m - m)) + __expf(v - m);
    acc.m = m;
    tot += v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    MS b; b.m = __shfl_xor(acc.m, o, 64); b.s = __shfl_xor(acc.s, o, 64);
    acc = ms_merge(acc, b);
    tot += __shfl_xor(tot, o, 64);
  }
  if (lane == 0) { sm[wave] = acc.m; ss[wave] = acc.s; st[wave] = tot; }
  __syncthreads();
  if (tid == 0) {
    MS a; a.m = sm[0]; a.s = ss[0];
    float t = st[0];
    for (int w = 1; w < 4; ++w) { MS b; b.m = sm[w]; b.s = ss[w]; a = ms_merge(a, b); t += st[w]; }
    const float lse = a.m + logf(a.s);
    const float nll = lse - z[y];
    const float smooth = lse - t / V;      // mean_c(-logp_c)
    row_out[row * 2] = lse;
    row_out[row * 2 + 1] = (1.f - eps) * nll + eps * smooth;
  }
}

// deterministic single-block reduction: loss_sum = sum row loss, count = #labels >= 0
__global__ __launch_bounds__(1024) void ce_reduce_kernel(const float* __restrict__ row_out, const long* __restrict__ labels,
                                                         int rows, float* __restrict__ loss_sum, float* __restrict__ count) {
  __shared__ float a[1024], c[1024];
  float s = 0.f, n = 0.f;
  for (int r = threadIdx.x; r < rows; r += 1024) {
    if (labels[r] >= 0) { s += row_out[r * 2 + 1]; n += 1.f; }
  }
  a[threadIdx.x] = s; c[threadIdx.x] = n;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) { a[threadIdx.x] += a[threadIdx.x + o]; c[threadIdx.x] += c[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { *loss_sum += a[0]; *count += c[0]; }
}

__device__ __forceinline__ void ce_st8(bf16_t* p, const float* f) { *reinterpret_cast<uint4*>(p) = pack8(f); }
__device__ __forceinline__ void ce_st8(float* p, const float* f) {
  *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
}
__device__ __forceinline__ void ce_st1(bf16_t* p, float f) { *p = f2bf(f); }
__device__ __forceinline__ void ce_st1(float* p, float f) { *p = f; }
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, long ld, const long* __restrict__ labels,
                                                     const float* __restrict__ row_out, int V, float eps,
                                                     const float* __restrict__ gscale, T* __restrict__ dl, long ldd) {
  const int row = blockIdx.x, tid = threadIdx.x;
  const long y = labels[row];
  T* d = dl + (long)row * ldd;
  const int n8 = V >> 3;
  for (int i = V + tid; i < ldd; i += 256) d[i] = 0;     // row padding (ragged V): keep it zero for the dgrad GEMM
  if (y < 0) {
    const float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < n8; i += 256) ce_st8(d + i * 8, z8);
    for (int i = n8 * 8 + tid; i < V; i += 256) d[i] = 0;
    return;
  }
  const float g = gscale[0];
  const float lse = row_out[row * 2];
  const float* z = logits + (long)row * ld;
  const float sm = eps / V;
  for (int i = tid; i < n8; i += 256) {
    const float4 v0 = *reinterpret_cast<const float4*>(z + i * 8), v1 = *reinterpret_cast<const float4*>(z + i * 8 + 4);
    float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float pr = __expf(f[j] - lse) - sm;
      if (i * 8 + j == y) pr -= (1.f - eps);
      f[j] = pr * g;
    }
    ce_st8(d + i * 8, f);
  }
  for (int i = n8 * 8 + tid; i < V; i += 256) {
    float pr = __expf(z[i] - lse) - sm;
    if (i == y) pr -= (1.f - eps);
    ce_st1(d + i, pr * g);
  }
}

// Measurement aid (bench.py `roofline.effective_sclk_mhz`): one wave per XCD stamps the shader-clock cycle counter (s_memtime: ticks at the
// CURRENT shader clock, so it slows down when the chip clocks down to its power budget) beside the constant 100 MHz counter
// (s_memrealtime).  Two probes around a region give the average shader clock the region ran at: d(cycles) / d(ticks) x 100 MHz.
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* __restrict__ out) {
  if (threadIdx.x != 0) return;
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;     // HW_REG_XCC_ID[3:0]
  const unsigned long long cyc = __builtin_amdgcn_s_memtime();
  const unsigned long long tick = __builtin_amdgcn_s_memrealtime();
  out[xcc * 4 + 0] = cyc; out[xcc * 4 + 1] = tick; out[xcc * 4 + 2] = xcc; out[xcc * 4 + 3] = 1;
}

}  // namespace

extern "C" int v2s_embed_fwd(const int64_t* ids, const void* table, void* out, int64_t n, int32_t d, int32_t vocab,
                             float dropout_p, uint32_t dropout_seed, void* stream) {
  V2S_CHECK(n > 0 && d > 0 && (d % 8) == 0 && vocab > 0, V2S_ERR_SHAPE, "v2s_embed_fwd: bad shape n=%ld d=%d", (long)n, d);
  const uint32_t p16 = (uint32_t)(dropout_p * 65536.0f + 0.5f);
  const float inv = p16 ? 1.0f / (1.0f - p16 / 65536.0f) : 1.f;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid_for(n * (d / 8))), dim3(256), 0, (hipStream_t)stream, (const long*)ids,
                     (const bf16_t*)table, (bf16_t*)out, (long)n, d, vocab, p16, inv, dropout_seed, v2s_seed_salt());
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_embed_bwd(const int64_t* ids, const void* dy, float* dtable, int64_t n, int32_t d, int32_t vocab,
                             float dropout_p, uint32_t dropout_seed, void* stream) {
  V2S_CHECK(n > 0 && d > 0 && (d % 8) == 0 && vocab > 0, V2S_ERR_SHAPE, "v2s_embed_bwd: bad shape n=%ld d=%d", (long)n, d);
  const uint32_t p16 = (uint32_t)(dropout_p * 65536.0f + 0.5f);
  const float inv = p16 ? 1.0f / (1.0f - p16 / 65536.0f) : 1.f;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid_for(n * d)), dim3(256), 0, (hipStream_t)stream, (const long*)ids,
                     (const bf16_t*)dy, dtable, (long)n, d, vocab, p16, inv, dropout_seed, v2s_seed_salt());
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_add_bcast(const void* x, const void* add, void* y, int64_t n, int64_t add_n, void* stream) {
  V2S_CHECK(n > 0 && add_n > 0 && (n % 8) == 0 && (add_n % 8) == 0 && (n % add_n) == 0, V2S_ERR_SHAPE, "v2s_add_bcast: bad sizes %ld %ld", (long)n, (long)add_n);
  hipLaunchKernelGGL(ew_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)add,
                     (bf16_t*)y, (long)(n / 8), (long)(add_n / 8), 0, 0u, 1.f, 0u, (const uint32_t*)nullptr);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_add(const void* a, const void* b, void* y, int64_t n, void* stream) {
  V2S_CHECK(n > 0 && (n % 8) == 0, V2S_ERR_SHAPE, "v2s_add: n must be a positive multiple of 8");
  hipLaunchKernelGGL(ew_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b,
                     (bf16_t*)y, (long)(n / 8), (long)(n / 8), 2, 0u, 1.f, 0u, (const uint32_t*)nullptr);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_sum_n(const void* parts, int64_t stride, int32_t nparts, void* y, int64_t n, void* stream) {
  V2S_CHECK(parts && y && n > 0 && (n % 8) == 0 && (stride % 8) == 0 && stride >= n && nparts >= 1 && nparts <= 64, V2S_ERR_SHAPE,
            "v2s_sum_n: n (%ld) and stride (%ld) must be multiples of 8, stride >= n, 1..64 parts (%d)", (long)n, (long)stride, nparts);
  hipLaunchKernelGGL(sum_n_kernel, dim3(grid_for(n / 8, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)parts, (long)stride, nparts,
                     (bf16_t*)y, (long)(n / 8));
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_dropout(const void* x, void* y, int64_t n, float p, uint32_t seed, void* stream) {
  V2S_CHECK(n > 0 && (n % 8) == 0 && p >= 0.f && p < 1.f, V2S_ERR_SHAPE, "v2s_dropout: bad args");
  const uint32_t p16 = (uint32_t)(p * 65536.0f + 0.5f);
  const float inv = p16 ? 1.0f / (1.0f - p16 / 65536.0f) : 1.f;
  hipLaunchKernelGGL(ew_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)nullptr,
                     (bf16_t*)y, (long)(n / 8), 1L, 1, p16, inv, seed, v2s_seed_salt());
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_bcast_grad(const void* dy, float* out, int64_t n, int64_t add_n, void* stream) {
  V2S_CHECK(n > 0 && add_n > 0 && (n % 8) == 0 && (add_n % 8) == 0 && (n % add_n) == 0, V2S_ERR_SHAPE, "v2s_bcast_grad: bad sizes");
  hipLaunchKernelGGL(bcast_grad_kernel, dim3((unsigned)((add_n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, out, (long)(n / 8), (long)(add_n / 8));
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_ce_fwd(const float* logits, int64_t ld, const int64_t* labels, int32_t rows, int32_t V, float eps,
                          float* row_lse, float* loss_sum, float* count, void* stream) {
  V2S_CHECK(rows > 0 && V > 0 && (ld % 4) == 0, V2S_ERR_SHAPE, "v2s_ce_fwd: bad shape rows=%d V=%d ld=%ld", rows, V, (long)ld);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(rows), dim3(256), 0, s, logits, (long)ld, (const long*)labels, V, eps, row_lse);
  V2S_LAUNCH_CHECK();
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(1024), 0, s, row_lse, (const long*)labels, rows, loss_sum, count);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_ce_bwd(const float* logits, int64_t ld, const int64_t* labels, const float* row_lse, int32_t rows,
                          int32_t V, float eps, const float* gscale, void* dlogits, int64_t ldd, void* stream) {
  V2S_CHECK(rows > 0 && V > 0 && (ld % 4) == 0 && (ldd % 8) == 0 && ldd >= V, V2S_ERR_SHAPE,
            "v2s_ce_bwd: bad shape rows=%d V=%d ld=%ld ldd=%ld (ldd must be a multiple of 8 >= V)", rows, V, (long)ld, (long)ldd);
  if (v2s_opt_fp32_io())                  // debug mode: fp32 d(logits)
    hipLaunchKernelGGL(ce_bwd_kernel<float>, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, (long)ld, (const long*)labels, row_lse,
                       V, eps, gscale, (float*)dlogits, (long)ldd);
  else
    hipLaunchKernelGGL(ce_bwd_kernel<bf16_t>, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, (long)ld, (const long*)labels, row_lse,
                       V, eps, gscale, (bf16_t*)dlogits, (long)ldd);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_clock_probe(uint64_t* out, void* stream) {
  V2S_CHECK(out != nullptr, V2S_ERR_ARG, "v2s_clock_probe: out is NULL");
  hipLaunchKernelGGL(clock_probe_kernel, dim3(8), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}
