// Greedy-decoding kernels (gfx950, HBM-bound): single-query attention against a KV cache, argmax with
// the HF-4.28 greedy_search finished-row rule, KV-cache append.
// Replaces the per-step cache path of model/modeling_t5.py:484-525,555-556 (torch.cat growth, full bias
// recompute) and transformers==4.28.0 GenerationMixin.greedy_search (call site model/vid2seq.py:150-162).
#include <math.h>
#include "v2s_common.h"

namespace {

struct DecP {
  int B, H, Nk;
  const bf16_t* q; long q_bs;
  const bf16_t *k, *v; long kv_bs, kv_rs;
  bf16_t* o; long o_bs;
  const float* bias_row; long bias_ld;
  const uint8_t* key_mask; long mask_ld;
  float scale;
  const int* pos_dev; int bias_maxlen; int kv_group;
};

// one block (4 waves) per (b, h).  lane = (key slot ks = lane>>3, d-chunk c = lane&7): 8 keys per
// wave-instruction, each key's 64-wide dot product = 8 lanes x 8 elements (16-byte loads).
__global__ __launch_bounds__(256) void decode_attn_kernel(const DecP p) {
  __shared__ float s_m[4][8], s_l[4][8], s_o[4][8][64];
  const int bh = blockIdx.x, h = bh % p.H, b = bh / p.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ks = lane >> 3, c = lane & 7;
  int Nk = p.Nk;
  const float* bias_row = p.bias_row;
  if (p.pos_dev) {                      // device-resident step counter (graph replay)
    const int pos = *p.pos_dev;
    Nk = pos + 1;
    if (bias_row) bias_row += p.bias_maxlen - 1 - pos;
  }
  const int bkv = p.kv_group > 1 ? b / p.kv_group : b;
  float qv[8];
  unpack8(*reinterpret_cast<const uint4*>(p.q + (long)b * p.q_bs + h * 64 + c * 8), qv);
  const bf16_t* kp = p.k + (long)bkv * p.kv_bs + h * 64 + c * 8;
  const bf16_t* vp = p.v + (long)bkv * p.kv_bs + h * 64 + c * 8;
  float m = -INFINITY, l = 0.f, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // each wave walks the keys in chunks of 32 (4 keys per 8-lane group): the 8 loads of a chunk are issued together so that
  // ~8 KiB per wave are in flight (this kernel is a pure HBM stream: K and V are read exactly once per step)
  for (int k0 = wave * 32; k0 < Nk; k0 += 128) {
    uint4 kr[4], vr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + ks + 8 * j;
      kr[j] = make_uint4(0, 0, 0, 0); vr[j] = make_uint4(0, 0, 0, 0);
      if (k < Nk) {
        kr[j] = *reinterpret_cast<const uint4*>(kp + (long)k * p.kv_rs);
        vr[j] = *reinterpret_cast<const uint4*>(vp + (long)k * p.kv_rs);
      }
    }
    float sc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float kv[8];
      unpack8(kr[j], kv);
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) d += qv[e] * kv[e];
      d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
      const int k = k0 + ks + 8 * j;
      float s = -INFINITY;
      if (k < Nk) {
        s = d * p.scale;
        if (bias_row) s += bias_row[(long)h * p.bias_ld + k];
        if (p.key_mask && p.key_mask[(long)bkv * p.mask_ld + k] == 0) s = -3.0e38f;
      }
      sc[j] = s;
    }
    const float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
    if (mx > -INFINITY) {
      const float mn = fmaxf(m, mx);
      const float alpha = __expf(m - mn);
      l *= alpha;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] *= alpha;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pr = __expf(sc[j] - mn);        // exp(-inf) = 0 for out-of-range slots
        float vv[8];
        unpack8(vr[j], vv);
        l += pr;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += pr * vv[e];
      }
      m = mn;
    }
  }
  // merge the 8 key slots of this wave (lanes differing in bits 3..5), then the 4 waves through LDS
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
    const float mn = fmaxf(m, m2);
    const float a1 = (m == -INFINITY) ? 0.f : __expf(m - mn), a2 = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
    l = l * a1 + l2 * a2;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = acc[j] * a1 + __shfl_xor(acc[j], o, 64) * a2;
    m = mn;
  }
  if (ks == 0) {
    s_m[wave][c] = m; s_l[wave][c] = l;
#pragma unroll
    for (int j = 0; j < 8; ++j) s_o[wave][c][j] = acc[j];
  }
  __syncthreads();
  if (tid < 64) {
    const int cc = tid >> 3, j = tid & 7;
    float mm = -INFINITY;
    for (int w = 0; w < 4; ++w) mm = fmaxf(mm, s_m[w][cc]);
    float ll = 0.f, oo = 0.f;
    for (int w = 0; w < 4; ++w) {
      const float a = (s_m[w][cc] == -INFINITY) ? 0.f : __expf(s_m[w][cc] - mm);
      ll += s_l[w][cc] * a;
      oo += s_o[w][cc][j] * a;
    }
    p.o[(long)b * p.o_bs + h * 64 + cc * 8 + j] = f2bf(oo / ll);
  }
}

__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, long ld, int V, long* __restrict__ next_tok,
                                                     int* __restrict__ unfinished, int eos_id, int pad_id, long* __restrict__ seq_out,
                                                     long seq_ld, const int* __restrict__ pos_dev) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* z = logits + (long)row * ld;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int i = tid; i < V; i += 256) {
    const float v = z[i];
    if (v > best || (v == best && i < idx)) { best = v; idx = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float v2 = __shfl_xor(best, o, 64);
    const int i2 = __shfl_xor(idx, o, 64);
    if (v2 > best || (v2 == best && i2 < idx)) { best = v2; idx = i2; }
  }
  if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = idx; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    const int un = unfinished[row];
    const long tok = un ? (long)idx : (long)pad_id;       // finished rows emit pad
    next_tok[row] = tok;
    unfinished[row] = un && (tok != eos_id);
    if (seq_out) seq_out[(long)row * seq_ld + *pos_dev + 1] = tok;
  }
}

__global__ void counter_add_kernel(int* ctr, int delta) { *ctr += delta; }

__global__ __launch_bounds__(256) void kv_append_kernel(const bf16_t* __restrict__ src, long src_bs, bf16_t* __restrict__ cache,
                                                        long cache_bs, long cache_rs, int B, int width8, int pos,
                                                        const int* __restrict__ pos_dev) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= B * width8) return;
  if (pos_dev) pos = *pos_dev;
  const int b = t / width8, c = t - b * width8;
  *reinterpret_cast<uint4*>(cache + (long)b * cache_bs + (long)pos * cache_rs + c * 8) =
      *reinterpret_cast<const uint4*>(src + (long)b * src_bs + c * 8);
}

}  // namespace

extern "C" int v2s_decode_attn(const v2s_decode_attn_args* a, void* stream) {
  V2S_CHECK(a && a->B > 0 && a->H > 0 && a->Nk > 0 && a->q && a->k && a->v && a->o, V2S_ERR_ARG, "v2s_decode_attn: bad args");
  V2S_CHECK(((a->q_bs | a->kv_bs | a->kv_rs | a->o_bs) % 8) == 0, V2S_ERR_ALIGN, "v2s_decode_attn: strides must be multiples of 8");
  DecP p;
  p.B = a->B; p.H = a->H; p.Nk = a->Nk; p.q = (const bf16_t*)a->q; p.q_bs = a->q_bs;
  p.k = (const bf16_t*)a->k; p.v = (const bf16_t*)a->v; p.kv_bs = a->kv_bs; p.kv_rs = a->kv_rs;
  p.o = (bf16_t*)a->o; p.o_bs = a->o_bs; p.bias_row = a->bias_row; p.bias_ld = a->bias_ld ? a->bias_ld : a->Nk; p.key_mask = a->key_mask; p.mask_ld = a->mask_ld;
  p.scale = a->scale; p.pos_dev = a->pos_dev; p.bias_maxlen = a->bias_maxlen; p.kv_group = a->kv_group;
  hipLaunchKernelGGL(decode_attn_kernel, dim3(p.B * p.H), dim3(256), 0, (hipStream_t)stream, p);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_argmax_step(const float* logits, int64_t ld, int32_t rows, int32_t V, int64_t* next_tok,
                               int32_t* unfinished, int32_t eos_id, int32_t pad_id, void* stream) {
  V2S_CHECK(logits && next_tok && unfinished && rows > 0 && V > 0, V2S_ERR_ARG, "v2s_argmax_step: bad args");
  hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, (long)ld, V, (long*)next_tok, unfinished,
                     eos_id, pad_id, (long*)nullptr, 0L, (const int*)nullptr);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_argmax_step_seq(const float* logits, int64_t ld, int32_t rows, int32_t V, int64_t* next_tok,
                                   int32_t* unfinished, int32_t eos_id, int32_t pad_id, int64_t* seq_out, int64_t seq_ld,
                                   const int32_t* pos_dev, void* stream) {
  V2S_CHECK(logits && next_tok && unfinished && seq_out && pos_dev && rows > 0 && V > 0, V2S_ERR_ARG, "v2s_argmax_step_seq: bad args");
  hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, (long)ld, V, (long*)next_tok, unfinished,
                     eos_id, pad_id, (long*)seq_out, (long)seq_ld, pos_dev);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_counter_add(int32_t* ctr, int32_t delta, void* stream) {
  V2S_CHECK(ctr != nullptr, V2S_ERR_ARG, "v2s_counter_add: null counter");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, ctr, delta);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_kv_append(const void* src, int64_t src_bs, void* cache, int64_t cache_bs, int64_t cache_rs, int32_t B,
                             int32_t width, int32_t pos, const int32_t* pos_dev, void* stream) {
  V2S_CHECK(src && cache && B > 0 && width > 0 && (width % 8) == 0 && pos >= 0, V2S_ERR_ARG, "v2s_kv_append: bad args");
  V2S_CHECK(((src_bs | cache_bs | cache_rs) % 8) == 0, V2S_ERR_ALIGN, "v2s_kv_append: strides must be multiples of 8");
  const int total = B * (width / 8);
  hipLaunchKernelGGL(kv_append_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (long)src_bs,
                     (bf16_t*)cache, (long)cache_bs, (long)cache_rs, B, width / 8, pos, pos_dev);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}
