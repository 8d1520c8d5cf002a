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
  const bf16_t *new_k, *new_v; long new_bs;     // the step's fresh K/V rows (fused cache append) or NULL
  int* row_map; long row_map_ld;                // beam search: physical cache row of (query row, key), or NULL
};

// one block (4 waves) per (KV row, h).  lane = (key slot ks = lane>>3, d-chunk c = lane&7): 8 keys per wave-instruction, each key's
// 64-wide dot product = 8 lanes x 8 elements (16-byte loads).  G = queries sharing the block's K/V rows: 1, or the beams of a batch
// entry against the encoder memory (kv_group): the K/V chunk is fetched and unpacked once and scored against all G queries -- with
// one block per (beam, head) the 4 beams of an entry each streamed the same rows (cross-attention of a 4-beam step: 90 us vs 40 us
// for the same K/V bytes at one query per entry; profiles/r02_decode_step.txt).
typedef float f32x2 __attribute__((ext_vector_type(2)));

// (768 blocks of a B = 64 beam step = 3 per CU: at G = 4 the kernel must fit 3 waves per SIMD -- 168 VGPRs -- or the third block of
// every CU runs in a second round)
template <int G>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(G == 8 ? 1 : (G == 4 ? 3 : 4))))
void decode_attn_kernel(const DecP p) {
  __shared__ float s_m[G][4][8], s_l[G][4][8], s_o[G][4][8][8];
  const int bh = blockIdx.x, h = bh % p.H;
  const int b0 = bh / p.H * G;                               // first query row of this block
  const int bkv = p.kv_group > 1 ? b0 / p.kv_group : b0 / G; // its K/V (and mask) row; G > 1 is launched with kv_group = 1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ks = lane >> 3, c = lane & 7;
  int Nk = p.Nk;
  const float* bias_row = p.bias_row;
  int newpos = -1;                      // key index whose K/V come from new_k / new_v (and are appended to the cache by this block)
  if (p.pos_dev) {                      // device-resident step counter (graph replay)
    const int pos = *p.pos_dev;
    Nk = pos + 1;
    if (bias_row) bias_row += p.bias_maxlen - 1 - pos;
    if (G == 1 && p.new_k) newpos = pos;
  }
  f32x2 qv[G][4];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(p.q + (long)(b0 + g) * p.q_bs + h * 64 + c * 8), f);
#pragma unroll
    for (int e = 0; e < 4; ++e) qv[g][e] = f32x2{f[2 * e], f[2 * e + 1]};
  }
  const bf16_t* kp = p.k + (long)bkv * p.kv_bs + h * 64 + c * 8;
  const bf16_t* vp = p.v + (long)bkv * p.kv_bs + h * 64 + c * 8;
  if (newpos >= 0 && tid < 16) {        // fused KV-cache append: this (b, h) block owns the 64-wide K and V pieces of the new row
    const bf16_t* src = (tid < 8 ? p.new_k : p.new_v) + (long)b0 * p.new_bs + h * 64 + (tid & 7) * 8;
    bf16_t* dst = const_cast<bf16_t*>(tid < 8 ? p.k : p.v) + (long)b0 * p.kv_bs + (long)newpos * p.kv_rs + h * 64 + (tid & 7) * 8;
    *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
  }
  // beam search: the keys of query row b live in the cache rows its ancestors wrote them to (row_map), staged once per block
  __shared__ unsigned short s_row[G == 1 ? 4096 : 1];
  const bool mapped = G == 1 && p.row_map != nullptr && Nk <= 4096;
  if constexpr (G == 1) {
    if (mapped) {
      for (int k = tid; k < Nk; k += 256) s_row[k] = (unsigned short)p.row_map[(long)b0 * p.row_map_ld + k];
      if (h == 0 && tid == 0 && newpos >= 0) p.row_map[(long)b0 * p.row_map_ld + newpos] = b0;
      __syncthreads();
    }
  }
  // padded encoder positions: their K/V rows are never fetched.  A masked key scores -3e38, so next to any valid key its weight is
  // exp(-3e38 - m) == 0 exactly and the result does not depend on what was loaded for it; the mask row is staged in LDS once, keys
  // past the last valid one are not visited and masked keys in between are not loaded (a synthetic ASR batch is ~14% padding: that
  // share of the cross-attention K/V stream, the largest HBM stream of a decode step).  A row without a single valid key keeps the
  // reference's uniform average over the masked keys, so nothing is skipped for it.
  __shared__ uint8_t s_valid[4096];
  __shared__ int s_last;
  const bool lds_mask = p.key_mask != nullptr && Nk <= 4096;
  if (lds_mask) {
    if (tid == 0) s_last = -1;
    __syncthreads();
    int last = -1;
    for (int k = tid; k < Nk; k += 256) {
      const uint8_t mv = p.key_mask[(long)bkv * p.mask_ld + k];
      s_valid[k] = mv;
      if (mv) last = k;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o, 64));
    if (lane == 0 && last >= 0) atomicMax(&s_last, last);
    __syncthreads();
  }
  const bool skip = lds_mask && s_last >= 0;
  const int Nvis = skip ? s_last + 1 : Nk;
  float m[G], l[G];
  f32x2 acc[G][4];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    m[g] = -INFINITY; l[g] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[g][e] = f32x2{0.f, 0.f};
  }
  // each wave walks the keys in chunks of 32 (4 keys per 8-lane group): the 8 loads of a chunk are issued together so that
  // ~8 KiB per wave are in flight (this kernel is a pure HBM stream: K and V are read at most once per step)
  for (int k0 = wave * 32; k0 < Nvis; k0 += 128) {
    uint4 kr[4], vr[4];
    bool masked[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + ks + 8 * j;
      kr[j] = make_uint4(0, 0, 0, 0); vr[j] = make_uint4(0, 0, 0, 0);
      masked[j] = false;
      if (k < Nvis) {
        if (lds_mask) masked[j] = s_valid[k] == 0;
        else if (p.key_mask) masked[j] = p.key_mask[(long)bkv * p.mask_ld + k] == 0;
        if (k == newpos) {              // not yet (visibly) in the cache: straight from the projection output
          kr[j] = *reinterpret_cast<const uint4*>(p.new_k + (long)b0 * p.new_bs + h * 64 + c * 8);
          vr[j] = *reinterpret_cast<const uint4*>(p.new_v + (long)b0 * p.new_bs + h * 64 + c * 8);
        } else if (!(skip && masked[j])) {
          long off = (long)k * p.kv_rs;
          if constexpr (G == 1) { if (mapped) off += ((long)s_row[k] - bkv) * p.kv_bs; }
          kr[j] = *reinterpret_cast<const uint4*>(kp + off);
          vr[j] = *reinterpret_cast<const uint4*>(vp + off);
        }
      }
    }
    float sc[G][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float kf[8];
      unpack8(kr[j], kf);
      const int k = k0 + ks + 8 * j;
      const float bias = (bias_row && k < Nvis) ? bias_row[(long)h * p.bias_ld + k] : 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        f32x2 d2 = qv[g][0] * f32x2{kf[0], kf[1]};
#pragma unroll
        for (int e = 1; e < 4; ++e) d2 = __builtin_elementwise_fma(qv[g][e], f32x2{kf[2 * e], kf[2 * e + 1]}, d2);
        float d = d2.x + d2.y;
        d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
        float sv = -INFINITY;
        if (k < Nvis) {
          sv = d * p.scale + bias;
          if (masked[j]) sv = -3.0e38f;
        }
        sc[g][j] = sv;
      }
    }
    f32x2 vf[4][4];                       // one query: unpacked per key below (registers: the short self-attention blocks want 8 waves/SIMD)
    if constexpr (G > 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float f[8];
        unpack8(vr[j], f);
#pragma unroll
        for (int e = 0; e < 4; ++e) vf[j][e] = f32x2{f[2 * e], f[2 * e + 1]};
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float mx = fmaxf(fmaxf(sc[g][0], sc[g][1]), fmaxf(sc[g][2], sc[g][3]));
      if (mx > -INFINITY) {
        const float mn = fmaxf(m[g], mx);
        const float alpha = __expf(m[g] - mn);
        l[g] *= alpha;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[g][e] *= alpha;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float pr = __expf(sc[g][j] - mn);        // exp(-inf) = 0 for out-of-range slots
          l[g] += pr;
          if constexpr (G == 1) {
            float f[8];
            unpack8(vr[j], f);
#pragma unroll
            for (int e = 0; e < 4; ++e) vf[j][e] = f32x2{f[2 * e], f[2 * e + 1]};
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[g][e] = __builtin_elementwise_fma(f32x2{pr, pr}, vf[j][e], acc[g][e]);
        }
        m[g] = mn;
      }
    }
  }
  // merge the 8 key slots of this wave (lanes differing in bits 3..5), then the 4 waves through LDS
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float mg = m[g], lg = l[g];
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      const float m2 = __shfl_xor(mg, o, 64), l2 = __shfl_xor(lg, o, 64);
      const float mn = fmaxf(mg, m2);
      const float a1 = (mg == -INFINITY) ? 0.f : __expf(mg - mn), a2 = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
      lg = lg * a1 + l2 * a2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[g][e].x = acc[g][e].x * a1 + __shfl_xor(acc[g][e].x, o, 64) * a2;
        acc[g][e].y = acc[g][e].y * a1 + __shfl_xor(acc[g][e].y, o, 64) * a2;
      }
      mg = mn;
    }
    if (ks == 0) {
      s_m[g][wave][c] = mg; s_l[g][wave][c] = lg;
#pragma unroll
      for (int e = 0; e < 4; ++e) { s_o[g][wave][c][2 * e] = acc[g][e].x; s_o[g][wave][c][2 * e + 1] = acc[g][e].y; }
    }
  }
  __syncthreads();
  for (int t = tid; t < G * 64; t += 256) {
    const int g = t >> 6, cc = (t >> 3) & 7, j = t & 7;
    float mm = -INFINITY;
    for (int w = 0; w < 4; ++w) mm = fmaxf(mm, s_m[g][w][cc]);
    float ll = 0.f, oo = 0.f;
    for (int w = 0; w < 4; ++w) {
      const float a = (s_m[g][w][cc] == -INFINITY) ? 0.f : __expf(s_m[g][w][cc] - mm);
      ll += s_l[g][w][cc] * a;
      oo += s_o[g][w][cc][j] * a;
    }
    p.o[(long)(b0 + g) * p.o_bs + h * 64 + cc * 8 + j] = f2bf(oo / ll);
  }
}

// ---- grouped cross-attention on the matrix pipe (round 3): the G (<= 16) beams of a batch entry against the entry's encoder K/V.
// The packed-FMA form above is VALU-bound at G = 4 (48 us per layer for 54 MB of K/V: 0.09 of the HBM roofline over a beam step);
// here the scores of a 32-key chunk are two 16 x 16 x 32 MFMA pairs  S^T[key][query] = K[key][d] . Q^T[d][query]  (queries padded
// to 16 columns), and the output is accumulated TRANSPOSED,  O^T[d][query] += V^T[d][key] . P^T[key][query],  so that a lane keeps
// all its values for ONE query (column = lane & 15): the running max / sum / rescale of the online softmax are lane-local, only the
// chunk maximum crosses the four key groups of a query (two shuffles).
//   * K fragments come straight from global memory (A operand: row = key, 8 consecutive d per lane = one 16-byte load); the key a
//     fragment row stands for is chosen so that the accumulator a lane ends up with -- rows 4g..4g+3 of the chunk's two tiles --
//     are the 8 CONSECUTIVE keys 8g..8g+7: exactly the B-operand layout P^T needs, no data movement between the two products;
//   * V goes through a wave-private 4 KB LDS tile ([key][d], 32-byte groups XOR-swizzled) and comes back as the A operand V^T
//     through ds_read_tr16_b64 (the transposing LDS read of gfx950), like the [k][row] operands of the GEMM kernels;
//   * P is split into bf16 hi + lo parts (two MFMAs per d tile): the products carry ~16 mantissa bits like the fp32 FMAs above.
// One block (4 waves) per (entry, head), the waves take the 32-key chunks round robin and are merged through LDS at the end.
// Mask semantics as above (masked key = -3e38; keys past the last valid one are not visited when the row has a valid key).
__device__ __forceinline__ int dtr_g(int k) { return (k & 3); }      // 32-byte-group swizzle of the V tile (4 groups per 128-byte row)

template <int NW>
__global__ __launch_bounds__(NW * 64) void decode_attn_mfma_kernel(const DecP p, int G) {
  // per wave: V tile [32 keys][64 d] bf16 (one buffer: LDS serves a wave's operations in order); merge buffers; NW waves: the
  // kernel is a latency-bound HBM stream (192 blocks at 16 entries x 12 heads), so the bytes in flight are what counts
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  char (*s_v)[4096] = reinterpret_cast<char (*)[4096]>(dsm);
  float (*s_o)[16][64] = reinterpret_cast<float (*)[16][64]>(dsm + NW * 4096);
  float (*s_m)[16] = reinterpret_cast<float (*)[16]>(dsm + NW * 4096 + NW * 4096);
  float (*s_l)[16] = s_m + NW;
  __shared__ __attribute__((aligned(8))) uint8_t s_valid[4096 + 64];
  __shared__ int s_last;
  const int bh = blockIdx.x, h = bh % p.H, ent = bh / p.H;
  const int b0 = ent * G;                                              // first query row of the entry; its K/V row = ent
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane & 15, g = lane >> 4;
  const int Nk = p.Nk;
  const bool has_mask = p.key_mask != nullptr;
  if (tid == 0) s_last = -1;
  __syncthreads();
  if (has_mask) {
    int last = -1;
    for (int k = tid; k < Nk; k += NW * 64) {
      const uint8_t mv = p.key_mask[(long)ent * p.mask_ld + k];
      s_valid[k] = mv;
      if (mv) last = k;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o, 64));
    if (lane == 0 && last >= 0) atomicMax(&s_last, last);
  } else {
    for (int k = tid; k < Nk; k += NW * 64) s_valid[k] = 1;
  }
  for (int k = Nk + tid; k < ((Nk + 63) & ~63); k += NW * 64) s_valid[k] = 1;      // chunk tail (those keys score -inf anyway)
  __syncthreads();
  const bool skipm = has_mask && s_last >= 0;                        // masked keys weigh exactly 0: their K/V rows are not fetched
  const int Nvis = skipm ? s_last + 1 : Nk;
  // Q^T as the B operand: column = query (zero beyond G), 8 consecutive d per lane
  bf16x8 qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (q < G) v = *reinterpret_cast<const uint4*>(p.q + (long)(b0 + q) * p.q_bs + h * 64 + ks * 32 + g * 8);
    qf[ks] = __builtin_bit_cast(bf16x8, v);
  }
  const bf16_t* kbase = p.k + (long)ent * p.kv_bs + h * 64;
  const bf16_t* vbase = p.v + (long)ent * p.kv_bs + h * 64;
  const int nchunk = (Nvis + 31) >> 5;
  uint4 kr[2][2], vr[4];
  auto load_chunk = [&](int ch) {
    const int kc = ch * 32;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int key = kc + (q >> 2) * 8 + 4 * t + (q & 3);            // fragment row q of tile t stands for this key
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        kr[t][ks] = (key < Nvis && !(skipm && s_valid[key] == 0)) ? *reinterpret_cast<const uint4*>(kbase + (long)key * p.kv_rs + ks * 32 + g * 8)
                                                                   : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = lane + 64 * i, key = kc + (idx >> 3), c16 = idx & 7;
      vr[i] = (key < Nvis && !(skipm && s_valid[key] == 0)) ? *reinterpret_cast<const uint4*>(vbase + (long)key * p.kv_rs + c16 * 8)
                                                             : make_uint4(0, 0, 0, 0);
    }
  };
  float m = -INFINITY, l = 0.f;
  f32x4 ot[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) ot[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (wave < nchunk) load_chunk(wave);
  for (int ch = wave; ch < nchunk; ch += NW) {
    const int kc = ch * 32;
    char* vt = s_v[wave];
    // stage V (this wave's tile)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = lane + 64 * i, kk = idx >> 3, c16 = idx & 7;
      *reinterpret_cast<uint4*>(vt + kk * 128 + (((c16 >> 1) ^ dtr_g(kk)) << 5) + ((c16 & 1) << 4)) = vr[i];
    }
    const bf16x8 k00 = __builtin_bit_cast(bf16x8, kr[0][0]), k01 = __builtin_bit_cast(bf16x8, kr[0][1]);
    const bf16x8 k10 = __builtin_bit_cast(bf16x8, kr[1][0]), k11 = __builtin_bit_cast(bf16x8, kr[1][1]);
    if (ch + NW < nchunk) load_chunk(ch + NW);                           // next chunk in flight under this one's arithmetic
    f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k00, qf[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k01, qf[1], s0, 0, 0, 0);
    f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k10, qf[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k11, qf[1], s1, 0, 0, 0);
    // lane (q, g): scores of keys kc + 8g + e, e = 0..7 (s0 = e 0..3, s1 = e 4..7)
    const uint2 vm = *reinterpret_cast<const uint2*>(s_valid + kc + 8 * g);
    float sv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int key = kc + 8 * g + e;
      const bool ok = (((e < 4 ? vm.x : vm.y) >> (8 * (e & 3))) & 0xffu) != 0;
      const float d = (e < 4 ? s0[e & 3] : s1[e & 3]) * p.scale;
      sv[e] = key < Nvis ? (ok ? d : -3.0e38f) : -INFINITY;
    }
    float mx = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);                                      // > -inf: a visited chunk has at least one key < Nvis
    const float alpha = __expf(m - mn);                                 // exp(-inf) = 0 on the first chunk
    float pr[8], ps = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { pr[e] = __expf(sv[e] - mn); ps += pr[e]; }
    l = l * alpha + ps;
    m = mn;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { ot[dt][0] *= alpha; ot[dt][1] *= alpha; ot[dt][2] *= alpha; ot[dt][3] *= alpha; }
    // P^T as the B operand, hi + lo bf16 parts
    uint4 ph, pl;
    {
      float lo[8];
      uint32_t hb[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { hb[e] = f2bf(pr[e]); lo[e] = pr[e] - __uint_as_float(hb[e] << 16); }
      ph = make_uint4(hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), hb[4] | (hb[5] << 16), hb[6] | (hb[7] << 16));
      pl = make_uint4(pack2bf(lo[0], lo[1]), pack2bf(lo[2], lo[3]), pack2bf(lo[4], lo[5]), pack2bf(lo[6], lo[7]));
    }
    const bf16x8 pfh = __builtin_bit_cast(bf16x8, ph), pfl = __builtin_bit_cast(bf16x8, pl);
    // V^T fragments (row = d, 8 consecutive keys per lane) through the transposing LDS read
    __builtin_amdgcn_s_waitcnt(0xc07f);                                  // lgkmcnt(0): this wave's V tile is in LDS (wave-private: no barrier)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int kk = g * 8 + (q >> 2), dm = dt * 16 + (q & 3) * 4, kk1 = kk + 4;
      const s16x4 vlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((V2S_LDS s16x4*)(vt + kk * 128 + (((dm >> 4) ^ dtr_g(kk)) << 5) + ((dm & 15) << 1)));
      const s16x4 vhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((V2S_LDS s16x4*)(vt + kk1 * 128 + (((dm >> 4) ^ dtr_g(kk1)) << 5) + ((dm & 15) << 1)));
      const s16x8 vv = {vlo[0], vlo[1], vlo[2], vlo[3], vhi[0], vhi[1], vhi[2], vhi[3]};
      const bf16x8 vf = __builtin_bit_cast(bf16x8, vv);
      ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pfh, ot[dt], 0, 0, 0);
      ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pfl, ot[dt], 0, 0, 0);
    }
  }
  // merge: the four key groups of a query hold partial sums (same m), then the four waves through LDS
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (g == 0) { s_m[wave][q] = m; s_l[wave][q] = l; }
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) s_o[wave][q][dt * 16 + 4 * g + r] = ot[dt][r];
  __syncthreads();
  for (int t = tid; t < G * 64; t += NW * 64) {
    const int qq = t >> 6, d = t & 63;
    float mm = -INFINITY;
    for (int w = 0; w < NW; ++w) mm = fmaxf(mm, s_m[w][qq]);
    float ll = 0.f, oo = 0.f;
    for (int w = 0; w < NW; ++w) {
      const float a = (s_m[w][qq] == -INFINITY) ? 0.f : __expf(s_m[w][qq] - mm);
      ll += s_l[w][qq] * a;
      oo += s_o[w][qq][d] * a;
    }
    p.o[(long)(b0 + qq) * p.o_bs + h * 64 + d] = f2bf(oo / ll);
  }
}

// one 1024-thread block per row: 16-byte loads, the whole row (V ~ 32k fp32 = 128 KiB) in flight in two rounds -- the earlier
// 256-thread scalar loop took 39.5 us per step for 64 rows (latency-bound; profiles/r02_decode_step.txt)
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ logits, long ld, int V, long* __restrict__ next_tok,
                                                      int* __restrict__ unfinished, int eos_id, int pad_id, long* __restrict__ seq_out,
                                                      long seq_ld, const int* __restrict__ pos_dev) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* z = logits + (long)row * ld;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  // torch.argmax's order: a NaN beats every number, the first one wins (a row of NaNs must not leave idx at its sentinel: it indexes the embedding table)
  auto take = [&](float v, int i) {
    const bool vn = v != v, bn = best != best;
    if (vn ? (!bn || i < idx) : (!bn && (v > best || (v == best && i < idx)))) { best = v; idx = i; }
  };
  const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(logits) & 15) == 0);
  const int V4 = vec ? (V & ~3) : 0;
  for (int i = tid * 4; i < V4; i += 4096) {
    const float4 q = *reinterpret_cast<const float4*>(z + i);
    take(q.x, i); take(q.y, i + 1); take(q.z, i + 2); take(q.w, i + 3);
  }
  for (int i = V4 + tid; i < V; i += 1024) take(z[i], i);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float v2 = __shfl_xor(best, o, 64);
    const int i2 = __shfl_xor(idx, o, 64);
    take(v2, i2);
  }
  if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = idx; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w) take(bv[w], bi[w]);
    const int un = unfinished[row];
    const long tok = un ? (long)idx : (long)pad_id;       // finished rows emit pad
    next_tok[row] = tok;
    unfinished[row] = un && (tok != eos_id);
    if (seq_out) seq_out[(long)row * seq_ld + *pos_dev + 1] = tok;
  }
}

// the tail of a greedy decode step as ONE launch: argmax_kernel's work, then the block copies the chosen token's embedding row into the
// residual-stream buffer the NEXT step starts from (what v2s_embed_fwd would do as the next step's first launch), and the last block to
// finish advances the device-resident step counter (what v2s_counter_add did as this step's last launch).  Every block reads the counter
// before it takes its ticket, so the increment cannot overtake a reader.
__global__ __launch_bounds__(1024) void argmax_tail_kernel(const float* __restrict__ logits, long ld, int V, long* __restrict__ next_tok,
                                                           int* __restrict__ unfinished, int eos_id, int pad_id, long* __restrict__ seq_out,
                                                           long seq_ld, int* __restrict__ pos_dev, const bf16_t* __restrict__ table,
                                                           bf16_t* __restrict__ h_out, int d, int vocab, int* __restrict__ ticket) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  __shared__ long s_tok;
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* z = logits + (long)row * ld;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  // torch.argmax's order: a NaN beats every number, the first one wins (a row of NaNs must not leave idx at its sentinel: it indexes the embedding table)
  auto take = [&](float v, int i) {
    const bool vn = v != v, bn = best != best;
    if (vn ? (!bn || i < idx) : (!bn && (v > best || (v == best && i < idx)))) { best = v; idx = i; }
  };
  const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(logits) & 15) == 0);
  const int V4 = vec ? (V & ~3) : 0;
  for (int i = tid * 4; i < V4; i += 4096) {
    const float4 q = *reinterpret_cast<const float4*>(z + i);
    take(q.x, i); take(q.y, i + 1); take(q.z, i + 2); take(q.w, i + 3);
  }
  for (int i = V4 + tid; i < V; i += 1024) take(z[i], i);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float v2 = __shfl_xor(best, o, 64);
    const int i2 = __shfl_xor(idx, o, 64);
    take(v2, i2);
  }
  if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = idx; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w) take(bv[w], bi[w]);
    const int un = unfinished[row];
    const long tok = un ? (long)idx : (long)pad_id;       // finished rows emit pad
    next_tok[row] = tok;
    unfinished[row] = un && (tok != eos_id);
    seq_out[(long)row * seq_ld + *pos_dev + 1] = tok;
    s_tok = tok;
  }
  __syncthreads();
  long id = s_tok;
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);       // (the clamp of embed_fwd_kernel)
  for (int c = tid; c < (d >> 3); c += 1024)
    *reinterpret_cast<uint4*>(h_out + (long)row * d + c * 8) = *reinterpret_cast<const uint4*>(table + id * d + c * 8);
  if (tid == 0) {
    __threadfence();
    if (atomicAdd(ticket, 1) == (int)gridDim.x - 1) {
      *ticket = 0;
      *pos_dev += 1;
    }
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


// ---- beam search helpers ----------------------------------------------------------------------------------------
// per row: log-softmax over V fp32 logits, add the row's running beam score, keep the K best (value, token) pairs, sorted
// descending (HF 4.28 beam_search: log_softmax + beam_scores[:, None] then topk over the beams of a batch entry; the top-2*nb
// of a batch entry are always among the per-beam top-2*nb, which the host merges).
// NT threads per row (1024, 512 for K = 16, 128 for K = 32: the candidate lists take NT * K * 8 bytes of LDS), 16-byte loads: the 256-thread scalar
// loop took 66 us per beam step for 256 rows (profiles/r02_decode_step.txt)
template <int K, int NT>
__global__ __launch_bounds__(NT) void topk_logprob_kernel(const float* __restrict__ logits, long ld, int V,
                                                          const float* __restrict__ beam_scores, float* __restrict__ out_val,
                                                          int* __restrict__ out_idx, int ban_tok, const int* __restrict__ pos_dev, int min_length, const float* __restrict__ row_lse) {
  constexpr int NW = NT / 64;
  __shared__ float sv[NT * K];
  __shared__ int si[NT * K];
  __shared__ float red_m[NW], red_s[NW];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* z = logits + (long)row * ld;
  // HF MinLengthLogitsProcessor: EOS is not a candidate while the decoder sequence (start token + decoded) is shorter than min_length
  const int ban = (ban_tok >= 0 && (pos_dev ? *pos_dev + 1 : 0) < min_length) ? ban_tok : -1;
  // Fast path (1024 threads, rows of <= 32768 aligned fp32): the per-thread sorted lists below cost ~12 us of a 64-row step -- with
  // 32 elements per thread some lane of every wave inserts at nearly every element, so the whole wave runs the 8-deep insertion
  // each time.  Here a thread only tracks its maximum; the K-th largest of the 1024 thread maxima is a lower bound T of the row's
  // K-th largest value (K distinct elements >= T exist), so the candidates are the elements >= T: K plus a handful.  They are
  // appended to a 64-slot list and ranked by one wave.  Rows that overflow the list (flat rows, masses of ties) take the general
  // path below.
  if constexpr (NT == 1024) {
    const bool al = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(logits) & 15) == 0);
    if (al && (V & 3) == 0 && V <= NT * 32) {
      __shared__ float c_v[64];
      __shared__ int c_i[64];
      __shared__ int c_n;
      __shared__ float c_thr;
      __shared__ float tw[NW * K];
      float4 q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = tid * 4 + u * NT * 4;
        q[u] = i < V ? *reinterpret_cast<const float4*>(z + i) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      }
      float fm = -INFINITY, fs = 0.f, tmax = -INFINITY;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = tid * 4 + u * NT * 4;
        const float mx = fmaxf(fmaxf(q[u].x, q[u].y), fmaxf(q[u].z, q[u].w));
        if (mx > -INFINITY) {
          const float mn = fmaxf(fm, mx);
          fs = fs * __expf(fm - mn) + ((__expf(q[u].x - mn) + __expf(q[u].y - mn)) + (__expf(q[u].z - mn) + __expf(q[u].w - mn)));
          fm = mn;
        }
        tmax = fmaxf(tmax, fmaxf(fmaxf(i == ban ? -INFINITY : q[u].x, i + 1 == ban ? -INFINITY : q[u].y),
                                 fmaxf(i + 2 == ban ? -INFINITY : q[u].z, i + 3 == ban ? -INFINITY : q[u].w)));
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(fm, o, 64), s2 = __shfl_xor(fs, o, 64);
        const float mn = fmaxf(fm, m2);
        fs = (fm == -INFINITY ? 0.f : fs * __expf(fm - mn)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mn));
        fm = mn;
      }
      if (lane == 0) { red_m[wave] = fm; red_s[wave] = fs; }
      // level 1: the K largest thread maxima of this wave (values only), descending
      {
        float v = tmax;
        for (int r = 0; r < K; ++r) {
          const float wm = wave_max(v);
          const unsigned long long hit = __ballot(v == wm);
          if (lane == (int)__ffsll((long long)hit) - 1) v = -INFINITY;
          if (lane == 0) tw[wave * K + r] = wm;
        }
      }
      if (tid == 0) c_n = 0;
      __syncthreads();
      float M = red_m[0];
      for (int w = 1; w < NW; ++w) M = fmaxf(M, red_m[w]);
      float S = 0.f;
      for (int w = 0; w < NW; ++w) S += red_m[w] == -INFINITY ? 0.f : red_s[w] * __expf(red_m[w] - M);
      const float lse = row_lse ? row_lse[row] : M + logf(S);
      const float base = beam_scores ? beam_scores[row] : 0.f;
      if (wave == 0) {                       // level 2: lane w walks the sorted list of wave w; the K-th extraction is the threshold
        int head = 0;
        float wm = -INFINITY;
        for (int r = 0; r < K; ++r) {
          const bool live = lane < NW && head < K;
          const float v = live ? tw[lane * K + head] : -INFINITY;
          wm = wave_max(v);
          const unsigned long long hit = __ballot(live && v == wm);
          if (hit != 0ull && lane == (int)__ffsll((long long)hit) - 1) ++head;
        }
        if (lane == 0) c_thr = wm;
      }
      __syncthreads();
      const float T = c_thr;
      if (T > -INFINITY) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = tid * 4 + u * NT * 4;
          const float e[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (e[c] >= T && i + c != ban) {
              const int slot = atomicAdd(&c_n, 1);
              if (slot < 64) { c_v[slot] = e[c]; c_i[slot] = i + c; }
            }
          }
        }
      }
      __syncthreads();
      const int n = c_n;
      if (T > -INFINITY && n >= K && n <= 64) {
        if (wave == 0) {
          const float v = lane < n ? c_v[lane] : -INFINITY;
          const int t = lane < n ? c_i[lane] : 0x7fffffff;
          int rank = 0;
          for (int j = 0; j < n; ++j) {
            const float vj = c_v[j]; const int tj = c_i[j];
            rank += (vj > v || (vj == v && tj < t)) ? 1 : 0;
          }
          if (lane < n && rank < K) { out_val[(long)row * K + rank] = v - lse + base; out_idx[(long)row * K + rank] = t; }
        }
        return;
      }
      __syncthreads();                       // overflow: the general path (re-reads the row)
    }
  }
  float tv[K]; int ti[K];
#pragma unroll
  for (int j = 0; j < K; ++j) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
  float m = -INFINITY, ssum = 0.f;
  auto consider = [&](float v, int i) {
    if (v > tv[K - 1] && i != ban) {           // insert (a banned token -- EOS below min_length -- counts in the softmax only) into the sorted (descending) list; equal values keep the lower index first
      tv[K - 1] = v; ti[K - 1] = i;
#pragma unroll
      for (int j = K - 1; j > 0; --j) {
        if (tv[j] > tv[j - 1]) { const float a = tv[j]; tv[j] = tv[j - 1]; tv[j - 1] = a; const int b = ti[j]; ti[j] = ti[j - 1]; ti[j - 1] = b; }
      }
    }
  };
  const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(logits) & 15) == 0);
  const int V4 = vec ? (V & ~3) : 0;
  for (int i = tid * 4; i < V4; i += NT * 4) {
    const float4 q = *reinterpret_cast<const float4*>(z + i);
    const float mn = fmaxf(fmaxf(m, fmaxf(q.x, q.y)), fmaxf(q.z, q.w));
    ssum = ssum * __expf(m - mn) + ((__expf(q.x - mn) + __expf(q.y - mn)) + (__expf(q.z - mn) + __expf(q.w - mn)));
    m = mn;
    consider(q.x, i); consider(q.y, i + 1); consider(q.z, i + 2); consider(q.w, i + 3);
  }
  for (int i = V4 + tid; i < V; i += NT) {
    const float v = z[i];
    const float mn = fmaxf(m, v);
    ssum = ssum * __expf(m - mn) + __expf(v - mn);
    m = mn;
    consider(v, i);
  }
  // block log-sum-exp
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(ssum, o, 64);
    const float mn = fmaxf(m, m2);
    ssum = (m == -INFINITY ? 0.f : ssum * __expf(m - mn)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mn));
    m = mn;
  }
  if (lane == 0) { red_m[wave] = m; red_s[wave] = ssum; }
#pragma unroll
  for (int j = 0; j < K; ++j) { sv[tid * K + j] = tv[j]; si[tid * K + j] = ti[j]; }
  __syncthreads();
  float M = red_m[0];
  for (int w = 1; w < NW; ++w) M = fmaxf(M, red_m[w]);
  float S = 0.f;
  for (int w = 0; w < NW; ++w) S += red_m[w] == -INFINITY ? 0.f : red_s[w] * __expf(red_m[w] - M);
  const float lse = row_lse ? row_lse[row] : M + logf(S);      // a processor may have rewritten logits against a stored lse
  const float base = beam_scores ? beam_scores[row] : 0.f;
  // two-level merge of the NT sorted lists (same order everywhere: value descending, equal values lower token first).  Level 1: every
  // wave extracts the K best of its 64 lists, wave-synchronous -- the earlier K rounds of BLOCK-wide argmax paid two barriers and a
  // serial scan over the waves per round (38 us per 64-row beam step, most of it here); level 2: wave 0 merges the NW wave lists.
  __shared__ float wv[NW * K];
  __shared__ int wt[NW * K];
  auto wave_best = [&](float& v, int& t, int& owner) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(v, o, 64); const int t2 = __shfl_xor(t, o, 64); const int o2 = __shfl_xor(owner, o, 64);
      if (v2 > v || (v2 == v && t2 < t)) { v = v2; t = t2; owner = o2; }
    }
  };
  int head = 0;
  for (int r = 0; r < K; ++r) {
    float v = head < K ? sv[tid * K + head] : -INFINITY;
    int t = head < K ? si[tid * K + head] : 0x7fffffff;
    int owner = lane;
    wave_best(v, t, owner);
    if (lane == owner) ++head;
    if (lane == 0) { wv[wave * K + r] = v; wt[wave * K + r] = t; }
  }
  __syncthreads();
  if (wave == 0) {
    int head2 = 0;
    for (int r = 0; r < K; ++r) {
      const bool live = lane < NW && head2 < K;
      float v = live ? wv[lane * K + head2] : -INFINITY;
      int t = live ? wt[lane * K + head2] : 0x7fffffff;
      int owner = lane;
      wave_best(v, t, owner);
      if (lane == owner) ++head2;
      if (lane == 0) { out_val[(long)row * K + r] = v - lse + base; out_idx[(long)row * K + r] = t; }
    }
  }
}

// dst[b, 0:len, :] = src[idx[b], 0:len, :]   (beam reorder of the growing self-attention cache, modeling_t5.py:1771-1793)
__global__ __launch_bounds__(256) void kv_gather_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                        const int* __restrict__ idx, long bs, long rs, int len, int width8) {
  const int b = blockIdx.y;
  const long sb = idx[b];
  const long total = (long)len * width8;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const long r = t / width8; const int c = (int)(t - r * width8);
    *reinterpret_cast<uint4*>(dst + (long)b * bs + r * rs + c * 8) = *reinterpret_cast<const uint4*>(src + sb * bs + r * rs + c * 8);
  }
}


// HF 4.28 RepetitionPenaltyLogitsProcessor on one row per block, in place.  hist[row][0..n) = decoder ids so far (start token
// included), n = *pos_dev + 1 (or n_static).  Gather first, scatter after a barrier: a token that occurs several times is penalised once.
// row_lse == NULL: the scores are raw logits (greedy_search): v < 0 ? v * pen : v / pen.
// row_lse != NULL: beam_search applies the processor to log-probabilities: the row's log-sum-exp is computed and stored, and the logit
// is rewritten so that logit' - lse equals the penalised log-probability (v2s_topk_logprob then takes the stored lse).
__global__ __launch_bounds__(256) void rep_penalty_kernel(float* __restrict__ scores, long ld, int V, const long* __restrict__ hist, long hist_ld,
                                                          const int* __restrict__ pos_dev, int n_static, float pen, float* __restrict__ row_lse) {
  __shared__ float red_m[4], red_s[4];
  __shared__ float gathered[1024];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* z = scores + (long)row * ld;
  const long* hs = hist + (long)row * hist_ld;
  int n = pos_dev ? *pos_dev + 1 : n_static;
  n = n < 1024 ? n : 1024;
  float lse = 0.f;
  if (row_lse) {
    float m = -INFINITY, ssum = 0.f;
    for (int i = tid; i < V; i += 256) {
      const float v = z[i];
      const float mn = fmaxf(m, v);
      ssum = ssum * __expf(m - mn) + __expf(v - mn);
      m = mn;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(ssum, o, 64);
      const float mn = fmaxf(m, m2);
      ssum = (m == -INFINITY ? 0.f : ssum * __expf(m - mn)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mn));
      m = mn;
    }
    if (lane == 0) { red_m[wave] = m; red_s[wave] = ssum; }
    __syncthreads();
    const float M = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
    float S = 0.f;
    for (int w = 0; w < 4; ++w) S += red_s[w] * __expf(red_m[w] - M);
    lse = M + logf(S);
    if (tid == 0) row_lse[row] = lse;
  }
  for (int j = tid; j < n; j += 256) {
    long t = hs[j];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    gathered[j] = z[t];
  }
  __syncthreads();
  for (int j = tid; j < n; j += 256) {
    long t = hs[j];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    float v = gathered[j] - lse;
    v = v < 0.f ? v * pen : v / pen;
    z[t] = v + lse;
  }
}


// ---- nucleus (top-p) sampling step (vid2seq.py:150-162 with do_sample=True: HF 4.28 sample() = TemperatureLogitsWarper +
// TopPLogitsWarper + multinomial).  One block per row; the row's softmax is built once in LDS.  Kept set = the most probable tokens
// whose preceding (larger) mass is < top_p, found by bisection on the probability threshold (no sort); the draw walks the kept
// tokens in index order with a counter-based uniform number (torch's RNG stream cannot be reproduced: distribution parity only).
__device__ __forceinline__ float block_sum256(float v, float* red, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// bit pattern of the k-th largest of V non-negative floats in LDS (non-negative floats order like their bit patterns): bisection on
// the 31 value bits, count(x >= T) per step -- exact, ties at the k-th value included (HF TopKLogitsWarper removes only scores strictly
// below the k-th).  red: >= 4 floats of LDS scratch.  All 256 threads call it.
__device__ __forceinline__ uint32_t kth_largest_bits(const float* pr, int V, int k, float* red, int tid) {
  uint32_t lo = 0u, hi = 0x7f800001u;     // count(>= lo) >= k (V >= k), count(>= hi) = 0 < k
  while (hi - lo > 1u) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    float c = 0.f;
    for (int i = tid; i < V; i += 256) c += __float_as_uint(pr[i]) >= mid ? 1.f : 0.f;
    c = block_sum256(c, red, tid);
    if (c >= (float)k) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void topp_sample_kernel(const float* __restrict__ logits, long ld, int V, float top_p, float inv_temp,
                                                          uint32_t seed, long* __restrict__ next_tok, int* __restrict__ unfinished, int eos_id,
                                                          int pad_id, long* __restrict__ seq_out, long seq_ld, const int* __restrict__ pos_dev,
                                                          float* __restrict__ probs_out, int min_length, int top_k) {
  extern __shared__ float pr[];          // V probabilities (unnormalised), then 256 + 8 floats of scratch
  float* part = pr + V;
  float* red = part + 256;
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* z = logits + (long)row * ld;
  // MinLengthLogitsProcessor: EOS gets -inf (probability 0) while the decoder sequence is shorter than min_length
  const int ban = (eos_id >= 0 && (pos_dev ? *pos_dev + 1 : 0) < min_length) ? eos_id : -1;
  float m = -INFINITY;
  for (int i = tid; i < V; i += 256) m = fmaxf(m, i == ban ? -INFINITY : z[i]);
  m = wave_max(m);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  for (int i = tid; i < V; i += 256) { const float e = i == ban ? 0.f : __expf((z[i] - m) * inv_temp); pr[i] = e; s += e; }
  if (top_k > 0 && top_k < V) {           // TopKLogitsWarper (HF generation default top_k = 50 applies whenever do_sample is set)
    const uint32_t kth = kth_largest_bits(pr, V, top_k, red, tid);
    s = 0.f;
    for (int i = tid; i < V; i += 256) { const float e = __float_as_uint(pr[i]) >= kth ? pr[i] : 0.f; pr[i] = e; s += e; }
  }
  const float Z = block_sum256(s, red, tid);
  // bisection: G(tau) = mass of tokens with p > tau; invariant G(lo) >= top_p > G(hi)
  float lo = -1.f, hi = 1.f;             // p_max = exp(0)/Z <= 1
  const float goal = top_p * Z;
  for (int it = 0; it < 40; ++it) {
    const float mid = 0.5f * (lo + hi), thr = mid * Z;
    float g = 0.f;
    for (int i = tid; i < V; i += 256) g += pr[i] > thr ? pr[i] : 0.f;
    g = block_sum256(g, red, tid);
    if (g < goal) hi = mid; else lo = mid;
  }
  const float thr = lo * Z;
  // kept mass per thread over a CONTIGUOUS chunk (the draw walks the tokens in index order)
  const int C = (V + 255) / 256, beg = min(V, tid * C), end = min(V, beg + C);
  float loc = 0.f;
  for (int i = beg; i < end; ++i) loc += pr[i] > thr ? pr[i] : 0.f;
  __syncthreads();
  part[tid] = loc;
  __syncthreads();
  if (tid == 0) {                         // exclusive scan of 256 partials (serial: trivial next to the passes above)
    float run = 0.f;
    for (int t = 0; t < 256; ++t) { const float v = part[t]; part[t] = run; run += v; }
    red[4] = run;                         // kept mass
    const int step = pos_dev ? *pos_dev : 0;
    const uint32_t h = v2s_hash32(seed ^ v2s_hash32((uint32_t)row * 0x9E3779B1u + (uint32_t)step * 0x85EBCA6Bu + 0x1234567u));
    red[5] = ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f) * run;       // target in (0, kept mass)
    red[6] = __int_as_float(-1);
  }
  __syncthreads();
  const float Zk = red[4], target = red[5];
  if (probs_out) {
    for (int i = tid; i < V; i += 256) probs_out[(long)row * V + i] = pr[i] > thr ? pr[i] / Zk : 0.f;
  }
  const float pre = part[tid];
  if (target >= pre && target < pre + loc) {
    float run = pre;
    int pick = -1;
    for (int i = beg; i < end; ++i) {
      if (pr[i] > thr) { pick = i; run += pr[i]; if (target < run) break; }
    }
    red[6] = __int_as_float(pick);
  }
  __syncthreads();
  if (tid == 0) {
    int pick = __float_as_int(red[6]);
    if (pick < 0) {                        // rounding left the target on a chunk boundary: take the most probable token
      float best = -1.f;
      for (int i = 0; i < V; ++i) if (pr[i] > best) { best = pr[i]; pick = i; }
    }
    const int un = unfinished[row];
    const long tok = un ? (long)pick : (long)pad_id;
    next_tok[row] = tok;
    unfinished[row] = un && (tok != eos_id);
    if (seq_out) seq_out[(long)row * seq_ld + *pos_dev + 1] = tok;
  }
}

// ---- beam-sample step (vid2seq.py:150-162 with do_sample=True AND num_beams > 1: HF 4.28 beam_sample()).  Per beam row:
//   scores = log_softmax(logits) -> processors (the caller's repetition penalty via row_lse; MinLength bans EOS) -> + beam score
//   -> warpers: / temperature, TopK(top_k, keep >= 2), TopP(top_p, keep >= 2)
// HF then draws 2*nb tokens WITHOUT replacement from the softmax over the nb * V warped scores of a batch entry, and sorts the
// draws by score.  Drawing without replacement from weights exp(score) = taking the largest keys score + Gumbel noise, so the
// kernel emits, per row, its K (>= 2*nb) kept candidates with the largest keys (sorted by key): [score | token | key]; the host
// merges the nb rows of an entry by key and orders the winners by score (beam.BeamScorer.advance).  The noise is a counter-based
// hash of (seed, step, row, token): torch's multinomial stream cannot be reproduced, the distribution is the same.
// One block per row; the row's tempered, unnormalised probabilities live in LDS; the kept set has at most 64 members (top_k <= 64).
__device__ __forceinline__ float beam_sample_gumbel(uint32_t seed, int step, int row, int tok) {
  const uint32_t h = v2s_hash32(seed ^ v2s_hash32((uint32_t)row * 0x9E3779B1u + (uint32_t)step * 0x85EBCA6Bu + 0x1234567u) ^
                                v2s_hash32((uint32_t)tok * 0xC2B2AE35u + 0x27D4EB2Fu));
  const float u = ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);
  return -logf(-logf(u));
}

__global__ __launch_bounds__(256) void beam_sample_cand_kernel(const float* __restrict__ logits, long ld, int V, const float* __restrict__ beam_scores,
                                                               float top_p, float inv_temp, int top_k, int min_keep, uint32_t seed,
                                                               float* __restrict__ out_val, int* __restrict__ out_tok, float* __restrict__ out_key,
                                                               int K, int ban_tok, const int* __restrict__ pos_dev, int min_length,
                                                               const float* __restrict__ row_lse) {
  extern __shared__ float pr[];          // V tempered probabilities (unnormalised), then 8 floats of scratch, then the kept list
  float* red = pr + V;
  int* s_tok = reinterpret_cast<int*>(red + 8);      // [64]
  int* s_cnt = s_tok + 64;
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* z = logits + (long)row * ld;
  const int step = pos_dev ? *pos_dev : 0;
  const int ban = (ban_tok >= 0 && step + 1 < min_length) ? ban_tok : -1;
  float m = -INFINITY;
  for (int i = tid; i < V; i += 256) m = fmaxf(m, z[i]);
  m = wave_max(m);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s1 = 0.f;                         // log_softmax normaliser: over ALL tokens, untempered (the processors come after it)
  for (int i = tid; i < V; i += 256) {
    const float d = z[i] - m;
    s1 += __expf(d);
    pr[i] = i == ban ? 0.f : __expf(d * inv_temp);
  }
  s1 = block_sum256(s1, red, tid);
  const float lse = row_lse ? row_lse[row] : m + logf(s1);
  if (tid == 0) *s_cnt = 0;
  const int k = top_k > 0 && top_k < 64 ? top_k : 64;
  const uint32_t kth = k < V ? kth_largest_bits(pr, V, k < min_keep ? min_keep : k, red, tid) : 0u;
  __syncthreads();
  for (int i = tid; i < V; i += 256) {
    const float e = pr[i];
    if (e > 0.f && __float_as_uint(e) >= kth) {
      const int slot = atomicAdd(s_cnt, 1);
      if (slot < 64) s_tok[slot] = i;
    }
  }
  __syncthreads();
  if (*s_cnt > 64) {
    // more than 64 candidates at or above the k-th value: a plateau of ties AT that value.  Keep everything strictly above it (fewer
    // than k <= 64 entries) and, of the ties, the LOWEST token ids -- the slots above were handed out in atomic arrival order, which
    // would make the kept set depend on scheduling
    __syncthreads();
    if (tid == 0) *s_cnt = 0;
    __syncthreads();
    for (int i = tid; i < V; i += 256) {
      const float e = pr[i];
      if (e > 0.f && __float_as_uint(e) > kth) s_tok[atomicAdd(s_cnt, 1)] = i;
    }
    __syncthreads();
    if (tid < 64) {
      int base = *s_cnt;
      for (int i0 = 0; i0 < V && base < 64; i0 += 64) {
        const int i = i0 + tid;
        const bool tie = i < V && pr[i] > 0.f && __float_as_uint(pr[i]) == kth;
        const unsigned long long bal = __ballot(tie);
        const int before = __popcll(bal & ((1ull << tid) - 1ull));
        if (tie && base + before < 64) s_tok[base + before] = i;
        base += __popcll(bal);
      }
      if (tid == 0) *s_cnt = min(base, 64);
    }
    __syncthreads();
  }
  if (tid >= 64) return;
  const int n = min(*s_cnt, 64), lane = tid;
  const int tok = lane < n ? s_tok[lane] : 0x7fffffff;
  const float e = lane < n ? pr[tok] : 0.f;
  // rank by probability (descending; equal values: lower token first) and the mass in front of each member
  int rank = 0;
  float before = 0.f, Zk = 0.f;
  for (int i = 0; i < n; ++i) {
    const float ei = __shfl(e, i, 64);
    const int ti = __shfl(tok, i, 64);
    const bool ahead = ei > e || (ei == e && ti < tok);
    rank += ahead ? 1 : 0;
    before += ahead ? ei : 0.f;
    Zk += ei;
  }
  const bool keep = lane < n && (before < top_p * Zk || rank < min_keep);
  const float base = beam_scores ? beam_scores[row] : 0.f;
  const float score = keep ? ((z[tok] - lse) + base) * inv_temp : -INFINITY;
  const float key = keep ? score + beam_sample_gumbel(seed, step, row, tok) : -INFINITY;
  int krank = 0;
  for (int i = 0; i < 64; ++i) {
    const float ki = __shfl(key, i, 64);
    krank += (ki > key || (ki == key && i < lane)) ? 1 : 0;
  }
  if (krank < K) {
    const long o = (long)row * K + krank;
    out_val[o] = score; out_tok[o] = keep ? tok : 0; out_key[o] = key;
  }
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
  p.new_k = (const bf16_t*)a->new_k; p.new_v = (const bf16_t*)a->new_v; p.new_bs = a->new_bs;
  p.row_map = a->row_map; p.row_map_ld = a->row_map_ld;
  V2S_CHECK(!a->row_map || (a->new_k && a->B <= 65535 && a->Nk <= 4096 && a->row_map_ld >= a->Nk), V2S_ERR_ARG,
            "v2s_decode_attn: row_map needs the fused append (new_k), at most 65535 rows and 4096 keys, row_map_ld >= Nk");
  V2S_CHECK(!a->new_k || (a->new_v && a->pos_dev && a->kv_group <= 1 && (a->new_bs % 8) == 0), V2S_ERR_ARG,
            "v2s_decode_attn: the fused cache append needs new_v, pos_dev and one KV row per batch entry");
  // the beams of a batch entry share a block (K/V fetched once) when they divide evenly; any other group size keeps one block per row
  // the beams of a batch entry against its encoder K/V on the matrix pipe: any group size 2..16 (no bias, no cache append, no row map)
  if (a->kv_group >= 2 && a->kv_group <= 16 && (a->B % a->kv_group) == 0 && !a->new_k && !a->bias_row && !a->pos_dev && !a->row_map &&
      a->Nk <= 4096 && (a->kv_rs % 8) == 0 && v2s_opt_gemm_skinny() != 3) {
    // waves per block: enough bytes in flight for the HBM stream at few blocks (16 entries x 12 heads = 192 blocks on 256 CUs).  Per layer,
    // twelve K|V buffers in rotation (tools/decode_xattn_ab.py): 16 entries x 4 beams 29.9 us packed-FMA, 20.5 / 16.1 / 18.1 at 4 / 8 / 16
    // waves; 64 x 4: 49.9 -> 43.4 / 45.2 / 53.0; 16 x 12 beams: 57.6 -> 16.4
    const unsigned nblk = (unsigned)(p.B / a->kv_group * p.H);
    const int nw_opt = v2s_opt_gemm_skinny();        // A/B hook (tools/decode_ab.py): gemm_skinny = 5 | 6 | 7 -> 4 | 8 | 16 waves
    const int nw = nw_opt == 5 ? 4 : (nw_opt == 6 ? 8 : (nw_opt == 7 ? 16 : (nblk >= 512 ? 4 : 8)));
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)decode_attn_mfma_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);   // + 4.1 KB static
      (void)hipFuncSetAttribute((const void*)decode_attn_mfma_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);   // + 4.1 KB static
      (void)hipFuncSetAttribute((const void*)decode_attn_mfma_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);   // + 4.1 KB static
      attr = true;
    }
    const size_t dyn = (size_t)nw * (4096 + 4096 + 128);
    if (nw == 4) hipLaunchKernelGGL(decode_attn_mfma_kernel<4>, dim3(nblk), dim3(256), dyn, (hipStream_t)stream, p, (int)a->kv_group);
    else if (nw == 8) hipLaunchKernelGGL(decode_attn_mfma_kernel<8>, dim3(nblk), dim3(512), dyn, (hipStream_t)stream, p, (int)a->kv_group);
    else hipLaunchKernelGGL(decode_attn_mfma_kernel<16>, dim3(nblk), dim3(1024), dyn, (hipStream_t)stream, p, (int)a->kv_group);
    V2S_LAUNCH_CHECK();
    return V2S_OK;
  }
  const int G = (a->kv_group == 2 || a->kv_group == 4 || a->kv_group == 8) && (a->B % a->kv_group) == 0 && !a->new_k ? a->kv_group : 1;
  const dim3 grid((unsigned)(p.B / G * p.H));
  hipStream_t s = (hipStream_t)stream;
  if (G == 8) { p.kv_group = 1; hipLaunchKernelGGL(decode_attn_kernel<8>, grid, dim3(256), 0, s, p); }
  else if (G == 4) { p.kv_group = 1; hipLaunchKernelGGL(decode_attn_kernel<4>, grid, dim3(256), 0, s, p); }
  else if (G == 2) { p.kv_group = 1; hipLaunchKernelGGL(decode_attn_kernel<2>, grid, dim3(256), 0, s, p); }
  else hipLaunchKernelGGL(decode_attn_kernel<1>, grid, dim3(256), 0, s, p);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_argmax_step(const float* logits, int64_t ld, int32_t rows, int32_t V, int64_t* next_tok,
                               int32_t* unfinished, int32_t eos_id, int32_t pad_id, void* stream) {
  V2S_CHECK(logits && next_tok && unfinished && rows > 0 && V > 0, V2S_ERR_ARG, "v2s_argmax_step: bad args");
  hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(1024), 0, (hipStream_t)stream, logits, (long)ld, V, (long*)next_tok, unfinished,
                     eos_id, pad_id, (long*)nullptr, 0L, (const int*)nullptr);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_argmax_step_seq(const float* logits, int64_t ld, int32_t rows, int32_t V, int64_t* next_tok,
                                   int32_t* unfinished, int32_t eos_id, int32_t pad_id, int64_t* seq_out, int64_t seq_ld,
                                   const int32_t* pos_dev, void* stream) {
  V2S_CHECK(logits && next_tok && unfinished && seq_out && pos_dev && rows > 0 && V > 0, V2S_ERR_ARG, "v2s_argmax_step_seq: bad args");
  hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(1024), 0, (hipStream_t)stream, logits, (long)ld, V, (long*)next_tok, unfinished,
                     eos_id, pad_id, (long*)seq_out, (long)seq_ld, pos_dev);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_argmax_step_tail(const float* logits, int64_t ld, int32_t rows, int32_t V, int64_t* next_tok, int32_t* unfinished,
                                    int32_t eos_id, int32_t pad_id, int64_t* seq_out, int64_t seq_ld, int32_t* pos_dev, const void* table,
                                    void* h_out, int32_t d, int32_t vocab, int32_t* ticket, void* stream) {
  V2S_CHECK(logits && next_tok && unfinished && seq_out && pos_dev && table && h_out && ticket && rows > 0 && V > 0 && vocab > 0, V2S_ERR_ARG,
            "v2s_argmax_step_tail: bad args");
  V2S_CHECK(d > 0 && (d % 8) == 0 && (((uintptr_t)table | (uintptr_t)h_out) & 15) == 0, V2S_ERR_ALIGN, "v2s_argmax_step_tail: d must be a multiple of 8, buffers 16-byte aligned");
  hipLaunchKernelGGL(argmax_tail_kernel, dim3(rows), dim3(1024), 0, (hipStream_t)stream, logits, (long)ld, V, (long*)next_tok, unfinished,
                     eos_id, pad_id, (long*)seq_out, (long)seq_ld, pos_dev, (const bf16_t*)table, (bf16_t*)h_out, d, vocab, ticket);
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

namespace {
// Any K (wide beams: num_beams > 16 needs more than 32 candidates per row): the row's scores in LDS, K rounds of a block-wide argmax
// (value descending, lower token first -- the order of the specialised kernels above).  ~1.5 us per round: for the rare wide-beam call.
__global__ __launch_bounds__(1024) void topk_rounds_kernel(const float* __restrict__ logits, long ld, int V, int K, const float* __restrict__ beam_scores,
                                                           float* __restrict__ out_val, int* __restrict__ out_idx, int ban_tok,
                                                           const int* __restrict__ pos_dev, int min_length, const float* __restrict__ row_lse) {
  extern __shared__ __attribute__((aligned(16))) float sv[];            // [V]
  __shared__ float red_v[16], red_s[16];
  __shared__ int red_i[16];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* z = logits + (long)row * ld;
  const int ban = (ban_tok >= 0 && (pos_dev ? *pos_dev + 1 : 0) < min_length) ? ban_tok : -1;
  float fm = -INFINITY;
  for (int i = tid; i < V; i += 1024) { const float v = z[i]; sv[i] = i == ban ? -INFINITY : v; fm = fmaxf(fm, v); }
  fm = wave_max(fm);
  if (lane == 0) red_v[wave] = fm;
  __syncthreads();
  float M = red_v[0];
  for (int w = 1; w < 16; ++w) M = fmaxf(M, red_v[w]);
  float fs = 0.f;
  for (int i = tid; i < V; i += 1024) fs += __expf(z[i] - M);            // normaliser over ALL tokens (the EOS ban comes after log_softmax)
  fs = wave_sum(fs);
  __syncthreads();
  if (lane == 0) red_s[wave] = fs;
  __syncthreads();
  float S = 0.f;
  for (int w = 0; w < 16; ++w) S += red_s[w];
  const float lse = row_lse ? row_lse[row] : M + logf(S);
  const float base = beam_scores ? beam_scores[row] : 0.f;
  for (int r = 0; r < K; ++r) {
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int i = tid; i < V; i += 1024) { const float v = sv[i]; if (v > bv) { bv = v; bi = i; } }       // ascending i: the first maximum is the lowest token
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(bv, o, 64); const int i2 = __shfl_xor(bi, o, 64);
      if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
    }
    __syncthreads();
    if (lane == 0) { red_v[wave] = bv; red_i[wave] = bi; }
    __syncthreads();
    float gv = red_v[0]; int gi = red_i[0];
    for (int w = 1; w < 16; ++w)
      if (red_v[w] > gv || (red_v[w] == gv && red_i[w] < gi)) { gv = red_v[w]; gi = red_i[w]; }
    if (tid == 0) {
      const bool ok = gv > -INFINITY;
      out_val[(long)row * K + r] = ok ? (gv - lse) + base : -INFINITY;
      out_idx[(long)row * K + r] = ok ? gi : 0;
      if (ok) sv[gi] = -INFINITY;
    }
    __syncthreads();
  }
}
}  // namespace

extern "C" int v2s_topk_logprob(const float* logits, int64_t ld, int32_t rows, int32_t V, int32_t K, const float* beam_scores,
                                float* out_val, int32_t* out_idx, int32_t ban_token, const int32_t* pos_dev, int32_t min_length, const float* row_lse,
                                void* stream) {
  V2S_CHECK(logits && out_val && out_idx && rows > 0 && V > 0, V2S_ERR_ARG, "v2s_topk_logprob: bad args");
  V2S_CHECK(K == 2 || K == 4 || K == 8 || K == 16 || K == 32 || (K > 32 && K <= 512), V2S_ERR_ARG, "v2s_topk_logprob: K must be 2, 4, 8, 16, 32 or 33..512 (got %d)", K);
  hipStream_t s = (hipStream_t)stream;
  if (K > 32) {                             // wide beams: generic K rounds over an LDS-resident row
    V2S_CHECK((size_t)V * 4 <= 150 * 1024, V2S_ERR_SHAPE, "v2s_topk_logprob: K > 32 keeps the row in LDS: vocabulary %d too large", V);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)topk_rounds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr = true; }
    hipLaunchKernelGGL(topk_rounds_kernel, dim3(rows), dim3(1024), (size_t)V * 4, s, logits, (long)ld, V, K, beam_scores, out_val, out_idx, ban_token, pos_dev,
                       min_length, row_lse);
    V2S_LAUNCH_CHECK();
    return V2S_OK;
  }
  if (K == 2) hipLaunchKernelGGL((topk_logprob_kernel<2, 1024>), dim3(rows), dim3(1024), 0, s, logits, (long)ld, V, beam_scores, out_val, out_idx, ban_token, pos_dev, min_length, row_lse);
  else if (K == 4) hipLaunchKernelGGL((topk_logprob_kernel<4, 1024>), dim3(rows), dim3(1024), 0, s, logits, (long)ld, V, beam_scores, out_val, out_idx, ban_token, pos_dev, min_length, row_lse);
  else if (K == 8) hipLaunchKernelGGL((topk_logprob_kernel<8, 1024>), dim3(rows), dim3(1024), 0, s, logits, (long)ld, V, beam_scores, out_val, out_idx, ban_token, pos_dev, min_length, row_lse);
  else if (K == 16) hipLaunchKernelGGL((topk_logprob_kernel<16, 512>), dim3(rows), dim3(512), 0, s, logits, (long)ld, V, beam_scores, out_val, out_idx, ban_token, pos_dev, min_length, row_lse);
  else hipLaunchKernelGGL((topk_logprob_kernel<32, 128>), dim3(rows), dim3(128), 0, s, logits, (long)ld, V, beam_scores, out_val, out_idx, ban_token, pos_dev, min_length, row_lse);   // 9..16 beams
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

// ---- beam bookkeeping on the device: what transformers==4.28.0 BeamSearchScorer.process / BeamHypotheses.add do between two decoder
// steps (call site model/vid2seq.py:150-162, num_beams > 1, do_sample = False, early_stopping = False), restated after
// vidchapters_amd/beam.py (the host scorer, which stays the reference and serves sampling / teacher forcing).  One block per batch
// entry: merges the entry's nb x K per-beam candidates into its 2 nb best (stable: value descending, flat index ascending), walks
// them like process() (EOS candidates of rank < nb become finished hypotheses, the first nb others the next beams), keeps the nb best
// hypotheses per entry (score = sum_logprobs / len^length_penalty in double, evicting the lowest (score, insertion order)), marks the
// entry done when the heap is full and cannot be beaten, then applies the beam permutation IN PLACE to the entry's rows of the token
// history and of the self-attention row map (column by column: a thread reads the nb values of its column, then writes them) and
// appends the new tokens.  With it a beam step needs no host round trip: the decode graph replays back to back.
struct BeamAdvP {
  const float* cand_val; const int* cand_tok; int K;          // [rows][K], sorted per row
  int B, nb, eos, pad;
  const double* len_pow;                                      // [max_length + 1]: n^length_penalty as the host computes it (bit-identical scores)
  const int* pos_dev;                                         // step index t: the sequences hold t + 1 tokens
  long* hist; long hist_ld; int max_length;                   // [rows][max_length] decoder ids so far
  int* row_map; long row_map_ld;                              // [rows][maxlen] or NULL
  long* next_tok; float* beam_scores; int* src_rows;          // [rows]: outputs for the next step
  int* hyp_tok; int* hyp_len; double* hyp_score; int* hyp_order;   // [B][nb][max_length] / [B][nb]
  int* heap_n; double* heap_worst; int* heap_added; int* done; int* ndone;   // [B], [1]
};

__global__ __launch_bounds__(256) void beam_advance_kernel(const BeamAdvP p) {
  __shared__ float s_val[512];
  __shared__ int s_tok[512];
  __shared__ float sel_val[32];
  __shared__ int sel_tok[32], sel_src[32];
  __shared__ int new_tok[16], new_src[16], push_src[16], push_slot[16], n_push;
  __shared__ float new_sc[16];
  const int ent = blockIdx.x, tid = threadIdx.x, nb = p.nb, K = p.K, n = nb * K;
  const int cur_len = *p.pos_dev + 1;
  const int row0 = ent * nb;
  if (p.done[ent]) {                         // a finished entry keeps its rows; its beams emit pad (BeamSearchScorer.process)
    if (tid < nb) { p.next_tok[row0 + tid] = p.pad; p.beam_scores[row0 + tid] = 0.f; p.src_rows[row0 + tid] = row0 + tid; }
    return;
  }
  for (int i = tid; i < n; i += 256) { s_val[i] = p.cand_val[(long)row0 * K + i]; s_tok[i] = p.cand_tok[(long)row0 * K + i]; }
  __syncthreads();
  for (int i = tid; i < n; i += 256) {       // rank of candidate i among the entry's n (stable descending)
    const float v = s_val[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += (s_val[j] > v || (s_val[j] == v && j < i)) ? 1 : 0;
    if (rank < 2 * nb) { sel_val[rank] = v; sel_tok[rank] = s_tok[i]; sel_src[rank] = i / K; }
  }
  __syncthreads();
  if (tid == 0) {
    int k = 0, np = 0;
    int hn = p.heap_n[ent], added = p.heap_added[ent];
    double worst = p.heap_worst[ent];
    const double denom = p.len_pow[cur_len];
    for (int rank = 0; rank < 2 * nb; ++rank) {
      const int t = sel_tok[rank];
      if (t == p.eos) {
        if (rank < nb) {                     // BeamHypotheses.add
          const double score = (double)sel_val[rank] / denom;
          if (hn < nb || score > worst) {
            int slot;
            if (hn < nb) {
              slot = hn++;
              worst = score < worst ? score : worst;
            } else {                         // evict the lowest (score, insertion order); the new one is strictly better than it
              slot = 0;
              for (int i = 1; i < nb; ++i) {
                const double a = p.hyp_score[ent * nb + i], b = p.hyp_score[ent * nb + slot];
                if (a < b || (a == b && p.hyp_order[ent * nb + i] < p.hyp_order[ent * nb + slot])) slot = i;
              }
            }
            p.hyp_score[ent * nb + slot] = score; p.hyp_order[ent * nb + slot] = added++; p.hyp_len[ent * nb + slot] = cur_len;
            if (hn >= nb) {
              worst = p.hyp_score[ent * nb];
              for (int i = 1; i < nb; ++i) worst = p.hyp_score[ent * nb + i] < worst ? p.hyp_score[ent * nb + i] : worst;
            }
            push_src[np] = sel_src[rank]; push_slot[np] = slot; ++np;
          }
        }
        continue;
      }
      new_tok[k] = t; new_src[k] = sel_src[rank]; new_sc[k] = sel_val[rank];
      if (++k == nb) break;
    }
    for (; k < nb; ++k) { new_tok[k] = p.pad; new_src[k] = k; new_sc[k] = 0.f; }     // (cannot happen with 2 nb candidates and one EOS id)
    p.heap_n[ent] = hn; p.heap_added[ent] = added; p.heap_worst[ent] = worst;
    n_push = np;
    if (hn >= nb && worst >= (double)sel_val[0] / denom) { p.done[ent] = 1; atomicAdd(p.ndone, 1); }
  }
  __syncthreads();
  for (int i = 0; i < n_push; ++i) {         // finished hypotheses: the sequence of the source beam BEFORE this step's permutation
    const long* src = p.hist + (long)(row0 + push_src[i]) * p.hist_ld;
    int* dst = p.hyp_tok + ((long)ent * nb + push_slot[i]) * p.max_length;
    for (int c = tid; c < cur_len; c += 256) dst[c] = (int)src[c];
    __syncthreads();                         // (a later push of this step may reuse the slot)
  }
  // the permutation, column by column in place; then the new tokens
  for (int c = tid; c < cur_len; c += 256) {
    long v[16];
    for (int j = 0; j < nb; ++j) v[j] = p.hist[(long)(row0 + new_src[j]) * p.hist_ld + c];
    for (int j = 0; j < nb; ++j) p.hist[(long)(row0 + j) * p.hist_ld + c] = v[j];
  }
  if (p.row_map) {
    for (int c = tid; c < cur_len; c += 256) {           // keys 0 .. t were written by this step's attention kernels
      int v[16];
      for (int j = 0; j < nb; ++j) v[j] = p.row_map[(long)(row0 + new_src[j]) * p.row_map_ld + c];
      for (int j = 0; j < nb; ++j) p.row_map[(long)(row0 + j) * p.row_map_ld + c] = v[j];
    }
  }
  if (tid < nb) {
    if (cur_len < p.max_length) p.hist[(long)(row0 + tid) * p.hist_ld + cur_len] = new_tok[tid];
    p.next_tok[row0 + tid] = new_tok[tid];
    p.beam_scores[row0 + tid] = new_sc[tid];
    p.src_rows[row0 + tid] = row0 + new_src[tid];
  }
}

extern "C" int v2s_beam_advance(const float* cand_val, const int32_t* cand_tok, int32_t K, int32_t B, int32_t nb, int32_t eos_id, int32_t pad_id,
                                const double* len_pow, const int32_t* pos_dev, int64_t* hist, int64_t hist_ld, int32_t max_length,
                                int32_t* row_map, int64_t row_map_ld, int64_t* next_tok, float* beam_scores, int32_t* src_rows,
                                int32_t* hyp_tok, int32_t* hyp_len, double* hyp_score, int32_t* hyp_order, int32_t* heap_n,
                                double* heap_worst, int32_t* heap_added, int32_t* done, int32_t* ndone, void* stream) {
  V2S_CHECK(cand_val && cand_tok && len_pow && pos_dev && hist && next_tok && beam_scores && src_rows && hyp_tok && hyp_len && hyp_score && hyp_order &&
            heap_n && heap_worst && heap_added && done && ndone, V2S_ERR_ARG, "v2s_beam_advance: null pointer");
  V2S_CHECK(B > 0 && nb >= 1 && nb <= 16 && K >= 2 && K <= 32 && nb * K <= 512 && K * nb >= 2 * nb && max_length > 1, V2S_ERR_SHAPE,
            "v2s_beam_advance: needs 1 <= num_beams <= 16, 2 <= K <= 32 (nb=%d K=%d)", nb, K);
  BeamAdvP p;
  p.cand_val = cand_val; p.cand_tok = cand_tok; p.K = K; p.B = B; p.nb = nb; p.eos = eos_id; p.pad = pad_id; p.len_pow = len_pow;
  p.pos_dev = pos_dev; p.hist = (long*)hist; p.hist_ld = hist_ld; p.max_length = max_length; p.row_map = row_map; p.row_map_ld = row_map_ld;
  p.next_tok = (long*)next_tok; p.beam_scores = beam_scores; p.src_rows = src_rows; p.hyp_tok = hyp_tok; p.hyp_len = hyp_len;
  p.hyp_score = hyp_score; p.hyp_order = hyp_order; p.heap_n = heap_n; p.heap_worst = heap_worst; p.heap_added = heap_added; p.done = done;
  p.ndone = ndone;
  hipLaunchKernelGGL(beam_advance_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, p);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_kv_gather(const void* src, void* dst, const int32_t* idx, int64_t bs, int64_t rs, int32_t B, int32_t len,
                             int32_t width, void* stream) {
  V2S_CHECK(src && dst && idx && B > 0 && len > 0 && width > 0 && (width % 8) == 0 && ((bs | rs) % 8) == 0 && src != dst, V2S_ERR_ARG,
            "v2s_kv_gather: bad args");
  long blocks = ((long)len * (width / 8) + 255) / 256;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(kv_gather_kernel, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, idx,
                     (long)bs, (long)rs, len, width / 8);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_repetition_penalty(float* scores, int64_t ld, int32_t rows, int32_t V, const int64_t* hist, int64_t hist_ld,
                                      const int32_t* pos_dev, int32_t n_static, float penalty, float* row_lse, void* stream) {
  V2S_CHECK(scores && hist && rows > 0 && V > 0 && penalty > 0.f && (pos_dev || n_static > 0), V2S_ERR_ARG, "v2s_repetition_penalty: bad args");
  hipLaunchKernelGGL(rep_penalty_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, scores, (long)ld, V, (const long*)hist, (long)hist_ld,
                     pos_dev, n_static, penalty, row_lse);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

// HF 4.28 MinLengthLogitsProcessor for greedy_search: scores[:, token] = -inf while the decoder sequence (start token + decoded) is
// shorter than min_length; the step counter lives on the device so that one captured graph serves every step
__global__ void ban_token_kernel(float* __restrict__ scores, long ld, int rows, int token, const int* __restrict__ pos_dev, int min_length) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows && *pos_dev + 1 < min_length) scores[(long)r * ld + token] = -INFINITY;
}
extern "C" int v2s_ban_token(float* scores, int64_t ld, int32_t rows, int32_t V, int32_t token, const int32_t* pos_dev, int32_t min_length,
                             void* stream) {
  V2S_CHECK(scores && pos_dev && rows > 0 && token >= 0 && token < V, V2S_ERR_ARG, "v2s_ban_token: bad args");
  hipLaunchKernelGGL(ban_token_kernel, dim3((rows + 63) / 64), dim3(64), 0, (hipStream_t)stream, scores, (long)ld, rows, token, pos_dev, min_length);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_topp_sample_step(const float* logits, int64_t ld, int32_t rows, int32_t V, float top_p, float temperature, uint32_t seed,
                                    int64_t* next_tok, int32_t* unfinished, int32_t eos_id, int32_t pad_id, int64_t* seq_out, int64_t seq_ld,
                                    const int32_t* pos_dev, float* probs_out, int32_t min_length, int32_t top_k, void* stream) {
  V2S_CHECK(logits && next_tok && unfinished && rows > 0 && V > 0 && top_k >= 0, V2S_ERR_ARG, "v2s_topp_sample_step: bad args");
  V2S_CHECK(top_p > 0.f && top_p <= 1.f && temperature > 0.f, V2S_ERR_ARG, "v2s_topp_sample_step: top_p in (0,1], temperature > 0");
  V2S_CHECK(!seq_out || pos_dev, V2S_ERR_ARG, "v2s_topp_sample_step: seq_out needs pos_dev");
  const size_t dyn = ((size_t)V + 256 + 8) * sizeof(float);
  V2S_CHECK(dyn <= 160 * 1024, V2S_ERR_SHAPE, "v2s_topp_sample_step: vocabulary %d too large for the LDS row buffer", V);
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)topp_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(topp_sample_kernel, dim3(rows), dim3(256), dyn, (hipStream_t)stream, logits, (long)ld, V, top_p, 1.0f / temperature, seed,
                     (long*)next_tok, unfinished, eos_id, pad_id, (long*)seq_out, (long)seq_ld, pos_dev, probs_out, min_length, top_k);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_beam_sample_cand(const float* logits, int64_t ld, int32_t rows, int32_t V, int32_t K, const float* beam_scores, float top_p,
                                    float temperature, int32_t top_k, uint32_t seed, float* out_val, int32_t* out_tok, float* out_key,
                                    int32_t ban_token, const int32_t* pos_dev, int32_t min_length, const float* row_lse, void* stream) {
  V2S_CHECK(logits && out_val && out_tok && out_key && rows > 0 && V > 0, V2S_ERR_ARG, "v2s_beam_sample_cand: bad args");
  V2S_CHECK(K >= 1 && K <= 64, V2S_ERR_ARG, "v2s_beam_sample_cand: K must be in [1, 64] (got %d)", K);
  V2S_CHECK(top_p > 0.f && top_p <= 1.f && temperature > 0.f && top_k >= 0 && top_k <= 64, V2S_ERR_ARG,
            "v2s_beam_sample_cand: top_p in (0,1], temperature > 0, top_k in [0, 64] (0 = 64: the kept list of a row has 64 slots)");
  const size_t dyn = ((size_t)V + 8) * sizeof(float) + 65 * sizeof(int);
  V2S_CHECK(dyn <= 160 * 1024, V2S_ERR_SHAPE, "v2s_beam_sample_cand: vocabulary %d too large for the LDS row buffer", V);
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)beam_sample_cand_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(beam_sample_cand_kernel, dim3(rows), dim3(256), dyn, (hipStream_t)stream, logits, (long)ld, V, beam_scores, top_p,
                     1.0f / temperature, top_k, 2, seed, out_val, out_tok, out_key, K, ban_token, pos_dev, min_length, row_lse);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}
