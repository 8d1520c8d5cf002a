// Decode-step cross-attention against the SHARED encoder memory (gfx950, HBM-bound) -- no per-layer cross K/V cache.
//
// The reference projects the encoder memory to K and V in every decoder layer and caches both (model/modeling_t5.py:484-525 with
// past_key_value, 12 layers x 2 x [B, S, 768] for t5-base): a decode step then streams 24 tensors of the memory's size.  Both derive from
// the same rows, so per head h
//     score_h[k] = q_h . (Wk_h mem_k)  = (Wk_h^T q_h) . mem_k          = qp_h . mem_k           (qp_h in R^d: "folded query")
//     ctx_h      = sum_k p_k (Wv_h mem_k) = Wv_h (sum_k p_k mem_k)       = Wv_h accn_h            (accn_h in R^d)
// i.e. all H heads of a query row read the SAME d-wide memory row: one pass over [S, d] per layer instead of K and V ([S, 2 * inner]),
// half the bytes, and the one tensor all twelve layers read is small enough (B = 64: 108 MB) to stay in the 256 MB last-level cache
// between layers.  The price is (H x) more MFMA work -- the matrix pipe idles in a decode step -- and two small per-head GEMMs:
//   v2s_decode_qfold   : qp[m, h, :] = (rstd_m * x_m Wq'_h^T) Wk_h         (RMSNorm folded like the other decode projections)
//   v2s_decode_memattn : flash-style pass over the memory rows of an entry for its G*H query rows, key range split over blocks
//   v2s_decode_ctxfold : merges the splits and applies Wv_h: ctx[m, h*64..] = Wv_h accn[m, h]
// d = 768, head width 64 (t5-base); other widths keep the K/V-cache path (v2s_decode_attn).
#include <math.h>
#include <stdlib.h>
#include "v2s_common.h"

namespace {

constexpr int D = 768;            // model width = memory row
constexpr int DH = 64;            // head width
constexpr int CW = 24;            // 16-byte chunks per wave slice of a row (192 columns)
constexpr int TK = 32;            // keys per tile
constexpr int WTILE = TK * CW * 16;        // 12 KiB: one wave's [32 keys][192 columns] piece of a tile
constexpr int STAGE = 4 * WTILE;           // 48 KiB

__device__ __forceinline__ uint32_t ma_lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)p);
}
// LDS image of a wave's tile piece: row r (key), chunk c (0..23) at (r * 24 + swz(c, r)) * 16.  The swizzle keeps both fragment reads
// conflict-free: the 16 rows of a score fragment (rows 8a + 4t + b) differ in (r0, r1, r3, r4) -> r0 flips bit 3 of the slot through
// the 24-chunk row stride, (r1, r3, r4) go to chunk bits (1, 2, 0); the 4 rows x 2 chunks of a transposing read differ in r0, r1.
__device__ __forceinline__ int ma_swz(int c, int r) {
  const int f = (((r >> 1) & 1) << 1) | (((r >> 3) & 1) << 2) | ((r >> 4) & 1);
  return c ^ f;                  // f < 8: stays inside the aligned group of 8 chunks
}

struct MemAttnP {
  const bf16_t* qp;                // [entries * R][D]
  const bf16_t* mem; long mem_es;  // [entries][S][D], entry stride in elements
  const int4* blk;                 // [gridDim.x]: (entry, first tile | end tile << 16, output slot, valid keys of the entry)
  bf16_t* part;                    // [slots][QT*16][D]: per-piece normalised sums
  float* ml;                       // [slots][QT*16][2]: running max (log2 domain), sum of weights
  int R, qr;                       // query rows per entry, padded to 16 (row stride of part / ml slots)
  float scale_log2;                // scale * log2(e)
  int dbg;
};

// One block (4 waves, the only block of its CU: 152 KiB of LDS) per piece of an entry's key range (v2s_decode_memattn_plan).  Wave w owns columns [192 w, 192 w + 192) of the
// memory rows for BOTH products: its share of the contraction of the scores, and its own output columns of the weighted sum -- so a
// tile piece is fetched (LDS-DMA), waited for and read by one wave only; the four score partials are summed through LDS (two
// barriers per tile).  Transposed formulation like decode_attn_mfma_kernel: S^T = tile . qp^T (fragment row q of key tile t stands
// for key 8 (q >> 2) + 4 t + (q & 3), so lane (q, g) ends up with keys 8 g .. 8 g + 7 of query q: the layout the second product wants
// its B operand in), acc^T = tile^T . P^T with the tile's transpose from ds_read_b64_tr_b16, P in bf16 (like the training kernels).
__device__ __forceinline__ void ma_dma12(const uint32_t (&o)[12], const char* src, uint32_t dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %[keep], m0\n\t"
      "s_mov_b32 m0, %[dst]\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[o0], %[src]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[o1], %[src]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[o2], %[src]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[o3], %[src]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[o4], %[src]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[o5], %[src]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[o6], %[src]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[o7], %[src]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[o8], %[src]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[o9], %[src]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[o10], %[src]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[o11], %[src]\n\t"
      "s_mov_b32 m0, %[keep]"
      : [keep] "=&s"(keep)
      : [dst] "s"(dst), [src] "s"(src), [o0] "v"(o[0]), [o1] "v"(o[1]), [o2] "v"(o[2]), [o3] "v"(o[3]), [o4] "v"(o[4]), [o5] "v"(o[5]),
        [o6] "v"(o[6]), [o7] "v"(o[7]), [o8] "v"(o[8]), [o9] "v"(o[9]), [o10] "v"(o[10]), [o11] "v"(o[11])
      : "memory", "scc");
}

template <int QT>
__global__ __launch_bounds__(256, 1) void mem_attn_kernel(const MemAttnP p) {
  constexpr int NST = 3;                         // tiles in the ring (LDS: 3 * 48 KiB + QT * 8 KiB of score partials)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_part = reinterpret_cast<float*>(smem + NST * STAGE);          // [4 waves][QT][2 x f32x4][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane & 15, g = lane >> 4;
  const int4 bk = p.blk[blockIdx.x];                          // (entry, first tile | end tile << 16, output slot, valid keys of the entry)
  const int ent = bk.x, t0 = bk.y & 0xffff, n = (bk.y >> 16) - t0, klen = bk.w;      // n >= 1: the plan never emits an empty piece
  const int qt0 = blockIdx.y * QT;                            // first query tile (of 16 rows) of this block
  const long orow = (long)bk.z * p.qr + qt0 * 16;
  // LDS-DMA source offsets of the wave's 12 KiB piece (12 instructions of 64 x 16 B): position P = j * 64 + lane of the image
  uint32_t doff[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    const int P = j * 64 + lane, r = P / CW, cp = P % CW;
    doff[j] = (uint32_t)(r * (D * 2) + (wave * CW + ma_swz(cp, r)) * 16);
  }
  const char* mbase = reinterpret_cast<const char*>(p.mem + (long)ent * p.mem_es);
  const uint32_t lds0 = ma_lds_addr(smem) + wave * WTILE;
  auto issue = [&](int i) {                       // tile i of this block -> ring stage i % NST
    const int key0 = (t0 + i) * TK;
    const char* src = mbase + (long)key0 * (D * 2);
    const uint32_t dst = lds0 + (i % NST) * STAGE;
    const int last = klen - 1 - key0;             // rows past the last valid key re-read it (scored -inf below)
    if (last >= TK - 1) {
      ma_dma12(doff, src, dst);
    } else {
      uint32_t o[12];
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const int r = (j * 64 + lane) / CW;
        o[j] = doff[j] - (uint32_t)((r > last ? r - last : 0) * (D * 2));
      }
      ma_dma12(o, src, dst);
    }
  };
  // the first tiles are requested before anything else is loaded: their latency covers the query fragments'
  issue(0);
  if (n > 1) issue(1);
  // folded queries as B operands: column = query row, 8 consecutive k per lane, this wave's 6 k-steps.  (These loads are younger
  // than the DMAs above: the counted waits below only ever become stricter by them.)
  bf16x8 qf[QT][6];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int r = (qt0 + qt) * 16 + q;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (r < p.R) v = *reinterpret_cast<const uint4*>(p.qp + ((long)ent * p.R + r) * D + wave * 192 + ks * 32 + g * 8);
      qf[qt][ks] = __builtin_bit_cast(bf16x8, v);
    }
  }
  float m[QT], l[QT];
  f32x4 acc[QT][12];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m[qt] = -INFINITY; l[qt] = 0.f;
#pragma unroll
    for (int ct = 0; ct < 12; ++ct) acc[qt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int i = 0; i < n; ++i) {
    // this wave's reads of the stage being refilled were consumed by MFMAs of the previous iteration: wave-private, no barrier
    if (i + NST - 1 < n && p.dbg != 2) issue(i + NST - 1);
    if (p.dbg == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (i + 2 < n) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (i + 1 < n) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.dbg == 1) continue;
    const char* tile = smem + (p.dbg == 2 ? 0 : (i % NST)) * STAGE + wave * WTILE;
    // ---- partial scores over this wave's 192 columns
    f32x4 s[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { s[qt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; s[qt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const int row = (q >> 2) * 8 + 4 * kt + (q & 3), c = ks * 4 + g;
        const bf16x8 a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(tile + (row * CW + ma_swz(c, row)) * 16));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[qt][ks], s[qt][kt], 0, 0, 0);
      }
    }
    __syncthreads();                                      // every wave has read the previous tile's partials
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
        *reinterpret_cast<f32x4*>(s_part + (((wave * QT + qt) * 2 + kt) * 64 + lane) * 4) = s[qt][kt];
    __syncthreads();
    const int kc = (t0 + i) * TK;
    bf16x8 ph[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {                      // fixed order: the four waves compute identical sums
        s0 += *reinterpret_cast<const f32x4*>(s_part + (((w * QT + qt) * 2 + 0) * 64 + lane) * 4);
        s1 += *reinterpret_cast<const f32x4*>(s_part + (((w * QT + qt) * 2 + 1) * 64 + lane) * 4);
      }
      float sv[8];                                        // log2 domain: exp(x) = exp2(x log2 e)
#pragma unroll
      for (int e = 0; e < 8; ++e) sv[e] = (e < 4 ? s0[e & 3] : s1[e & 3]) * p.scale_log2;
      if (kc + TK > klen) {                               // only the last tile of an entry has keys past the valid prefix
#pragma unroll
        for (int e = 0; e < 8; ++e) sv[e] = kc + 8 * g + e < klen ? sv[e] : -INFINITY;
      }
      float mx = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m[qt], mx);                  // finite: a visited tile has a key < klen
      float pr[8], ps = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { pr[e] = __builtin_amdgcn_exp2f(sv[e] - mn); ps += pr[e]; }
      if (__builtin_amdgcn_ballot_w64(mn != m[qt])) {     // the running maximum settles after a few tiles: no rescale then
        const float alpha = __builtin_amdgcn_exp2f(m[qt] - mn);
        l[qt] *= alpha;
#pragma unroll
        for (int ct = 0; ct < 12; ++ct) acc[qt][ct] *= alpha;
        m[qt] = mn;
      }
      l[qt] += ps;
      const uint4 uh = make_uint4(pack2bf(pr[0], pr[1]), pack2bf(pr[2], pr[3]), pack2bf(pr[4], pr[5]), pack2bf(pr[6], pr[7]));
      ph[qt] = __builtin_bit_cast(bf16x8, uh);
    }
    // ---- weighted sum of this wave's columns: A = tile^T (row = column 16 ct + q, 8 consecutive keys 8 g .. 8 g + 7)
#pragma unroll
    for (int ct = 0; ct < 12; ++ct) {
      const int r0 = g * 8 + (q >> 2), r1 = r0 + 4, c = 2 * ct + ((q & 3) >> 1), hb8 = (q & 1) * 8;
      const s16x4 vlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((V2S_LDS s16x4*)(tile + (r0 * CW + ma_swz(c, r0)) * 16 + hb8));
      const s16x4 vhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((V2S_LDS s16x4*)(tile + (r1 * CW + ma_swz(c, r1)) * 16 + hb8));
      const s16x8 vv = {vlo[0], vlo[1], vlo[2], vlo[3], vhi[0], vhi[1], vhi[2], vhi[3]};
      const bf16x8 a = __builtin_bit_cast(bf16x8, vv);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) acc[qt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, ph[qt], acc[qt][ct], 0, 0, 0);
    }
  }
  // ---- write the split's partial: lane (query q, group g) holds columns 192 w + 16 ct + 4 g + 0..3
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float ls = l[qt];
    ls += __shfl_xor(ls, 16, 64);
    ls += __shfl_xor(ls, 32, 64);
    const float inv = 1.0f / ls;
    const long row = orow + qt * 16 + q;
    if (wave == 0 && g == 0) { p.ml[row * 2] = m[qt]; p.ml[row * 2 + 1] = ls; }
#pragma unroll
    for (int ct = 0; ct < 12; ++ct) {
      const uint2 o = make_uint2(pack2bf(acc[qt][ct][0] * inv, acc[qt][ct][1] * inv), pack2bf(acc[qt][ct][2] * inv, acc[qt][ct][3] * inv));
      *reinterpret_cast<uint2*>(p.part + row * D + wave * 192 + ct * 16 + g * 4) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- folded queries
struct QFoldP {
  const bf16_t* x; long ldx; int rows;
  const bf16_t* wq;       // [H * 64][D]  (RMSNorm weight folded into the columns when rms_eps > 0)
  const bf16_t* wkT;      // [D][H * 64]  (transpose of the K projection)
  bf16_t* qp;             // [rows][H][D]
  int H, m16;             // row tiles of 16
  float rms_eps;
};

// One block per (head, row tile of 16, slice of 16 * 4 * NT columns).  Stage 1: q_h^T[64 x 16] = Wq'_h x^T, the contraction (768) split
// over the four waves (6 k-steps each: all 30 fragment loads of a wave in flight at once -- the kernel is a chain of load latencies,
// not of flops) and summed through LDS together with the rows' sums of squares (RMSNorm) taken from the x fragments.  Stage 2:
// qp^T[slice x 16] = Wk_h^T q_h^T over K = 64, NT column tiles per wave, its Wk^T fragments requested before stage 1 starts.
// Both products are transposed so that the summed stage-1 accumulators ARE the stage-2 B operand: lane (row m, group g) holds
// q[m][16 t + 4 g + i], the contraction index (ks, e) of stage 2 stands for j = 32 ks + 16 (e >> 2) + 4 g + (e & 3), and the Wk^T
// fragment is read with the same map.
template <int NT>
__global__ __launch_bounds__(256) void qfold_kernel(const QFoldP p) {
  __shared__ __attribute__((aligned(16))) float red[4][4][64][4];
  __shared__ float ssr[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane & 15, g = lane >> 4;
  const int h = blockIdx.x;
  const int rt = blockIdx.y % p.m16, sl = blockIdx.y / p.m16;
  const int mrow = rt * 16 + q;
  const bool live = mrow < p.rows;
  const int c0 = sl * (64 * NT) + wave * (16 * NT);          // this wave's first output column
  // stage-2 operands first (they do not depend on stage 1)
  uint2 kf[NT][2][2];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    const bf16_t* kr = p.wkT + (long)(c0 + ct * 16 + q) * (p.H * DH) + h * DH + 4 * g;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[ct][ks][0] = *reinterpret_cast<const uint2*>(kr + 32 * ks);
      kf[ct][ks][1] = *reinterpret_cast<const uint2*>(kr + 32 * ks + 16);
    }
  }
  const bf16_t* xr = p.x + (long)(live ? mrow : 0) * p.ldx + wave * 192 + g * 8;
  const bf16_t* wr = p.wq + ((long)h * DH + q) * D + wave * 192 + g * 8;
  uint4 xv[6], wf[6][4];
#pragma unroll
  for (int ks = 0; ks < 6; ++ks) {
    xv[ks] = live ? *reinterpret_cast<const uint4*>(xr + ks * 32) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) wf[ks][t] = *reinterpret_cast<const uint4*>(wr + (long)t * 16 * D + ks * 32);
  }
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ss = 0.f;
#pragma unroll
  for (int ks = 0; ks < 6; ++ks) {
    float xf[8];
    unpack8(xv[ks], xf);
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += xf[e] * xf[e];
    const bf16x8 b = __builtin_bit_cast(bf16x8, xv[ks]);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ks][t]), b, acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(&red[wave][t][lane][0]) = acc[t];
  ssr[wave][lane] = ss;
  __syncthreads();
  float st = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    st += ssr[w][lane];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] += *reinterpret_cast<const f32x4*>(&red[w][t][lane][0]);
  }
  float rstd = 1.f;
  if (p.rms_eps > 0.f) {
    st += __shfl_xor(st, 16, 64);
    st += __shfl_xor(st, 32, 64);
    rstd = rsqrtf(st * (1.0f / D) + p.rms_eps);
  }
  bf16x8 qb[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const f32x4 a0 = acc[2 * ks], a1 = acc[2 * ks + 1];
    const uint4 u = make_uint4(pack2bf(a0[0] * rstd, a0[1] * rstd), pack2bf(a0[2] * rstd, a0[3] * rstd),
                               pack2bf(a1[0] * rstd, a1[1] * rstd), pack2bf(a1[2] * rstd, a1[3] * rstd));
    qb[ks] = __builtin_bit_cast(bf16x8, u);
  }
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4 u = make_uint4(kf[ct][ks][0].x, kf[ct][ks][0].y, kf[ct][ks][1].x, kf[ct][ks][1].y);
      o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, u), qb[ks], o, 0, 0, 0);
    }
    if (live)
      *reinterpret_cast<uint2*>(p.qp + ((long)mrow * p.H + h) * D + c0 + ct * 16 + 4 * g) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
  }
}

// ---------------------------------------------------------------------------------------------------------------- merge + Wv
struct CtxFoldP {
  const bf16_t* part; const float* ml;
  const int* slot_off;    // [entries + 1]: the pieces of entry e are slots [slot_off[e], slot_off[e + 1])
  const bf16_t* wv;       // [H * 64][D]
  bf16_t* ctx; long ld_ctx;
  int rows, G, H, qr;     // qr = padded query rows per slot in part / ml
};

// One block per (head, row tile of 16): ctx_h^T[64 x 16] = Wv_h accn_h^T with the pieces merged on the fly (weights
// l_s exp2(m_s - max) / sum: per lane, its row's; kept in LDS).  The four waves split the contraction (192 columns each) and are
// summed through LDS.  The piece loop is unrolled by four: 24 independent 16-byte loads in flight per lane.
__global__ __launch_bounds__(256) void ctxfold_kernel(const CtxFoldP p) {
  __shared__ __attribute__((aligned(16))) float red[4][4][64][4];
  __shared__ float wsh[16][65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane & 15, g = lane >> 4;
  const int h = blockIdx.x, rt = blockIdx.y;
  const int mrow = rt * 16 + q;
  const bool live = mrow < p.rows;
  const int ent = live ? mrow / p.G : 0, r = live ? (mrow % p.G) * p.H + h : 0;
  const int s0 = p.slot_off[ent], ns = p.slot_off[ent + 1] - s0;         // 1 <= ns <= 64
  const long prow0 = (long)s0 * p.qr + r;                                 // + s * qr
  const bf16_t* wr = p.wv + ((long)h * DH + q) * D + wave * 192 + g * 8;
  uint4 wf[6][4];
#pragma unroll
  for (int ks = 0; ks < 6; ++ks)
#pragma unroll
    for (int t = 0; t < 4; ++t) wf[ks][t] = *reinterpret_cast<const uint4*>(wr + (long)t * 16 * D + ks * 32);
  if (wave == 0) {                                    // merge weights of the 16 rows: lanes (q, g) take pieces g, g + 4, ...
    float mmax = -INFINITY;
    for (int s = g; s < ns; s += 4) mmax = fmaxf(mmax, p.ml[(prow0 + (long)s * p.qr) * 2]);
    mmax = fmaxf(mmax, __shfl_xor(mmax, 16, 64));
    mmax = fmaxf(mmax, __shfl_xor(mmax, 32, 64));
    float wsum = 0.f;
    for (int s = g; s < ns; s += 4) {
      const float* e = p.ml + (prow0 + (long)s * p.qr) * 2;
      const float w = e[1] * __builtin_amdgcn_exp2f(e[0] - mmax);
      wsh[q][s] = w;
      wsum += w;
    }
    wsum += __shfl_xor(wsum, 16, 64);
    wsum += __shfl_xor(wsum, 32, 64);
    if (g == 0) wsh[q][64] = live ? 1.0f / wsum : 0.f;
  }
  __syncthreads();
  const float winv = wsh[q][64];
  float a8[6][8];
#pragma unroll
  for (int ks = 0; ks < 6; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) a8[ks][e] = 0.f;
  const bf16_t* pr = p.part + prow0 * D + wave * 192 + g * 8;
  const long pstep = (long)p.qr * D;
  int s = 0;
  for (; s + 4 <= ns; s += 4) {
    uint4 v[4][6];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int ks = 0; ks < 6; ++ks) v[u][ks] = *reinterpret_cast<const uint4*>(pr + (s + u) * pstep + ks * 32);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float w = wsh[q][s + u] * winv;
#pragma unroll
      for (int ks = 0; ks < 6; ++ks) {
        float f[8];
        unpack8(v[u][ks], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) a8[ks][e] += w * f[e];
      }
    }
  }
  for (; s < ns; ++s) {
    uint4 v[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) v[ks] = *reinterpret_cast<const uint4*>(pr + s * pstep + ks * 32);
    const float w = wsh[q][s] * winv;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
      float f[8];
      unpack8(v[ks], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) a8[ks][e] += w * f[e];
    }
  }
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 6; ++ks) {
    const uint4 bu = pack8(a8[ks]);
    const bf16x8 b = __builtin_bit_cast(bf16x8, bu);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ks][t]), b, acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(&red[wave][t][lane][0]) = acc[t];
  __syncthreads();
  f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < 4; ++w) o += *reinterpret_cast<const f32x4*>(&red[w][wave][lane][0]);
  if (live)
    *reinterpret_cast<uint2*>(p.ctx + (long)mrow * p.ld_ctx + h * DH + wave * 16 + 4 * g) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
}

}  // namespace

extern "C" int v2s_decode_qfold(const void* x, int64_t ldx, int32_t rows, const void* wq, const void* wkT, float rms_eps, void* qp,
                                int32_t H, int32_t d, void* stream) {
  V2S_CHECK(x && wq && wkT && qp, V2S_ERR_ARG, "v2s_decode_qfold: null pointer");
  V2S_CHECK(d == D && H > 0 && rows > 0 && (ldx % 8) == 0, V2S_ERR_SHAPE, "v2s_decode_qfold: needs d == 768, ldx %% 8 == 0 (d=%d rows=%d)", d, rows);
  QFoldP p;
  p.x = (const bf16_t*)x; p.ldx = ldx; p.rows = rows; p.wq = (const bf16_t*)wq; p.wkT = (const bf16_t*)wkT; p.qp = (bf16_t*)qp;
  p.H = H; p.m16 = (rows + 15) / 16; p.rms_eps = rms_eps;
  hipStream_t s = (hipStream_t)stream;
  // >= 144 blocks: slices of 192 columns (3 tiles per wave) from four row tiles on, 64 columns (one tile per wave) below
  if (p.m16 >= 4) hipLaunchKernelGGL(qfold_kernel<3>, dim3(H, p.m16 * 4), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(qfold_kernel<1>, dim3(H, p.m16 * 12), dim3(256), 0, s, p);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

// Host-side plan of v2s_decode_memattn (no GPU work): cuts every entry's ceil(klen / 32) key tiles into pieces of at most `tpb` tiles,
// tpb = the smallest value for which the pieces number <= target_blocks (one block per CU and launch: 152 KiB of LDS each), so that
// all blocks of the launch are equally long whatever the entries' lengths.  blk[i] = (entry, first tile | end tile << 16, slot, klen);
// slot_off[e] .. slot_off[e + 1] = the entry's slots.  Returns the number of blocks through *nblk (<= max_blocks, else an error).
extern "C" int v2s_decode_memattn_plan(const int32_t* klen_host, int32_t entries, int32_t target_blocks, int32_t max_blocks, int32_t* blk,
                                       int32_t* slot_off, int32_t* nblk) {
  V2S_CHECK(klen_host && blk && slot_off && nblk && entries > 0 && target_blocks > 0, V2S_ERR_ARG, "v2s_decode_memattn_plan: bad arguments");
  long total = 0;
  int ntmax = 0;
  for (int e = 0; e < entries; ++e) {
    V2S_CHECK(klen_host[e] >= 1 && klen_host[e] <= 32 * 65535, V2S_ERR_SHAPE, "v2s_decode_memattn_plan: klen[%d] = %d (needs >= 1)", e, klen_host[e]);
    const int nt = (klen_host[e] + TK - 1) / TK;
    total += nt;
    ntmax = nt > ntmax ? nt : ntmax;
  }
  int tpb = (int)((total + target_blocks - 1) / target_blocks);
  if (tpb < 1) tpb = 1;
  for (;; ++tpb) {
    long nb = 0;
    bool ok = true;
    for (int e = 0; e < entries; ++e) {
      const int nt = (klen_host[e] + TK - 1) / TK, sp = (nt + tpb - 1) / tpb;
      if (sp > 64) ok = false;
      nb += sp;
    }
    if ((ok && nb <= (entries > target_blocks ? entries : target_blocks)) || tpb >= ntmax) break;
  }
  int b = 0;
  for (int e = 0; e < entries; ++e) {
    const int nt = (klen_host[e] + TK - 1) / TK, sp = (nt + tpb - 1) / tpb;
    V2S_CHECK(sp <= 64, V2S_ERR_SHAPE, "v2s_decode_memattn_plan: entry %d needs %d pieces (> 64)", e, sp);
    slot_off[e] = b;
    for (int s = 0; s < sp; ++s, ++b) {
      V2S_CHECK(b < max_blocks, V2S_ERR_SHAPE, "v2s_decode_memattn_plan: more than %d blocks", max_blocks);
      const int t0 = (int)((long)s * nt / sp), t1 = (int)((long)(s + 1) * nt / sp);
      blk[4 * b] = e; blk[4 * b + 1] = t0 | (t1 << 16); blk[4 * b + 2] = b; blk[4 * b + 3] = klen_host[e];
    }
  }
  slot_off[entries] = b;
  *nblk = b;
  return V2S_OK;
}

extern "C" int v2s_decode_memattn(const void* qp, const void* mem, int64_t mem_es, const int32_t* blk, int32_t nblk, int32_t R,
                                  float scale, void* part, float* ml, int32_t d, void* stream) {
  V2S_CHECK(qp && mem && blk && part && ml, V2S_ERR_ARG, "v2s_decode_memattn: null pointer");
  V2S_CHECK(d == D && nblk > 0 && R > 0 && R <= 48, V2S_ERR_SHAPE,
            "v2s_decode_memattn: needs d == 768, 1 <= query rows per entry <= 48 (d=%d R=%d)", d, R);
  MemAttnP p;
  p.qp = (const bf16_t*)qp; p.mem = (const bf16_t*)mem; p.mem_es = mem_es; p.blk = (const int4*)blk; p.part = (bf16_t*)part; p.ml = ml;
  p.R = R; p.scale_log2 = scale * 1.4426950408889634f;
  { static int dbg = -1; if (dbg < 0) { const char* e = getenv("V2S_MEMATTN_DBG"); dbg = e ? atoi(e) : 0; } p.dbg = dbg; }
  // query tiles of 16 rows per block: one (R <= 16), two (R <= 32), else one per block and a grid row per tile -- three tiles in
  // one block (144 accumulators + 72 query-fragment registers) spill, and scratch traffic would break the counted DMA waits
  const int QT = R <= 16 ? 1 : (R <= 32 ? 2 : 1);
  p.qr = (R + 15) / 16 * 16;
  const size_t lds = (size_t)3 * STAGE + (size_t)4 * QT * 2 * 64 * 16;
  const dim3 grid(nblk, (p.qr / 16 + QT - 1) / QT), block(256);
  hipStream_t s = (hipStream_t)stream;
#define V2S_MA(QTV)                                                                                                         \
  {                                                                                                                         \
    static bool attr_set = false;                                                                                           \
    if (!attr_set) {                                                                                                        \
      if (hipFuncSetAttribute((const void*)mem_attn_kernel<QTV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
        v2s_set_error("v2s_decode_memattn: cannot raise the dynamic LDS limit to %zu", lds);                                \
        return V2S_ERR_LAUNCH;                                                                                              \
      }                                                                                                                     \
      attr_set = true;                                                                                                      \
    }                                                                                                                       \
    hipLaunchKernelGGL(mem_attn_kernel<QTV>, grid, block, lds, s, p);                                                       \
  }
  if (QT == 1) V2S_MA(1) else V2S_MA(2)
#undef V2S_MA
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_decode_ctxfold(const void* part, const float* ml, const int32_t* slot_off, int32_t rows, int32_t G, int32_t H,
                                  const void* wv, void* ctx, int64_t ld_ctx, int32_t d, void* stream) {
  V2S_CHECK(part && ml && slot_off && wv && ctx, V2S_ERR_ARG, "v2s_decode_ctxfold: null pointer");
  V2S_CHECK(d == D && rows > 0 && G > 0 && (rows % G) == 0 && G * H <= 48 && (ld_ctx % 4) == 0, V2S_ERR_SHAPE,
            "v2s_decode_ctxfold: needs d == 768, rows %% G == 0, G * H <= 48 (rows=%d G=%d H=%d)", rows, G, H);
  CtxFoldP p;
  p.part = (const bf16_t*)part; p.ml = ml; p.slot_off = slot_off; p.wv = (const bf16_t*)wv; p.ctx = (bf16_t*)ctx; p.ld_ctx = ld_ctx;
  p.rows = rows; p.G = G; p.H = H; p.qr = (G * H + 15) / 16 * 16;
  hipLaunchKernelGGL(ctxfold_kernel, dim3(H, (rows + 15) / 16), dim3(256), 0, (hipStream_t)stream, p);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}
