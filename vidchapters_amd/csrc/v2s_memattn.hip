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
#include "v2s_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // (a native vector: HIP's uint4 struct is copied by memcpy, which keeps arrays of it in scratch)
constexpr int D = 768;            // model width = memory row
constexpr int DH = 64;            // head width
constexpr int RC = D / 8;         // 16-byte chunks per row (96)
constexpr int TK = 32;            // keys per plan tile (a block takes whole tiles; a wave works on 16 keys at a time)
constexpr int QS_BYTES = 16 * D * 2;       // 24 KiB: the block's 16 folded query rows
constexpr int TB_BYTES = 16 * D * 2;       // 24 KiB per wave: its current [16 keys][768] piece, staged for the transposing read

// LDS images, 16-byte chunk c (0..95) of row r at (r * 96 + swz(c, r)) * 16 (rows are 1536 B = 6 x 256 B apart: without a swizzle all
// rows of a chunk column share their banks).  Both swizzles permute the low 4 chunk bits by a bijection of the row's 4 bits, so the 16 rows
// of one chunk column (a ds_read_b128 lane group: the score fragments of both images) land in 16 different 16-byte slots.  The tile
// image additionally sends key bits (0, 1) to chunk bits (2, 3): the 4 keys x 2 chunks x 2 halves of a ds_read_b64_tr_b16 lane group are
// conflict-free too, and so are its writes (8 consecutive lanes = 8 consecutive chunks of one row).
__device__ __forceinline__ int q_swz(int c, int r) { return c ^ (r & 15); }
__device__ __forceinline__ int t_swz(int c, int r) { return c ^ (((r & 3) << 2) | ((r >> 2) & 3)); }

struct MemAttnP {
  const bf16_t* qp;                // [entries * R][D]
  const bf16_t* mem; long mem_es;  // [entries][S][D], entry stride in elements
  const int4* blk;                 // [gridDim.x]: (entry, first tile | end tile << 16, output slot, valid keys of the entry)
  bf16_t* part;                    // [slots][qr][D]: per-piece normalised sums
  float* ml;                       // [slots][qr][2]: running max (log2 domain), sum of weights
  int R, qr;                       // query rows per entry, padded to 16 (row stride of part / ml slots)
  float scale_log2;                // scale * log2(e)
};

// One block (4 waves, one per SIMD: 192 accumulators in AGPRs + ~230 other registers) per (piece of an entry's key range --
// v2s_decode_memattn_plan --, 16 query rows).  Every wave is an independent flash-attention stream over its own 16-key groups of the
// piece (groups w, w + 4, ...) and ALL 768 columns; no barrier and no hand-off between waves inside the loop:
//   * the group's 16 rows are one contiguous 24 KiB block: 24 coalesced 1 KiB loads into 96 registers, staged to a swizzled LDS copy as
//     soon as they arrive; the copy doubles as register spill space -- the registers take the NEXT group right away (24 KiB per wave
//     in flight during the whole iteration) and the products read their operands from LDS;
//   * S^T[16 keys x 16 queries] = group . qp^T: 24 MFMAs (16x16x32) against the folded queries in LDS; lane (query, g) then holds the
//     scores of keys 4 g .. 4 g + 3 -- which is the B layout of v_mfma_f32_16x16x16_bf16, so the weights go from the softmax
//     registers into the second product as they are;
//   * acc^T[768 x 16 queries] += group^T . P^T: the transpose comes from the LDS copy through ds_read_b64_tr_b16 (one read per MFMA),
//     48 column tiles, the reads of the next eight tiles requested before the MFMAs of these eight;
//   * the exponent reference of a stream moves only on jumps of more than 2^64 (see below): the accumulators are never rescaled.
// The four streams are merged through LDS (bf16: the cut of an entry into pieces and streams is a function of the entry alone, so
// every rounding is too) and written as one normalised partial per piece; pieces are merged by ctxfold_kernel.
__global__ __launch_bounds__(256, 1) void mem_attn_kernel(const MemAttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane & 15, g = lane >> 4;
  const int4 bk = p.blk[blockIdx.x];
  const int ent = bk.x, t0 = bk.y & 0xffff, t1 = bk.y >> 16, klen = bk.w;
  const int qt0 = blockIdx.y;                                 // which 16 of the entry's query rows
  const long orow = (long)bk.z * p.qr + qt0 * 16;
  const int kbase = t0 * TK, kend = min(klen, t1 * TK);      // kbase < kend: the plan never emits an empty piece
  char* qs = smem;
  char* tbw = smem + QS_BYTES + wave * TB_BYTES;
  const char* mbase = reinterpret_cast<const char*>(p.mem + (long)ent * p.mem_es);
  // global -> registers: the 16 rows of a piece are one contiguous 24 KiB block; instruction i fetches its chunks 64 i .. 64 i + 63
  // (1 KiB, whole cache lines -- fragment-shaped loads, 16 rows x 64 B per instruction, ran at half this rate).  Rows past the valid
  // prefix re-read the entry's last row (their weight is 0 below).
#define MA_LOAD(key0_)                                                                                                       \
  {                                                                                                                          \
    const int last_ = klen - 1 - (key0_);                                                                                    \
    const char* src_ = mbase + (long)(key0_) * (D * 2);             /* wave-uniform */                                       \
    if (last_ >= 15) {                                                                                                       \
      _Pragma("unroll") for (int i = 0; i < 24; ++i) t[i] = *reinterpret_cast<const u32x4*>(src_ + (uint32_t)(lane * 16 + i * 1024)); \
    } else {                                                                                                                 \
      _Pragma("unroll") for (int i = 0; i < 24; ++i) {                                                                      \
        /* row of chunk 64 i + lane: (64 i) / 96, one more from lane 96 - (64 i) % 96 on (compile-time threshold: a mask, no register) */ \
        const int r_ = (i * 64) / RC + (lane >= RC - (i * 64) % RC ? 1 : 0);                                                 \
        const int back_ = last_ < 0 ? r_ - last_ : (r_ > last_ ? r_ - last_ : 0);   /* rows to step back to the last valid one */ \
        t[i] = *reinterpret_cast<const u32x4*>(src_ + (long)(lane * 16 + i * 1024 - back_ * (D * 2)));                      \
      }                                                                                                                      \
    }                                                                                                                        \
  }
  // the block's folded queries (rows beyond R read as zero): requested first -- they are small and the barrier below waits for them
  uint4 qv[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int idx = tid + 256 * i, r = idx / RC, c = idx % RC;
    const int gr = qt0 * 16 + r;
    qv[i] = make_uint4(0, 0, 0, 0);
    if (gr < p.R) qv[i] = *reinterpret_cast<const uint4*>(p.qp + ((long)ent * p.R + gr) * D + c * 8);
  }
  u32x4 t[24];                                                // the wave's current 16-key piece, 24 x 1 KiB (96 registers)
  int j = wave;
  bool have = kbase + 16 * j < kend;
  if (have) MA_LOAD(kbase + 16 * j)
  // (everything that does not depend on the loads goes before the barrier: it runs under their latency)
  float mref = -INFINITY, l = 0.f;
  f32x4 acc[48];
#pragma unroll
  for (int ct = 0; ct < 48; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  // LDS addresses as (a few per-lane bases) + compile-time offsets (one address register per access would not fit).  Reads: the
  // swizzles only touch the low 4 chunk bits, so chunk 16 a + b is base[b] + a * 256 bytes.  Writes: instruction i = 3 m + e puts chunk
  // ce(lane) of row 2 m + re(lane); the swizzle term of that row is K(m) ^ (re << 2) with K(m) = ((m & 1) << 3) | (m >> 1) known at
  // compile time, so the address is (w3[e] ^ (K(m) << 4)) + m * 3072.
  uint32_t w3[3], ab[4], qb[4], rb[8];
  {
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const int P = e * 64 + lane, re = P / RC, ce = P % RC;
      w3[e] = (uint32_t)((re * RC + (ce ^ (re << 2))) * 16);
    }
    const int r0 = 4 * g + (q >> 2), x = (q & 3) >> 1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ab[e] = (uint32_t)((q * RC + t_swz(e * 4 + g, q)) * 16);          // score A fragment: row = key q, chunk 16 a + 4 e + g
      qb[e] = (uint32_t)((q * RC + q_swz(e * 4 + g, q)) * 16);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) rb[e] = (uint32_t)((r0 * RC + t_swz(2 * e + x, r0)) * 16 + (q & 1) * 8);   // transposing read: chunk 16 a + 2 e + x
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int idx = tid + 256 * i, r = idx / RC, c = idx % RC;
    *reinterpret_cast<uint4*>(qs + (r * RC + q_swz(c, r)) * 16) = qv[i];
  }
  __syncthreads();
  while (have) {
    const int key0 = kbase + 16 * j;
    asm volatile("" : "+v"(w3[0]), "+v"(w3[1]), "+v"(w3[2]));       // (opaque: keeps the 24 xor-ed addresses from being hoisted into registers)
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      const int m = i / 3, K = ((m & 1) << 3) | (m >> 1);
      *reinterpret_cast<u32x4*>(tbw + (w3[i % 3] ^ (uint32_t)(K << 4)) + m * 3072) = t[i];
    }
    // the staged copy doubles as register spill space: the piece's registers take the NEXT piece right away (in flight during this
    // whole iteration), and the score product reads its A fragments back from LDS (each lane exactly what it wrote)
    j += 4;
    have = kbase + 16 * j < kend;
    if (have) MA_LOAD(kbase + 16 * j)           // (an unconditional request past the end cost 24 KiB per wave and block: +25 % HBM traffic by PMC)
    f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, sb = sa;
#pragma unroll
    for (int ks = 0; ks < 24; ks += 2) {
      const bf16x8 a0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(tbw + ab[ks & 3] + (ks >> 2) * 256));
      const bf16x8 a1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(tbw + ab[(ks + 1) & 3] + (ks >> 2) * 256));
      const bf16x8 b0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(qs + qb[ks & 3] + (ks >> 2) * 256));
      const bf16x8 b1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(qs + qb[(ks + 1) & 3] + (ks >> 2) * 256));
      sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, sa, 0, 0, 0);
      sb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, sb, 0, 0, 0);
      if ((ks & 3) == 2) __builtin_amdgcn_sched_barrier(0);      // keep the fragment reads from being hoisted wholesale (registers)
    }
    float sv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sv[i] = (sa[i] + sb[i]) * p.scale_log2;          // lane (query q, g): keys key0 + 4 g + i
    if (key0 + 16 > klen) {
#pragma unroll
      for (int i = 0; i < 4; ++i) sv[i] = key0 + 4 * g + i < klen ? sv[i] : -INFINITY;
    }
    // Reference of the exponentials: the maximum of the query's FIRST key group.  The accumulators (192 AGPRs) are never rescaled --
    // a multiply pulls them through VGPRs and the register allocator then keeps them there for the whole loop (256-956 B of scratch,
    // reloaded in every iteration) --: weights may grow to 2^64 (fp32 and bf16 have the exponent range for it), and when a query's
    // scores jump by more than that (44 nats) everything summed for it so far weighs < 1100 x 2^-64 of the new key, far below fp32
    // resolution: its reference moves up and its sums restart from zero.  The zeroing is per lane (EXEC-masked v_accvgpr_write of the
    // constant 0: no VGPR traffic); the asm barriers keep the compiler from turning the branch into selects, which spill again.
    // (An earlier version redid the whole stream on a jump: with a trained model's logits that restarted again and again -- a 256-step
    // generation went from 0.3 to 1.1 s.)
    const float mxl = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
    if (__builtin_amdgcn_ballot_w64(mxl > mref + 64.0f)) {                        // (-inf + 64 = -inf: always true for the first group)
      float mx = fmaxf(mxl, __shfl_xor(mxl, 16, 64));                             // per query: over the four key groups
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const bool jump = mx > mref + 64.0f;
      const bool had = mref != -INFINITY;
      mref = jump ? mx : mref;
      if (jump && had) {                      // divergent branch: only the lanes of the queries that jumped
        asm volatile("" ::: "memory");
        l = 0.f;
#pragma unroll
        for (int ct = 0; ct < 48; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("" ::: "memory");
      }
    }
    float pr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pr[i] = __builtin_amdgcn_exp2f(sv[i] - mref);
    l += (pr[0] + pr[1]) + (pr[2] + pr[3]);
    const uint2 pu = make_uint2(pack2bf(pr[0], pr[1]), pack2bf(pr[2], pr[3]));
    const s16x4 pb = __builtin_bit_cast(s16x4, pu);
    // groups of 8 column tiles, the next group's transposing reads requested before this group's MFMAs (one wave per SIMD: nothing else
    // hides the LDS latency)
    s16x4 ta[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ta[0][e] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((V2S_LDS s16x4*)(tbw + rb[e]));
#pragma unroll
    for (int gq = 0; gq < 6; ++gq) {
      if (gq < 5) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          ta[(gq + 1) & 1][e] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((V2S_LDS s16x4*)(tbw + rb[e] + (gq + 1) * 256));
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[gq * 8 + e] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ta[gq & 1][e], pb, acc[gq * 8 + e], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);      // 8 DS reads (the next group) ...
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);      // ... then 8 MFMAs (this group)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- merge the four streams: every wave publishes its sums as bf16 (the exponent range of fp32: no normalisation needed) in its
  // own staging buffer, then adds up its 192 output columns of all four with the streams' weights exp2(m_w - M) / L
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  __syncthreads();                                   // every wave is done with the query image
  float* sm = reinterpret_cast<float*>(smem);        // [4 waves][16 queries][2]
  if (g == 0) { sm[(wave * 16 + q) * 2] = mref; sm[(wave * 16 + q) * 2 + 1] = l; }
#pragma unroll
  for (int ct = 0; ct < 48; ++ct)                    // [48 column tiles][64 lanes] x 4 bf16
    *reinterpret_cast<uint2*>(tbw + (ct * 64 + lane) * 8) = make_uint2(pack2bf(acc[ct][0], acc[ct][1]), pack2bf(acc[ct][2], acc[ct][3]));
  __syncthreads();
  float M = -INFINITY;
#pragma unroll
  for (int w = 0; w < 4; ++w) M = fmaxf(M, sm[(w * 16 + q) * 2]);
  float sc[4], L = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float mw = sm[(w * 16 + q) * 2];
    sc[w] = mw == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw - M);      // a wave without a key group has weight 0
    L += sc[w] * sm[(w * 16 + q) * 2 + 1];
  }
  const float inv = 1.0f / L;
  float fw[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) fw[w] = sc[w] * inv;
  const long row = orow + q;
  if (wave == 0 && g == 0) { p.ml[row * 2] = M; p.ml[row * 2 + 1] = L; }
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    f32x2 oe = f32x2{0.f, 0.f}, oo = oe;               // even / odd elements (packed fp32 FMAs)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint2 u = *reinterpret_cast<const uint2*>(smem + QS_BYTES + w * TB_BYTES + ((wave * 12 + c) * 64 + lane) * 8);
      const f32x2 f = f32x2{fw[w], fw[w]};
      oe += f * f32x2{__uint_as_float(u.x << 16), __uint_as_float(u.y << 16)};
      oo += f * f32x2{__uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y & 0xffff0000u)};
    }
    const float o[4] = {oe[0], oo[0], oe[1], oo[1]};
    *reinterpret_cast<uint2*>(p.part + row * D + wave * 192 + c * 16 + g * 4) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
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
// l_s exp2(m_s - max) / sum: per lane, its row's).  The four waves split the contraction (192 columns each) and are summed through LDS.
// Everything a wave needs from memory is requested before anything is computed: its Wv fragments, the first four pieces' sums, the
// (max, weight sum) pairs -- the kernel is one round trip, not a chain of them.  Each wave derives the merge weights itself
// (wave-private LDS rows, no barrier).
template <int JT>          // 16-wide tiles of the head's 64 outputs per block (blockIdx.z walks the 4 / JT slices)
__global__ __launch_bounds__(256) void ctxfold_kernel(const CtxFoldP p) {
  __shared__ __attribute__((aligned(16))) float red[4][JT][64][4];
  __shared__ float wsh[4][16][65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane & 15, g = lane >> 4;
  const int h = blockIdx.x, rt = blockIdx.y;
  const int mrow = rt * 16 + q;
  const bool live = mrow < p.rows;
  const int ent = live ? mrow / p.G : 0, r = live ? (mrow % p.G) * p.H + h : 0;
  const int s0 = p.slot_off[ent], ns = p.slot_off[ent + 1] - s0;         // 1 <= ns <= 64
  const long prow0 = (long)s0 * p.qr + r;                                 // + s * qr
  const int j0 = blockIdx.z * (JT * 16);                                  // first of this block's outputs of the head
  const bf16_t* wr = p.wv + ((long)h * DH + j0 + q) * D + wave * 192 + g * 8;
  const bf16_t* pr = p.part + prow0 * D + wave * 192 + g * 8;
  const long pstep = (long)p.qr * D;
  uint4 wf[6][JT];
#pragma unroll
  for (int ks = 0; ks < 6; ++ks)
#pragma unroll
    for (int t = 0; t < JT; ++t) wf[ks][t] = *reinterpret_cast<const uint4*>(wr + (long)t * 16 * D + ks * 32);
  uint4 v0[4][6];                                     // the first four pieces (pieces past ns re-read the last one, weight 0)
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) v0[u][ks] = *reinterpret_cast<const uint4*>(pr + (u < ns ? u : ns - 1) * pstep + ks * 32);
  {                                                   // merge weights of the 16 rows: lanes (q, g) take pieces g, g + 4, ...
    float mmax = -INFINITY;
    for (int s = g; s < ns; s += 4) mmax = fmaxf(mmax, p.ml[(prow0 + (long)s * p.qr) * 2]);
    mmax = fmaxf(mmax, __shfl_xor(mmax, 16, 64));
    mmax = fmaxf(mmax, __shfl_xor(mmax, 32, 64));
    float wsum = 0.f;
    for (int s = g; s < ns; s += 4) {
      const float* e = p.ml + (prow0 + (long)s * p.qr) * 2;
      const float w = e[1] * __builtin_amdgcn_exp2f(e[0] - mmax);
      wsh[wave][q][s] = w;
      wsum += w;
    }
    wsum += __shfl_xor(wsum, 16, 64);
    wsum += __shfl_xor(wsum, 32, 64);
    if (g == 0) wsh[wave][q][64] = live ? 1.0f / wsum : 0.f;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): this wave's weights are in LDS
  const float winv = wsh[wave][q][64];
  float a8[6][8];
#pragma unroll
  for (int ks = 0; ks < 6; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) a8[ks][e] = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float w = u < ns ? wsh[wave][q][u] * winv : 0.f;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
      float f[8];
      unpack8(v0[u][ks], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) a8[ks][e] += w * f[e];
    }
  }
  int s = 4;
  for (; s + 4 <= ns; s += 4) {
    uint4 v[4][6];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int ks = 0; ks < 6; ++ks) v[u][ks] = *reinterpret_cast<const uint4*>(pr + (s + u) * pstep + ks * 32);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float w = wsh[wave][q][s + u] * winv;
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
    const float w = wsh[wave][q][s] * winv;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
      float f[8];
      unpack8(v[ks], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) a8[ks][e] += w * f[e];
    }
  }
  f32x4 acc[JT];
#pragma unroll
  for (int t = 0; t < JT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 6; ++ks) {
    const uint4 bu = pack8(a8[ks]);
    const bf16x8 b = __builtin_bit_cast(bf16x8, bu);
#pragma unroll
    for (int t = 0; t < JT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ks][t]), b, acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < JT; ++t) *reinterpret_cast<f32x4*>(&red[wave][t][lane][0]) = acc[t];
  __syncthreads();
  if (wave < JT) {                                    // wave t sums tile t of the four contraction quarters (fixed order)
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) o += *reinterpret_cast<const f32x4*>(&red[w][wave][lane][0]);
    if (live)
      *reinterpret_cast<uint2*>(p.ctx + (long)mrow * p.ld_ctx + h * DH + j0 + wave * 16 + 4 * g) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
  }
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

// Host-side plan of v2s_decode_memattn (no GPU work): cuts every entry's ceil(klen / 32) key tiles into ceil(tiles / tiles_per_piece)
// pieces of equal length (+-1 tile).  The cut of an entry depends on ITS length only -- never on the batch it is decoded in -- so a
// sequence's result is bit-identical whatever the batch composition (like the K/V-cache kernel's).  The engine uses 9 tiles (288
// keys) per piece: the reference's longest memory (100 frames + 1000 tokens = 35 tiles) is 4 pieces, so 64 entries fill the 256 CUs
// once (one block per CU: the kernel takes a whole CU's registers; a fifth piece per entry would mean a second, nearly empty round).  blk[i] = (entry, first tile | end tile << 16, slot,
// klen); slot_off[e] .. slot_off[e + 1] = the entry's slots.  Returns the number of blocks through *nblk (<= max_blocks, else an error).
extern "C" int v2s_decode_memattn_plan(const int32_t* klen_host, int32_t entries, int32_t tiles_per_piece, int32_t max_blocks, int32_t* blk,
                                       int32_t* slot_off, int32_t* nblk) {
  V2S_CHECK(klen_host && blk && slot_off && nblk && entries > 0 && tiles_per_piece > 0, V2S_ERR_ARG, "v2s_decode_memattn_plan: bad arguments");
  int b = 0;
  for (int e = 0; e < entries; ++e) {
    V2S_CHECK(klen_host[e] >= 1 && klen_host[e] <= 32 * 65535, V2S_ERR_SHAPE, "v2s_decode_memattn_plan: klen[%d] = %d (needs >= 1)", e, klen_host[e]);
    const int nt = (klen_host[e] + TK - 1) / TK, sp = (nt + tiles_per_piece - 1) / tiles_per_piece;
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
  p.qr = (R + 15) / 16 * 16;
  const size_t lds = (size_t)QS_BYTES + 4 * (size_t)TB_BYTES;           // 120 KiB
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)mem_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      v2s_set_error("v2s_decode_memattn: cannot raise the dynamic LDS limit to %zu", lds);
      return V2S_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(mem_attn_kernel, dim3(nblk, p.qr / 16), dim3(256), lds, (hipStream_t)stream, p);
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
  // 16 of the head's 64 outputs per block: 4 x 12 x row-tile blocks (192 at 64 rows) that each pull ~125 KB, instead of 48 that pull ~200 KB --
  // the kernel is bound by what one CU can load (8.8 -> 7.0 us at 64 rows; identical arithmetic either way)
  hipLaunchKernelGGL(ctxfold_kernel<1>, dim3(H, (rows + 15) / 16, 4), dim3(256), 0, (hipStream_t)stream, p);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}
