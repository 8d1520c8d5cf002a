// Fused (flash-style) attention for gfx950: forward, dQ and dK/dV kernels, head_dim = 64, bf16 I/O,
// fp32 softmax statistics, scores never written to HBM.
//
// Replaces model/vit.py:47-51 and model/modeling_t5.py:539-580 (incl. compute_bias :445-460 and the
// additive masks built at :996,1005,559) plus their autograd.
//
// Layout idea (all three kernels): the score tile is computed TRANSPOSED with respect to the operand that a
// wave owns, so that (a) the softmax statistics of a row live in one lane (+2 xor-shuffles), and (b) the
// accumulator registers of the first GEMM are *already* the MFMA operand registers of the second GEMM:
// the contraction index is simply enumerated in the order (16*(j>>2) + 4*(lane>>4) + (j&3)), which the
// other operand reproduces with two ds_read_b64_tr_b16 transpose reads from a row-major LDS tile.
// One LDS image (32-byte column chunks XOR-swizzled by (row>>1)&3) serves both the ds_read_b128 row
// fragments and the transpose reads without bank conflicts.
//
// head_dim 64 makes these kernels VALU-bound (few MFMA flops per score element), so the per-element work is
// specialised at compile time (bias / causal / dropout) and per tile at run time:
//   * "clean" tiles (no masked or out-of-range key, no causal edge) take a branch-free path;
//   * tiles whose keys are all masked (padding tail) or all in the causal future are skipped outright once
//     every row of the wave has seen a real key -- their probabilities are exactly 0 then, so the result is
//     bit-identical to processing them (rows that have seen no real key keep the reference's uniform
//     distribution semantics and are never skipped);
//   * the relative-position bias window is staged in LDS in four 1-float-shifted copies so that each lane
//     fetches its 4 consecutive diagonals with one aligned ds_read_b128.
#include <math.h>
#include "v2s_common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float MASKED2 = -3.0e38f;     // finite stand-in for finfo(float32).min in the log2 domain
constexpr float REAL_MIN = -1.0e37f;    // running max above this <=> the row has seen an unmasked key
constexpr int HD = 64;

struct AttnP {
  int B, H, Nq, Nk;
  const bf16_t *q, *k, *v;
  long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs;
  bf16_t* o; long o_bs, o_rs;
  float* ml;
  float scale;
  const float* bias_diag;
  const uint8_t* key_mask;
  int causal, causal_off;
  uint32_t p16; float inv_keep; uint32_t seed;
  const uint32_t* salt;   // device word XOR-ed into seed (v2s_set_seed_salt) or NULL
  const bf16_t* d_o; long do_bs, do_rs;
  const float* delta;
  bf16_t *dq, *dk, *dv;
  long dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs;
  float* dbias_diag;
  int far_lo, far_hi;   // all relative positions <= far_lo (>= far_hi) share one bias bucket
  const int* seq_off;   // packed self-attention: sequence b = rows [seq_off[b], seq_off[b+1]) of every operand (batch strides unused)
  int seq_q_only;       // seq_off applies to the query side only (q, o, d_o, dq); K / V / dK / dV stay dense [B][Nk] (cross-attention)
  const int* kv_seq_off; // K / V / dK / dV packed with their OWN row offsets (cross-attention over a padding-free memory); else see above
};

// byte offset of element (row, d) inside a [rows][64] bf16 LDS tile
__device__ __forceinline__ int tile_off(int row, int d) {
  return row * 128 + ((((d >> 4) ^ ((row >> 1) & 3))) << 5) + ((d & 15) << 1);
}

// row fragment (A or B operand with the contraction over d): lane -> row (lane&15), d = ks*32+(lane>>4)*8+j
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int rowbase, int ks, int lane) {
  const int row = rowbase + (lane & 15);
  const int d = ks * 32 + (lane >> 4) * 8;
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(tile + tile_off(row, d)));
}

// transposed fragment: lane -> column d = dbase+(lane&15); contraction rows enumerated as
// slot j of lane group g=(lane>>4):  row = rbase + 16*(j>>2) + 4*g + (j&3)
template <bool TR>
__device__ __forceinline__ bf16x8 col_frag(const char* tile, int rbase, int dbase, int lane) {
  const int g = lane >> 4, i = lane & 15;
  if (TR) {
    const int r0 = rbase + 4 * g + (i >> 2);
    const int d = dbase + (i & 3) * 4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((V2S_LDS s16x4*)(tile + tile_off(r0, d)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((V2S_LDS s16x4*)(tile + tile_off(r0 + 16, d)));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  } else {
    s16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = rbase + 16 * (j >> 2) + 4 * g + (j & 3);
      v[j] = *reinterpret_cast<const short*>(tile + tile_off(r, dbase + i));
    }
    return __builtin_bit_cast(bf16x8, v);
  }
}

// cooperative 64x64 tile load: thread -> (row = tid>>3 (+32), 16-byte chunk = tid&7)
__device__ __forceinline__ void tile_load(const bf16_t* base, long rs, int row0, int nrows, int tid, uint4 (&r)[2]) {
  const int chunk = tid & 7, rr = tid >> 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = row0 + rr + i * 32;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < nrows) v = *reinterpret_cast<const uint4*>(base + (long)row * rs + chunk * 8);
    r[i] = v;
  }
}
__device__ __forceinline__ void tile_store(char* tile, int tid, const uint4 (&r)[2]) {
  const int chunk = tid & 7, rr = tid >> 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(tile + tile_off(rr + i * 32, chunk * 8)) = r[i];
}

// two fp32 -> one packed bf16 pair with a single v_cvt_pk_bf16_f32 (element-wise casts compile to one conversion per element
// plus a v_perm to merge them)
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ uint32_t cvt_pk(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
// a + b on two fp32 lanes per instruction (the compiler splits a <2 x float> add that feeds scalar transcendental ops)
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ bf16x8 pack_frag(const f32x4& a, const f32x4& b) {
  const uint4 w = make_uint4(cvt_pk(a[0], a[1]), cvt_pk(a[2], a[3]), cvt_pk(b[0], b[1]), cvt_pk(b[2], b[3]));
  return __builtin_bit_cast(bf16x8, w);
}

// attention-probability dropout: keep(b,h,q,k) <=> mul24(rowseed(b,h,q) ^ keyhash(k), C2) >= (p16 << 16): a per-row seed (one
// full hash per row) and one xor-multiply per element, decided on the top 16 bits of the low product word.  The multiply is the
// 24-bit one (v_mul_u32_u24, full rate; v_mul_lo_u32 is quarter rate and was ~1/6 of the VALU time of a tile).  keyhash(k) =
// ((k & ~63) * C1) ^ ((k & 0xC) * C1) ^ ((k & 0x33) * C1): in the forward / dQ kernels a lane's 16 keys of a tile are
// k = k0 + 4g + (16 kb + r), so the 4g term is folded into the row seed once per kernel, the k0 term once per tile (scalar
// multiply), and the last term is an instruction literal -- one v_xor per element, no address arithmetic.  Same definition in
// all three kernels (forward mask == backward mask).
constexpr uint32_t DROP_C1 = 0x9E3779B1u, DROP_C2 = 0x00EBCA77u;
__device__ __forceinline__ uint32_t drop_rowseed(uint32_t seed, uint32_t rowid) { return v2s_hash32(seed ^ (rowid * 0x9E3779B1u)); }
__device__ __forceinline__ uint32_t drop_keyhash(uint32_t k) { return ((k & ~63u) * DROP_C1) ^ ((k & 0xCu) * DROP_C1) ^ ((k & 0x33u) * DROP_C1); }
__device__ __forceinline__ bool drop_keep(uint32_t rowseed, uint32_t kc1, uint32_t thr) {   // kc1 = (part of) keyhash(k), thr = p16 << 16
  return __umul24(rowseed ^ kc1, DROP_C2) >= thr;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32; x <= ~0 here

// ---- LDS stage layout ---------------------------------------------------------------------------------
constexpr int KV_TILE = 8192;                 // one [64][64] bf16 tile
constexpr int OFF_BIAS = 2 * KV_TILE;         // 4 x 192 floats: copy s holds w[i+s] at index i
constexpr int BIAS_COPY = 208;                // floats per copy: 192 used; 208 = 3 * 64 + 16 puts copy s 16 banks after copy s-1, so the
                                              // four copies that lanes with consecutive diagonals read (same 4-float group, different shift)
                                              // fall on different banks (at 192 they shared one 4-bank window: 4-way conflict on every read)
constexpr int OFF_FLAG = OFF_BIAS + 4 * BIAS_COPY * 4;   // 64 key flags (0 keep, 1 masked, 2 out of range)
constexpr int OFF_STATE = OFF_FLAG + 64;      // int[2]: OR of flags, AND of (flag != 0)
constexpr int OFF_MS = OFF_STATE + 16;        // dkv kernel: (m + log2 l)[64], masked-element exponent[64], delta[64], dropout row seed[64]
constexpr int STAGE = OFF_MS + 4 * 64 * 4;    // + 64 dropout row seeds (dkv kernel)
static_assert(STAGE % 16 == 0 && OFF_MS % 16 == 0 && OFF_BIAS % 16 == 0, "LDS carve alignment");

// bias window write: thread i (< 192) holds w[i]; copy s stores it at index i - s
__device__ __forceinline__ void bias_store(char* stage, int tid, float w) {
  float* b = reinterpret_cast<float*>(stage + OFF_BIAS);
#pragma unroll
  for (int s = 0; s < 4; ++s)
    if (tid - s >= 0) b[s * BIAS_COPY + tid - s] = w;
}
// aligned read of w[i0 .. i0+3] for any i0 >= 0
__device__ __forceinline__ float4 bias_read4(const char* stage, int i0) {
  const int s = i0 & 3;
  return *reinterpret_cast<const float4*>(stage + OFF_BIAS + (s * BIAS_COPY + (i0 - s)) * 4);
}

// flags of one tile -> (any, all) in LDS state words; executed by the first wave (tid < 64)
__device__ __forceinline__ void state_store(char* stage, int tid, uint32_t flag) {
  const unsigned long long any = __ballot(flag != 0);
  if (tid == 0) {
    int* st = reinterpret_cast<int*>(stage + OFF_STATE);
    st[0] = any != 0ull;
    st[1] = any == ~0ull;
  }
}

// ====================================================================================== forward
// block = 4 waves x 32 query rows; loop over 64-key tiles.  Three blocks per CU (<= 168 VGPRs; a dozen cold spills in the bias
// variants) instead of two: inside a wave the score MFMAs, the softmax VALU work and the PV MFMAs are serial, and only waves in
// different phases overlap them, so a third wave per SIMD is worth 8-13 % (encoder layer 308 -> 284 us, tools/attn_ab.py).  The
// causal variants spill into their hot path at that budget (28 -> 32 us) and stay at two.
template <bool TR, bool BIAS, bool CAUSAL, bool DROP>
__global__ __launch_bounds__(256, CAUSAL ? 2 : 3) void attn_fwd_kernel(const AttnP p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int nqb = (p.Nq + 127) >> 7;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int qblk = id % nqb, bh = id / nqb, h = bh % p.H, b = bh / p.H;
  const int Q0 = qblk * 128, wq0 = wave * 32;
  const int row0_ = p.seq_off ? p.seq_off[b] : 0;                          // packed (varlen) self-attention: first row of sequence b
  const int nq_ = p.seq_off ? p.seq_off[b + 1] - row0_ : p.Nq;
  const int* kso_ = p.kv_seq_off ? p.kv_seq_off : ((p.seq_off && !p.seq_q_only) ? p.seq_off : nullptr);     // row offsets of the key side
  const int krow0_ = kso_ ? kso_[b] : 0, nk_ = kso_ ? kso_[b + 1] - krow0_ : p.Nk;
  if (Q0 >= nq_) return;                                                   // block beyond the end of a short sequence

  const bf16_t* qp = p.q + (p.seq_off ? (long)row0_ * p.q_rs : (long)b * p.q_bs) + h * HD;
  const bf16_t* kp = p.k + (kso_ ? (long)krow0_ * p.k_rs : (long)b * p.k_bs) + h * HD;
  const bf16_t* vp = p.v + (kso_ ? (long)krow0_ * p.v_rs : (long)b * p.v_bs) + h * HD;

  bf16x8 qf[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int q = Q0 + wq0 + qb * 16 + li;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q < nq_) v = *reinterpret_cast<const uint4*>(qp + (long)q * p.q_rs + ks * 32 + g * 8);
      qf[qb][ks] = __builtin_bit_cast(bf16x8, v);
    }
  }
  float m[2] = {-INFINITY, -INFINITY}, lsum[2] = {0.f, 0.f};
  uint32_t rowseed[2] = {0u, 0u};
  if (DROP) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) rowseed[qb] = drop_rowseed(v2s_salted(p.seed, p.salt), (uint32_t)((b * p.H + h) * p.Nq + Q0 + wq0 + qb * 16 + li)) ^ ((uint32_t)(4 * g) * DROP_C1);
  }
  f32x4 ot[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int db = 0; db < 4; ++db) ot[qb][db] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ntiles = (nk_ + 63) >> 6;
  const float sc2 = p.scale * LOG2E;
  const int qmin = Q0 + wq0, qmax = qmin + 31;
  uint4 rk[2], rv[2];
  float rbias = 0.f;
  uint32_t rflag = 0;

  auto prefetch = [&](int t) {
    const int k0 = t * 64;
    tile_load(kp, p.k_rs, k0, nk_, tid, rk);
    tile_load(vp, p.v_rs, k0, nk_, tid, rv);
    if (BIAS && tid < 192) {
      const int idx = k0 - Q0 - 127 + tid + p.Nq - 1;
      rbias = (idx >= 0 && idx < p.Nq + p.Nk - 1) ? p.bias_diag[(long)h * (p.Nq + p.Nk - 1) + idx] * LOG2E : 0.f;
    }
    if (tid < 64) {
      const int k = k0 + tid;
      rflag = (k >= nk_) ? 2u : ((p.key_mask && p.key_mask[(long)b * p.Nk + k] == 0) ? 1u : 0u);
    }
  };
  auto commit = [&](int s) {
    char* st = smem + s * STAGE;
    tile_store(st, tid, rk);
    tile_store(st + KV_TILE, tid, rv);
    if (BIAS && tid < 192) bias_store(st, tid, rbias);
    if (tid < 64) {
      reinterpret_cast<uint8_t*>(st + OFF_FLAG)[tid] = (uint8_t)rflag;
      state_store(st, tid, rflag);
    }
  };

  prefetch(0);
  commit(0);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const char* sK = smem + (t & 1) * STAGE;
    const char* sV = sK + KV_TILE;
    const int k0 = t * 64;
    if (t + 1 < ntiles) prefetch(t + 1);

    const int* tstate = reinterpret_cast<const int*>(sK + OFF_STATE);
    const bool any_flag = tstate[0] != 0, all_flag = tstate[1] != 0;
    const bool future = CAUSAL && (k0 > qmax + p.causal_off);            // every element causally masked
    const bool edge = CAUSAL && (k0 + 63 > qmin + p.causal_off);         // some element causally masked
    const bool seen = __all((m[0] > REAL_MIN) && (m[1] > REAL_MIN));     // every row already saw a real key
    const bool skip = (all_flag || future) && seen;

    if (!skip) {
      f32x4 st[2][4];
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) st[qb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const bf16x8 kf = row_frag(sK, kb * 16, ks, lane);
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) st[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qb][ks], st[qb][kb], 0, 0, 0);
        }

      bf16x8 pf[2][2];
      const bool clean = !any_flag && !edge;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const int qq = wq0 + qb * 16 + li, q = Q0 + qq;
        float mx = -INFINITY;
        if (clean) {
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) {
            float4 bw = make_float4(0.f, 0.f, 0.f, 0.f);
            if (BIAS) bw = bias_read4(sK, kb * 16 + 4 * g + 127 - qq);
            st[qb][kb][0] = fmaf(st[qb][kb][0], sc2, bw.x);
            st[qb][kb][1] = fmaf(st[qb][kb][1], sc2, bw.y);
            st[qb][kb][2] = fmaf(st[qb][kb][2], sc2, bw.z);
            st[qb][kb][3] = fmaf(st[qb][kb][3], sc2, bw.w);
            mx = fmaxf(fmaxf(mx, fmaxf(st[qb][kb][0], st[qb][kb][1])), fmaxf(st[qb][kb][2], st[qb][kb][3]));
          }
        } else {
          const uint8_t* fl = reinterpret_cast<const uint8_t*>(sK + OFF_FLAG);
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) {
            const uint32_t f4 = *reinterpret_cast<const uint32_t*>(fl + kb * 16 + 4 * g);
            float4 bw = make_float4(0.f, 0.f, 0.f, 0.f);
            if (BIAS) bw = bias_read4(sK, kb * 16 + 4 * g + 127 - qq);
            const float bwv[4] = {bw.x, bw.y, bw.z, bw.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int kk = kb * 16 + 4 * g + r;
              float s = fmaf(st[qb][kb][r], sc2, bwv[r]);
              uint32_t f = (f4 >> (8 * r)) & 0xffu;
              if (CAUSAL && (k0 + kk) > q + p.causal_off) f |= 1u;
              s = (f & 2u) ? -INFINITY : ((f & 1u) ? MASKED2 : s);
              st[qb][kb][r] = s;
              mx = fmaxf(mx, s);
            }
          }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m[qb], mx);
        const float alpha = fast_exp2(m[qb] - mn);
        m[qb] = mn;
        f32x4 rs4 = f32x4{0.f, 0.f, 0.f, 0.f};     // subtraction and row sum as 4-vectors: v_pk_add_f32 handles two elements each
        const uint32_t rseed = rowseed[qb] ^ ((uint32_t)k0 * DROP_C1);
        const uint32_t thr = p.p16 << 16;
        const f32x2 nmn2 = f32x2{-mn, -mn};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const f32x2 x01 = pk_add(f32x2{st[qb][kb][0], st[qb][kb][1]}, nmn2), x23 = pk_add(f32x2{st[qb][kb][2], st[qb][kb][3]}, nmn2);
          const f32x4 pv4 = f32x4{fast_exp2(x01[0]), fast_exp2(x01[1]), fast_exp2(x23[0]), fast_exp2(x23[1])};
          rs4 += pv4;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            st[qb][kb][r] = (!DROP || drop_keep(rseed, (uint32_t)(kb * 16 + r) * DROP_C1, thr)) ? pv4[r] : 0.f;   // 1/(1-p) is applied once, to the output row
        }
        const float rs = (rs4[0] + rs4[1]) + (rs4[2] + rs4[3]);
        lsum[qb] = lsum[qb] * alpha + rs;
#pragma unroll
        for (int db = 0; db < 4; ++db) ot[qb][db] *= alpha;
        pf[qb][0] = pack_frag(st[qb][0], st[qb][1]);
        pf[qb][1] = pack_frag(st[qb][2], st[qb][3]);
      }
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const bf16x8 vf = col_frag<TR>(sV, kh * 32, db * 16, lane);
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) ot[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qb][kh], ot[qb][db], 0, 0, 0);
        }
    }
    if (t + 1 < ntiles) commit((t + 1) & 1);
    __syncthreads();
  }

#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int q = Q0 + wq0 + qb * 16 + li;
    float l = lsum[qb];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (q < nq_) {
      const float inv = (DROP ? p.inv_keep : 1.0f) / l;
      bf16_t* op = p.o + (p.seq_off ? (long)row0_ * p.o_rs : (long)b * p.o_bs) + (long)q * p.o_rs + h * HD;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        uint2 w;
        w.x = pack2bf(ot[qb][db][0] * inv, ot[qb][db][1] * inv);
        w.y = pack2bf(ot[qb][db][2] * inv, ot[qb][db][3] * inv);
        *reinterpret_cast<uint2*>(op + db * 16 + 4 * g) = w;
      }
      if (g == 0 && p.ml) {
        float* mp = p.ml + (((long)(b * p.H + h)) * p.Nq + q) * 2;
        mp[0] = m[qb];
        mp[1] = l;
      }
    }
  }
}

// ====================================================================================== delta = rowsum(dO * O)
__global__ __launch_bounds__(256) void attn_delta_kernel(const AttnP p, float* __restrict__ delta) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;   // one thread per 8 elements of a (b,q,h) row
  const long total = (long)p.B * p.Nq * p.H * 8;
  float s = 0.f;
  int b = 0, q = 0, h = 0;
  if (t < total) {
    const int c = (int)(t & 7);
    long r = t >> 3;
    h = (int)(r % p.H); r /= p.H;
    q = (int)(r % p.Nq); b = (int)(r / p.Nq);
    const int row0_ = p.seq_off ? p.seq_off[b] : 0;
    if (p.seq_off && q >= p.seq_off[b + 1] - row0_) q = -1;         // row beyond the end of a packed sequence: delta stays unwritten
  }
  if (t < total && q >= 0) {
    const int c = (int)(t & 7);
    const int row0_ = p.seq_off ? p.seq_off[b] : 0;
    float a[8], d[8];
    unpack8(*reinterpret_cast<const uint4*>(p.o + (p.seq_off ? (long)row0_ * p.o_rs : (long)b * p.o_bs) + (long)q * p.o_rs + h * HD + c * 8), a);
    unpack8(*reinterpret_cast<const uint4*>(p.d_o + (p.seq_off ? (long)row0_ * p.do_rs : (long)b * p.do_bs) + (long)q * p.do_rs + h * HD + c * 8), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j] * d[j];
  }
  s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
  if (t < total && q >= 0 && (t & 7) == 0) delta[((long)(b * p.H + h)) * p.Nq + q] = s;
}

// ====================================================================================== backward: dQ (+ dbias)
// same decomposition as the forward: wave = 32 query rows, loop over key tiles.
// Tiles that are fully masked / fully in the causal future contribute exactly zero to dQ and dbias whenever
// the row statistics come from a real key (m > REAL_MIN), and are skipped under that condition.
template <bool TR, bool BIAS, bool CAUSAL, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2*STAGE + (Nk+128)*8 (dbias window) bytes
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // scalar: the per-block bias-gradient routing below branches on it
  const int nqb = (p.Nq + 127) >> 7;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int qblk = id % nqb, bh = id / nqb, h = bh % p.H, b = bh / p.H;
  const int Q0 = qblk * 128, wq0 = wave * 32;
  const int row0_ = p.seq_off ? p.seq_off[b] : 0;                          // packed (varlen) self-attention: first row of sequence b
  const int nq_ = p.seq_off ? p.seq_off[b + 1] - row0_ : p.Nq;
  const int* kso_ = p.kv_seq_off ? p.kv_seq_off : ((p.seq_off && !p.seq_q_only) ? p.seq_off : nullptr);     // row offsets of the key side
  const int krow0_ = kso_ ? kso_[b] : 0, nk_ = kso_ ? kso_[b + 1] - krow0_ : p.Nk;
  if (Q0 >= nq_) return;                                                   // block beyond the end of a short sequence
  const bool want_dbias = BIAS && p.dbias_diag != nullptr;
  // per-diagonal bias-gradient window of this block, index (k - q) + (Q0 + 127) in [0, Nk+127), accumulated in 64-bit
  // fixed point (2^-40 units): LDS float atomics run at ~0.33 lane-ops/clk/CU on gfx950, 64-bit integer ones at ~8
  // (tools/ubench/lds_atomic.hip) -- and the integer sum is exact and order-independent.
  unsigned long long* dbw = reinterpret_cast<unsigned long long*>(smem + 2 * STAGE);
  const int ndb = p.Nk + 127;
  if (want_dbias) {
    for (int i = tid; i < ndb; i += 256) dbw[i] = 0ull;
  }
  float acc_lo = 0.f, acc_hi = 0.f;    // bias-gradient mass of the two "far" buckets (no per-diagonal resolution needed)

  const bf16_t* qp = p.q + (p.seq_off ? (long)row0_ * p.q_rs : (long)b * p.q_bs) + h * HD;
  const bf16_t* dop = p.d_o + (p.seq_off ? (long)row0_ * p.do_rs : (long)b * p.do_bs) + h * HD;
  const bf16_t* kp = p.k + (kso_ ? (long)krow0_ * p.k_rs : (long)b * p.k_bs) + h * HD;
  const bf16_t* vp = p.v + (kso_ ? (long)krow0_ * p.v_rs : (long)b * p.v_bs) + h * HD;

  bf16x8 qf[2][2], dof[2][2];
  float m2[2], xmask[2], dl[2];
  uint32_t rowseed[2] = {0u, 0u};
  bool rows_real = true;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int q = Q0 + wq0 + qb * 16 + li;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0), w = make_uint4(0, 0, 0, 0);
      if (q < nq_) {
        v = *reinterpret_cast<const uint4*>(qp + (long)q * p.q_rs + ks * 32 + g * 8);
        w = *reinterpret_cast<const uint4*>(dop + (long)q * p.do_rs + ks * 32 + g * 8);
      }
      qf[qb][ks] = __builtin_bit_cast(bf16x8, v);
      dof[qb][ks] = __builtin_bit_cast(bf16x8, w);
    }
    // 1/l is folded into the exponent: P = exp2(s - (m + log2 l)).  Rows >= Nq: huge offset => P = 0 => dS = 0.  A row whose
    // keys are ALL masked (m <= REAL_MIN) keeps the reference's uniform distribution: its masked elements evaluate to
    // exp2(-log2 l) = 1/l (xmask), every other row's masked elements to 0.
    m2[qb] = 1.0e30f; xmask[qb] = -3.0e38f; dl[qb] = 0.f;
    bool real = true;
    if (q < nq_) {
      const long r = ((long)(b * p.H + h)) * p.Nq + q;
      const float mm = p.ml[r * 2], l2 = __log2f(p.ml[r * 2 + 1]);
      real = mm > REAL_MIN;
      m2[qb] = real ? mm + l2 : 0.f;
      xmask[qb] = real ? -3.0e38f : -l2;
      dl[qb] = p.delta[r];
    }
    rows_real = rows_real && real;
    if (DROP) rowseed[qb] = drop_rowseed(v2s_salted(p.seed, p.salt), (uint32_t)((b * p.H + h) * p.Nq + q)) ^ ((uint32_t)(4 * g) * DROP_C1);
  }
  const bool seen = __all(rows_real);
  const float isc2 = 1.0f / (p.scale * LOG2E);
  const float ninit[2] = {-m2[0] * isc2, -m2[1] * isc2};
  f32x4 dqt[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int db = 0; db < 4; ++db) dqt[qb][db] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ntiles = (nk_ + 63) >> 6;
  const float sc2 = p.scale * LOG2E;
  const int qmin = Q0 + wq0, qmax = qmin + 31;
  uint4 rk[2], rv[2];
  float rbias = 0.f;
  uint32_t rflag = 0;
  auto prefetch = [&](int t) {
    const int k0 = t * 64;
    tile_load(kp, p.k_rs, k0, nk_, tid, rk);
    tile_load(vp, p.v_rs, k0, nk_, tid, rv);
    if (BIAS && tid < 192) {
      const int idx = k0 - Q0 - 127 + tid + p.Nq - 1;
      rbias = (idx >= 0 && idx < p.Nq + p.Nk - 1) ? p.bias_diag[(long)h * (p.Nq + p.Nk - 1) + idx] * LOG2E : 0.f;
    }
    if (tid < 64) {
      const int k = k0 + tid;
      rflag = (k >= nk_) ? 2u : ((p.key_mask && p.key_mask[(long)b * p.Nk + k] == 0) ? 1u : 0u);
    }
  };
  auto commit = [&](int s) {
    char* st = smem + s * STAGE;
    tile_store(st, tid, rk);
    tile_store(st + KV_TILE, tid, rv);
    if (BIAS && tid < 192) bias_store(st, tid, rbias);
    if (tid < 64) {
      reinterpret_cast<uint8_t*>(st + OFF_FLAG)[tid] = (uint8_t)rflag;
      state_store(st, tid, rflag);
    }
  };
  prefetch(0);
  commit(0);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const char* sK = smem + (t & 1) * STAGE;
    const char* sV = sK + KV_TILE;
    const int k0 = t * 64;
    if (t + 1 < ntiles) prefetch(t + 1);

    const int* tstate = reinterpret_cast<const int*>(sK + OFF_STATE);
    const bool any_flag = tstate[0] != 0, all_flag = tstate[1] != 0;
    const bool future = CAUSAL && (k0 > qmax + p.causal_off);
    const bool edge = CAUSAL && (k0 + 63 > qmin + p.causal_off);
    const bool skip = (all_flag || future) && seen;

    if (!skip) {
      // the score accumulators start at -m/sc2 so that fma(acc, sc2, bias) is already (s - m) in the log2 domain
      f32x4 st[2][4], dp[2][4];
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) { st[qb][kb] = f32x4{ninit[qb], ninit[qb], ninit[qb], ninit[qb]}; dp[qb][kb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const bf16x8 kf = row_frag(sK, kb * 16, ks, lane);
          const bf16x8 vf = row_frag(sV, kb * 16, ks, lane);
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) {
            st[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qb][ks], st[qb][kb], 0, 0, 0);
            dp[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[qb][ks], dp[qb][kb], 0, 0, 0);
          }
        }
      const bool clean = !any_flag && !edge;
      // bias-gradient routing per 16x16 block (qb, kb): relative positions d = k - q span a 31-wide range; blocks entirely in
      // a far bucket just sum their dS (1: far-low, 2: far-high), only the near-diagonal blocks (3) resolve diagonals
      bf16x8 dsf[2][2];
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const int qq = wq0 + qb * 16 + li, q = Q0 + qq;
        const uint8_t* fl = reinterpret_cast<const uint8_t*>(sK + OFF_FLAG);
        const uint32_t rseed = rowseed[qb] ^ ((uint32_t)k0 * DROP_C1);
        const uint32_t thr = p.p16 << 16;
        const float dl_q = dl[qb], ndl_q = -dl_q, xm_q = xmask[qb];
        const int qlo = Q0 + wq0 + qb * 16;            // rows of this block: qlo .. qlo+15
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const int klo = k0 + kb * 16;
          const int route = !want_dbias ? 0 : ((klo + 15 - qlo) <= p.far_lo ? 1 : ((klo - (qlo + 15)) >= p.far_hi ? 2 : 3));
          float4 bw = make_float4(0.f, 0.f, 0.f, 0.f);
          if (BIAS) bw = bias_read4(sK, kb * 16 + 4 * g + 127 - qq);
          const float bwv[4] = {bw.x, bw.y, bw.z, bw.w};
          float x[4];
          if (clean) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = fmaf(st[qb][kb][r], sc2, bwv[r]);
          } else {
            const uint32_t f4 = *reinterpret_cast<const uint32_t*>(fl + kb * 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              uint32_t f = (f4 >> (8 * r)) & 0xffu;
              if (CAUSAL && (k0 + kb * 16 + 4 * g + r) > q + p.causal_off) f |= 1u;
              x[r] = (f & 2u) ? -INFINITY : ((f & 1u) ? xm_q : fmaf(st[qb][kb][r], sc2, bwv[r]));
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = fast_exp2(x[r]);                       // already divided by l
            float u = fmaf(dp[qb][kb][r], DROP ? p.inv_keep : 1.0f, ndl_q);      // dP/(1-p) - delta on kept elements
            if (DROP) u = drop_keep(rseed, (uint32_t)(kb * 16 + r) * DROP_C1, thr) ? u : ndl_q;
            st[qb][kb][r] = pr * u;
          }
          if (BIAS && route != 0) {      // route is wave-uniform: one scalar branch per 16x16 block, none per element
            if (route == 3) {            // near-diagonal block: every element goes to its own diagonal of the window
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                // float -> 2^-40 fixed point without the (emulated, ~11 instruction) float->int64 conversion: adding
                // 1.5 * 2^12 in double leaves round(ds * 2^40) (two's complement, |ds| < 2048) in the low 51 mantissa bits;
                // the magic number's bit pattern has a zero low word, so subtracting its high word yields the integer.
                const double md = (double)st[qb][kb][r] + 6144.0;
                const unsigned long long bits = __builtin_bit_cast(unsigned long long, md) - 0x40B8000000000000ull;
                atomicAdd(&dbw[(k0 + kb * 16 + 4 * g + r) + 127 - qq], bits);
              }
            } else {
              const float lsum_ds = (st[qb][kb][0] + st[qb][kb][1]) + (st[qb][kb][2] + st[qb][kb][3]);
              if (route == 1) acc_lo += lsum_ds;
              else acc_hi += lsum_ds;
            }
          }
        }
        dsf[qb][0] = pack_frag(st[qb][0], st[qb][1]);
        dsf[qb][1] = pack_frag(st[qb][2], st[qb][3]);
      }
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const bf16x8 ktf = col_frag<TR>(sK, kh * 32, db * 16, lane);
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) dqt[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf[qb][kh], dqt[qb][db], 0, 0, 0);
        }
    }
    if (t + 1 < ntiles) commit((t + 1) & 1);
    __syncthreads();
  }

#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int q = Q0 + wq0 + qb * 16 + li;
    if (q < nq_) {
      bf16_t* op = p.dq + (p.seq_off ? (long)row0_ * p.dq_rs : (long)b * p.dq_bs) + (long)q * p.dq_rs + h * HD;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        uint2 w;
        w.x = pack2bf(dqt[qb][db][0] * p.scale, dqt[qb][db][1] * p.scale);
        w.y = pack2bf(dqt[qb][db][2] * p.scale, dqt[qb][db][3] * p.scale);
        *reinterpret_cast<uint2*>(op + db * 16 + 4 * g) = w;
      }
    }
  }
  if (want_dbias) {
    // dbw index i <-> (k - q) = i - (Q0 + 127);  global index = (k - q) + Nq - 1.  The far-bucket masses go to the
    // diagonals far_lo / far_hi themselves (same bucket): dbias_diag is meaningful after bucket reduction.
    float* dst = p.dbias_diag + (long)h * (p.Nq + p.Nk - 1);
    acc_lo = wave_sum(acc_lo);
    acc_hi = wave_sum(acc_hi);
    if (lane == 0) {
      const int glo = p.far_lo + p.Nq - 1, ghi = p.far_hi + p.Nq - 1;
      if (acc_lo != 0.f && glo >= 0 && glo < p.Nq + p.Nk - 1) atomicAdd(dst + glo, acc_lo);
      if (acc_hi != 0.f && ghi >= 0 && ghi < p.Nq + p.Nk - 1) atomicAdd(dst + ghi, acc_hi);
    }
    for (int i = tid; i < ndb; i += 256) {
      const int gi = i - (Q0 + 127) + p.Nq - 1;
      const float v = (float)((double)(long long)dbw[i] * (1.0 / 1099511627776.0));
      if (gi >= 0 && gi < p.Nq + p.Nk - 1 && v != 0.f) atomicAdd(dst + gi, v);
    }
  }
}

// ====================================================================================== backward: dK, dV
// wave = 32 keys (2 key blocks), block = 128 keys, loop over 64-query tiles (Q and dO tiles in LDS).
template <bool TR, bool BIAS, bool CAUSAL, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const AttnP p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int nkb = (p.Nk + 127) >> 7;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int kblk = id % nkb, bh = id / nkb, h = bh % p.H, b = bh / p.H;
  const int K0 = kblk * 128, wk0 = wave * 32;
  const int row0_ = p.seq_off ? p.seq_off[b] : 0;
  const int nq_ = p.seq_off ? p.seq_off[b + 1] - row0_ : p.Nq;
  const int* kso_ = p.kv_seq_off ? p.kv_seq_off : ((p.seq_off && !p.seq_q_only) ? p.seq_off : nullptr);     // row offsets of the key side
  const int krow0_ = kso_ ? kso_[b] : 0, nk_ = kso_ ? kso_[b + 1] - krow0_ : p.Nk;
  if (K0 >= nk_) return;

  const bf16_t* qp = p.q + (p.seq_off ? (long)row0_ * p.q_rs : (long)b * p.q_bs) + h * HD;
  const bf16_t* dop = p.d_o + (p.seq_off ? (long)row0_ * p.do_rs : (long)b * p.do_bs) + h * HD;
  const bf16_t* kp = p.k + (kso_ ? (long)krow0_ * p.k_rs : (long)b * p.k_bs) + h * HD;
  const bf16_t* vp = p.v + (kso_ ? (long)krow0_ * p.v_rs : (long)b * p.v_bs) + h * HD;

  bf16x8 kf[2][2], vf[2][2];
  uint32_t kflag[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int k = K0 + wk0 + kb * 16 + li;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 a = make_uint4(0, 0, 0, 0), c = make_uint4(0, 0, 0, 0);
      if (k < nk_) {
        a = *reinterpret_cast<const uint4*>(kp + (long)k * p.k_rs + ks * 32 + g * 8);
        c = *reinterpret_cast<const uint4*>(vp + (long)k * p.v_rs + ks * 32 + g * 8);
      }
      kf[kb][ks] = __builtin_bit_cast(bf16x8, a);
      vf[kb][ks] = __builtin_bit_cast(bf16x8, c);
    }
    kflag[kb] = (k >= nk_) ? 2u : ((p.key_mask && p.key_mask[(long)b * p.Nk + k] == 0) ? 1u : 0u);
  }
  const bool keys_clean = __all(kflag[0] == 0u && kflag[1] == 0u);
  const int kmin = K0 + wk0, kmax = kmin + 31;
  f32x4 dkt[2][4], dvt[2][4];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int db = 0; db < 4; ++db) { dkt[kb][db] = f32x4{0.f, 0.f, 0.f, 0.f}; dvt[kb][db] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int ntiles = (nq_ + 63) >> 6;
  const float sc2 = p.scale * LOG2E, isc2 = 1.0f / sc2;
  const uint32_t thr = p.p16 << 16;
  uint4 rq[2], rdo[2];
  float rbias = 0.f;
  float rm = 0.f, rl = 0.f, rd = 0.f;
  int rreal = 1;
  uint32_t rseed = 0;
  // bias window for this (128-key block, 64-query tile): index (k - q) - dmin, dmin = K0 - (q0 + 63); 191 entries
  auto prefetch = [&](int t) {
    const int q0 = t * 64;
    tile_load(qp, p.q_rs, q0, nq_, tid, rq);
    tile_load(dop, p.do_rs, q0, nq_, tid, rdo);
    if (BIAS && tid < 192) {
      const int idx = K0 - q0 - 63 + tid + p.Nq - 1;
      rbias = (idx >= 0 && idx < p.Nq + p.Nk - 1) ? p.bias_diag[(long)h * (p.Nq + p.Nk - 1) + idx] * LOG2E : 0.f;
    }
    if (tid < 64) {
      const int q = q0 + tid;
      if (DROP) rseed = drop_rowseed(v2s_salted(p.seed, p.salt), (uint32_t)((b * p.H + h) * p.Nq + q));
      // rm = m + log2 l (1/l folded into the exponent), rl = value of a MASKED element's exponent: -log2 l for a row whose keys
      // are all masked (uniform distribution, like the reference), -huge otherwise; rows >= Nq: P = 0.  See the dQ kernel.
      rm = 1.0e30f; rl = -3.0e38f; rd = 0.f; rreal = 1;
      if (q < nq_) {
        const long r = ((long)(b * p.H + h)) * p.Nq + q;
        const float mm = p.ml[r * 2], l2 = __log2f(p.ml[r * 2 + 1]);
        rreal = mm > REAL_MIN;
        rm = rreal ? mm + l2 : 0.f; rl = rreal ? -3.0e38f : -l2; rd = p.delta[r];
      }
    }
  };
  auto commit = [&](int s) {
    char* st = smem + s * STAGE;
    tile_store(st, tid, rq);
    tile_store(st + KV_TILE, tid, rdo);
    if (BIAS && tid < 192) bias_store(st, tid, rbias);
    if (tid < 64) {
      float* ms = reinterpret_cast<float*>(st + OFF_MS);
      ms[tid] = rm; ms[64 + tid] = rl; ms[128 + tid] = rd;
      if (DROP) reinterpret_cast<uint32_t*>(ms)[192 + tid] = rseed;
      // "every query row of this tile has real statistics": lets all-masked / all-future key blocks skip the tile
      const unsigned long long real = __ballot(rreal != 0);
      if (tid == 0) reinterpret_cast<int*>(st + OFF_STATE)[0] = (real == ~0ull);
    }
  };
  prefetch(0);
  commit(0);
  __syncthreads();

  const bool keys_all_masked = __all(kflag[0] != 0u && kflag[1] != 0u);
  for (int t = 0; t < ntiles; ++t) {
    const char* sQ = smem + (t & 1) * STAGE;
    const char* sDO = sQ + KV_TILE;
    const float* ms = reinterpret_cast<const float*>(sQ + OFF_MS);
    const int q0 = t * 64;
    if (t + 1 < ntiles) prefetch(t + 1);

    const bool rows_real = reinterpret_cast<const int*>(sQ + OFF_STATE)[0] != 0;
    const bool future = CAUSAL && (kmin > q0 + 63 + p.causal_off);     // every (q, k) pair of this tile is causally masked
    const bool edge = CAUSAL && (kmax > q0 + p.causal_off);
    const bool skip = (keys_all_masked || future) && rows_real;

    if (!skip) {
      const bool clean = keys_clean && !edge && (q0 + 63 < nq_);
      // two halves of 32 query rows each (keeps the live score registers at 2x2 fragments)
#pragma unroll
      for (int qh = 0; qh < 2; ++qh) {
        // S[q][key] and dP[q][key]:  D[row = q = qb*16 + 4g + r][col = key = kb*16 + li]
        // score accumulators start at -m[q]/sc2 (row q = 4g + r), see the dQ kernel
        f32x4 st[2][2], dp[2][2];
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
          const float4 mi4 = *reinterpret_cast<const float4*>(ms + (2 * qh + qi) * 16 + 4 * g);
          const f32x4 init = f32x4{-mi4.x * isc2, -mi4.y * isc2, -mi4.z * isc2, -mi4.w * isc2};
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) { st[qi][kb] = init; dp[qi][kb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int qi = 0; qi < 2; ++qi) {
            const bf16x8 qfr = row_frag(sQ, (2 * qh + qi) * 16, ks, lane);
            const bf16x8 dfr = row_frag(sDO, (2 * qh + qi) * 16, ks, lane);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
              st[qi][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[kb][ks], st[qi][kb], 0, 0, 0);
              dp[qi][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfr, vf[kb][ks], dp[qi][kb], 0, 0, 0);
            }
          }
        // P (dropped) -> dp registers become Pd ; st registers become dS
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
          const int qb = 2 * qh + qi;
          const float4 lv = *reinterpret_cast<const float4*>(ms + 64 + qb * 16 + 4 * g);
          const float4 dv4 = *reinterpret_cast<const float4*>(ms + 128 + qb * 16 + 4 * g);
          const float lr[4] = {lv.x, lv.y, lv.z, lv.w}, dr[4] = {dv4.x, dv4.y, dv4.z, dv4.w};
          uint4 sd4 = make_uint4(0, 0, 0, 0);
          if (DROP) sd4 = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint32_t*>(ms) + 192 + qb * 16 + 4 * g);
          const uint32_t sdr[4] = {sd4.x, sd4.y, sd4.z, sd4.w};
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            const int kk = wk0 + kb * 16 + li, k = K0 + kk;
            const uint32_t kc = drop_keyhash((uint32_t)k);
            // window entries for r = 0..3 sit at decreasing indices i0 - r with i0 = kk + 63 - (qb*16 + 4g)
            float bwv[4] = {0.f, 0.f, 0.f, 0.f};
            if (BIAS) {
              const float4 bw = bias_read4(sQ, kk + 63 - (qb * 16 + 4 * g) - 3);
              bwv[0] = bw.w; bwv[1] = bw.z; bwv[2] = bw.y; bwv[3] = bw.x;
            }
            float x[4];
            if (clean) {
#pragma unroll
              for (int r = 0; r < 4; ++r) x[r] = fmaf(st[qi][kb][r], sc2, bwv[r]);
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int q = q0 + qb * 16 + 4 * g + r;
                uint32_t f = kflag[kb];
                if (CAUSAL && k > q + p.causal_off) f |= 1u;
                x[r] = (f & 2u) ? -INFINITY : ((f & 1u) ? lr[r] : fmaf(st[qi][kb][r], sc2, bwv[r]));
              }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pr = fast_exp2(x[r]);                    // already divided by l
              float pd = pr;
              if (DROP) pd = drop_keep(sdr[r], kc, thr) ? pr * p.inv_keep : 0.f;
              st[qi][kb][r] = fmaf(pd, dp[qi][kb][r], -(pr * dr[r]));   // dS = P * (dP_dropped - delta)
              dp[qi][kb][r] = pd;
            }
          }
        }
        // dV^T[d][key] += dO^T[d][q] * Pd[q][key] ; dK^T[d][key] += Q^T[d][q] * dS[q][key]
        bf16x8 pdf[2], dsf[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          pdf[kb] = pack_frag(dp[0][kb], dp[1][kb]);
          dsf[kb] = pack_frag(st[0][kb], st[1][kb]);
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const bf16x8 dot = col_frag<TR>(sDO, qh * 32, db * 16, lane);
          const bf16x8 qt = col_frag<TR>(sQ, qh * 32, db * 16, lane);
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            dvt[kb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pdf[kb], dvt[kb][db], 0, 0, 0);
            dkt[kb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsf[kb], dkt[kb][db], 0, 0, 0);
          }
        }
      }
    }
    if (t + 1 < ntiles) commit((t + 1) & 1);
    __syncthreads();
  }

#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int k = K0 + wk0 + kb * 16 + li;
    if (k < nk_) {
      bf16_t* dkp = p.dk + (kso_ ? (long)krow0_ * p.dk_rs : (long)b * p.dk_bs) + (long)k * p.dk_rs + h * HD;
      bf16_t* dvp = p.dv + (kso_ ? (long)krow0_ * p.dv_rs : (long)b * p.dv_bs) + (long)k * p.dv_rs + h * HD;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        uint2 w;
        w.x = pack2bf(dkt[kb][db][0] * p.scale, dkt[kb][db][1] * p.scale);
        w.y = pack2bf(dkt[kb][db][2] * p.scale, dkt[kb][db][3] * p.scale);
        *reinterpret_cast<uint2*>(dkp + db * 16 + 4 * g) = w;
        uint2 u;
        u.x = pack2bf(dvt[kb][db][0], dvt[kb][db][1]);
        u.y = pack2bf(dvt[kb][db][2], dvt[kb][db][3]);
        *reinterpret_cast<uint2*>(dvp + db * 16 + 4 * g) = u;
      }
    }
  }
}

// ---------------------------------------------------------------------------- bias table <-> diagonal
__global__ void bias_diag_fwd_kernel(const float* __restrict__ table, const int* __restrict__ lut, float* __restrict__ out,
                                     int H, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * H) return;
  const int h = i / n, d = i - h * n;
  out[i] = table[lut[d] * H + h];
}
// one block per (bucket, head): deterministic reduction over the diagonals mapping to that bucket
__global__ __launch_bounds__(256) void bias_bucket_bwd_kernel(const float* __restrict__ dd, const int* __restrict__ lut,
                                                              float* __restrict__ dtable, int H, int n) {
  __shared__ float red[256];
  const int bucket = blockIdx.x, h = blockIdx.y;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256)
    if (lut[i] == bucket) s += dd[(long)h * n + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) dtable[bucket * H + h] += red[0];
}

int fill(AttnP& p, const v2s_attn_args* a, const char* who, bool bwd) {
  V2S_CHECK(a != nullptr, V2S_ERR_ARG, "%s: null args", who);
  V2S_CHECK(a->B > 0 && a->H > 0 && a->Nq > 0 && a->Nk > 0, V2S_ERR_SHAPE, "%s: bad shape B=%d H=%d Nq=%d Nk=%d", who, a->B, a->H, a->Nq, a->Nk);
  V2S_CHECK(((a->q_rs | a->k_rs | a->v_rs | a->o_rs | a->q_bs | a->k_bs | a->v_bs | a->o_bs) % 8) == 0, V2S_ERR_ALIGN, "%s: strides must be multiples of 8 elements", who);
  V2S_CHECK((((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->o) & 15) == 0, V2S_ERR_ALIGN, "%s: pointers must be 16-byte aligned", who);
  V2S_CHECK(a->dropout_p >= 0.f && a->dropout_p < 1.f, V2S_ERR_ARG, "%s: dropout_p out of range", who);
  p.B = a->B; p.H = a->H; p.Nq = a->Nq; p.Nk = a->Nk;
  p.q = (const bf16_t*)a->q; p.k = (const bf16_t*)a->k; p.v = (const bf16_t*)a->v;
  p.q_bs = a->q_bs; p.q_rs = a->q_rs; p.k_bs = a->k_bs; p.k_rs = a->k_rs; p.v_bs = a->v_bs; p.v_rs = a->v_rs;
  p.o = (bf16_t*)a->o; p.o_bs = a->o_bs; p.o_rs = a->o_rs;
  p.ml = a->ml; p.scale = a->scale; p.bias_diag = a->bias_diag; p.key_mask = a->key_mask;
  p.causal = a->causal; p.causal_off = a->causal_off;
  p.p16 = (uint32_t)(a->dropout_p * 65536.0f + 0.5f);
  p.inv_keep = p.p16 ? 1.0f / (1.0f - (float)p.p16 / 65536.0f) : 1.0f;
  p.seed = a->dropout_seed;
  p.salt = v2s_seed_salt();
  p.d_o = (const bf16_t*)a->d_o; p.do_bs = a->do_bs; p.do_rs = a->do_rs; p.delta = a->delta;
  p.dq = (bf16_t*)a->dq; p.dk = (bf16_t*)a->dk; p.dv = (bf16_t*)a->dv;
  p.dq_bs = a->dq_bs; p.dq_rs = a->dq_rs; p.dk_bs = a->dk_bs; p.dk_rs = a->dk_rs; p.dv_bs = a->dv_bs; p.dv_rs = a->dv_rs;
  p.dbias_diag = a->dbias_diag;
  p.seq_off = a->seq_off; p.seq_q_only = (a->seq_off && a->seq_q_only) ? 1 : 0; p.kv_seq_off = a->kv_seq_off;
  V2S_CHECK(!a->seq_off || a->seq_q_only || a->kv_seq_off || (a->Nq == a->Nk && !a->key_mask), V2S_ERR_ARG, "%s: seq_off (packed self-attention) needs Nq == Nk and no key_mask", who);
  V2S_CHECK(!a->kv_seq_off || !a->key_mask, V2S_ERR_ARG, "%s: kv_seq_off (packed keys) excludes key_mask: pad keys simply do not exist", who);
  // far buckets: disabled (every diagonal resolved) unless the caller states lo < hi
  if (a->bias_far_lo < a->bias_far_hi) { p.far_lo = a->bias_far_lo; p.far_hi = a->bias_far_hi; }
  else { p.far_lo = -(1 << 30); p.far_hi = (1 << 30); }
  if (bwd) {
    V2S_CHECK(a->d_o && a->ml && a->dq && a->dk && a->dv, V2S_ERR_ARG, "%s: backward needs d_o, ml, dq, dk, dv", who);
    V2S_CHECK(((a->do_rs | a->do_bs | a->dq_rs | a->dk_rs | a->dv_rs | a->dq_bs | a->dk_bs | a->dv_bs) % 8) == 0, V2S_ERR_ALIGN, "%s: grad strides must be multiples of 8", who);
  }
  return V2S_OK;
}

// compile-time specialisation dispatch: (tr_read, bias, causal, dropout)
#define V2S_DISPATCH4(KERNEL, tr, bias, causal, drop, ...)                                                    \
  do {                                                                                                        \
    const int key__ = ((tr) ? 8 : 0) | ((bias) ? 4 : 0) | ((causal) ? 2 : 0) | ((drop) ? 1 : 0);              \
    switch (key__) {                                                                                          \
      case 0: hipLaunchKernelGGL((KERNEL<false, false, false, false>), __VA_ARGS__); break;                   \
      case 1: hipLaunchKernelGGL((KERNEL<false, false, false, true>), __VA_ARGS__); break;                    \
      case 2: hipLaunchKernelGGL((KERNEL<false, false, true, false>), __VA_ARGS__); break;                    \
      case 3: hipLaunchKernelGGL((KERNEL<false, false, true, true>), __VA_ARGS__); break;                     \
      case 4: hipLaunchKernelGGL((KERNEL<false, true, false, false>), __VA_ARGS__); break;                    \
      case 5: hipLaunchKernelGGL((KERNEL<false, true, false, true>), __VA_ARGS__); break;                     \
      case 6: hipLaunchKernelGGL((KERNEL<false, true, true, false>), __VA_ARGS__); break;                     \
      case 7: hipLaunchKernelGGL((KERNEL<false, true, true, true>), __VA_ARGS__); break;                      \
      case 8: hipLaunchKernelGGL((KERNEL<true, false, false, false>), __VA_ARGS__); break;                    \
      case 9: hipLaunchKernelGGL((KERNEL<true, false, false, true>), __VA_ARGS__); break;                     \
      case 10: hipLaunchKernelGGL((KERNEL<true, false, true, false>), __VA_ARGS__); break;                    \
      case 11: hipLaunchKernelGGL((KERNEL<true, false, true, true>), __VA_ARGS__); break;                     \
      case 12: hipLaunchKernelGGL((KERNEL<true, true, false, false>), __VA_ARGS__); break;                    \
      case 13: hipLaunchKernelGGL((KERNEL<true, true, false, true>), __VA_ARGS__); break;                     \
      case 14: hipLaunchKernelGGL((KERNEL<true, true, true, false>), __VA_ARGS__); break;                     \
      default: hipLaunchKernelGGL((KERNEL<true, true, true, true>), __VA_ARGS__); break;                      \
    }                                                                                                         \
  } while (0)

}  // namespace

extern "C" int v2s_attn_fwd(const v2s_attn_args* a, void* stream) {
  AttnP p;
  if (int e = fill(p, a, "v2s_attn_fwd", false)) return e;
  const int grid = ((p.Nq + 127) / 128) * p.H * p.B;
  V2S_DISPATCH4(attn_fwd_kernel, v2s_opt_tr_read() != 0, p.bias_diag != nullptr, p.causal != 0, p.p16 != 0, dim3(grid), dim3(256), 0,
                (hipStream_t)stream, p);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_attn_delta(const v2s_attn_args* a, float* delta, void* stream) {
  AttnP p;
  if (int e = fill(p, a, "v2s_attn_delta", false)) return e;
  V2S_CHECK(a->d_o && delta, V2S_ERR_ARG, "v2s_attn_delta: needs d_o and delta");
  const long total = (long)p.B * p.Nq * p.H * 8;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, delta);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_attn_bwd(const v2s_attn_args* a, void* stream) {
  AttnP p;
  if (int e = fill(p, a, "v2s_attn_bwd", true)) return e;
  V2S_CHECK(a->delta != nullptr, V2S_ERR_ARG, "v2s_attn_bwd: delta missing (call v2s_attn_delta first)");
  hipStream_t s = (hipStream_t)stream;
  const bool tr = v2s_opt_tr_read() != 0, bias = p.bias_diag != nullptr, causal = p.causal != 0, drop = p.p16 != 0;
  const int gq = ((p.Nq + 127) / 128) * p.H * p.B;
  const size_t dyn = 2 * (size_t)STAGE + (size_t)(p.Nk + 128) * 8;
  V2S_CHECK(dyn <= 64 * 1024, V2S_ERR_SHAPE, "v2s_attn_bwd: Nk=%d too large for the LDS dbias window", p.Nk);
  const int part = v2s_opt_attn_bwd_part();      // profiling aid: 1 = dQ kernel only, 2 = dK/dV kernel only (0 = both)
  if (part != 2) {
    V2S_DISPATCH4(attn_bwd_dq_kernel, tr, bias, causal, drop, dim3(gq), dim3(256), dyn, s, p);
    V2S_LAUNCH_CHECK();
  }
  if (part != 1) {
    const int gk = ((p.Nk + 127) / 128) * p.H * p.B;
    V2S_DISPATCH4(attn_bwd_dkv_kernel, tr, bias, causal, drop, dim3(gk), dim3(256), 0, s, p);
    V2S_LAUNCH_CHECK();
  }
  return V2S_OK;
}

extern "C" int v2s_bias_diag_fwd(const float* table, const int32_t* lut, float* bias_diag, int32_t H, int32_t n,
                                 int32_t num_buckets, void* stream) {
  V2S_CHECK(table && lut && bias_diag && H > 0 && n > 0 && num_buckets > 0, V2S_ERR_ARG, "v2s_bias_diag_fwd: bad args");
  hipLaunchKernelGGL(bias_diag_fwd_kernel, dim3((n * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, table, lut, bias_diag, H, n);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_bias_bucket_bwd(const float* dbias_diag, const int32_t* lut, float* dtable, int32_t H, int32_t n,
                                   int32_t num_buckets, void* stream) {
  V2S_CHECK(dbias_diag && lut && dtable && H > 0 && n > 0 && num_buckets > 0, V2S_ERR_ARG, "v2s_bias_bucket_bwd: bad args");
  hipLaunchKernelGGL(bias_bucket_bwd_kernel, dim3(num_buckets, H), dim3(256), 0, (hipStream_t)stream, dbias_diag, lut, dtable, H, n);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}
