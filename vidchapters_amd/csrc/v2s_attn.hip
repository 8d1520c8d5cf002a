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
// head_dim 64 makes these kernels instruction-issue-bound (few MFMA flops per score element: rocprofv3 SQ counters put some wave
// issuing on 70-80 % of the SIMD cycles with the matrix pipe 16-23 % busy, profiles/r01_pmc_sq_attention.txt), so round 3 rebuilt the
// loops around the instruction count per score element:
//   * everything that does not change from tile to tile is staged ONCE per block: the relative-position bias window of the block
//     (all diagonals k - q it can touch, four 1-float-shifted copies so that a lane fetches its 4 consecutive diagonals with one
//     aligned ds_read_b128), the key flags (masked / out of range) and the per-tile "any / all flagged" state.  The loop body only
//     moves K/V (or Q/dO) tiles: four buffer_load_dwordx4 per thread whose out-of-range rows the hardware bounds check zero-fills
//     (no address compares, no branches) and four ds_write_b128;
//   * the bias enters through the ACCUMULATOR: the score MFMAs start from bias / scale read straight from the LDS window, so the
//     softmax needs no bias add; max(), the exponent argument (one packed fma per two elements: s * scale*log2e - m), the row sum,
//     the rescale of the output accumulator (skipped when no row maximum of the wave moved), the bf16 packing and the backward
//     products all run on two elements per instruction (v_pk_fma/mul/add_f32, v_max3_f32, v_cvt_pk_bf16_f32);
//   * attention-probability dropout draws TWO 16-bit numbers from one 32-bit hash (see drop_* below) and applies them without
//     v_cmp / v_cndmask: in the forward as an AND mask on the packed bf16 pair (5 instructions per 2 elements instead of 8);
//   * "clean" tiles (no masked or out-of-range key, no causal edge) take a branch-free path; tiles whose keys are all masked
//     (padding tail) or all in the causal future are skipped outright once every row of the wave has seen a real key -- their
//     probabilities are exactly 0 then, so the result is bit-identical to processing them (rows that have seen no real key keep
//     the reference's uniform distribution semantics and are never skipped);
//   * delta = rowsum(dO * O) is computed by the dQ kernel from the fragments it loads anyway and handed to the dK/dV kernel,
//     together with m + log2 l and the dropout row seed, as one float4 per row (no separate delta launch, one 16-byte load per
//     row and tile in the dK/dV kernel instead of three loads, a log2 and a hash).
#include <math.h>
#include "v2s_common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float MASKED2 = -3.0e38f;     // finite stand-in for finfo(float32).min in the log2 domain
constexpr float REAL_MIN = -1.0e37f;    // running max above this <=> the row has seen an unmasked key
constexpr int HD = 64;
// blocks per CU the two backward kernels are compiled for (register budget 512 / (2 * blocks) per lane) -- an experiment knob
// (tools/build_attn_variants.sh): 3 needs <= 168 registers and the two-copy bias window
#ifndef ATTN_BWD_OCC
#define ATTN_BWD_OCC 2
#endif
#ifndef ATTN_DKV_KBW
#define ATTN_DKV_KBW 2
#endif
#ifndef ATTN_DKV_OCC
#define ATTN_DKV_OCC ATTN_BWD_OCC
#endif
#ifndef ATTN_WAVEZ
#define ATTN_WAVEZ 1            // 0: a dQ wave whose 32 rows have a zero dO still computes (A/B builds)
#endif
#ifndef ATTN_ZROWS
#define ATTN_ZROWS 1            // 0: the backward kernels do not look for all-zero dO rows (A/B builds)
#endif
#ifndef ATTN_TRIM
#define ATTN_TRIM 1             // 0: the forward / dQ loops stage the trailing all-masked key tiles too (rounds 1-5; A/B builds)
#endif
#ifndef ATTN_DKV_EARLY
#define ATTN_DKV_EARLY 1        // 0: no early exit of all-masked key blocks (debug builds)
#endif
#ifndef ATTN_ABL
#define ATTN_ABL 0
#endif
#ifndef ATTN_DKV_PIPE2
#define ATTN_DKV_PIPE2 0
#endif
#ifndef ATTN_DQ_PREF
#define ATTN_DQ_PREF 0
#endif
#ifndef ATTN_DKV_PREF
#define ATTN_DKV_PREF 1
#endif
#ifndef ATTN_PRIO
#define ATTN_PRIO 0
#endif

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
  uint32_t tm1pk, tpk; int ts32;   // dropout thresholds: (Ts-1) and Ts replicated in both 16-bit halves, Ts << 16 (Ts = p16 - 32768)
  const bf16_t* d_o; long do_bs, do_rs;
  float* rowstat;         // [B][H][Nq][4]: -(m + log2 l), exponent of a masked element, -delta, dropout row seed (written by the dQ kernel)
  bf16_t *dq, *dk, *dv;
  long dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs;
  float* dbias_diag;
  int far_lo, far_hi;   // all relative positions <= far_lo (>= far_hi) share one bias bucket
  const int* seq_off;   // packed self-attention: sequence b = rows [seq_off[b], seq_off[b+1]) of every operand (batch strides unused)
  int seq_q_only;       // seq_off applies to the query side only (q, o, d_o, dq); K / V / dK / dV stay dense [B][Nk] (cross-attention)
  const int* kv_seq_off; // K / V / dK / dV packed with their OWN row offsets (cross-attention over a padding-free memory); else see above
  int order;             // option attn_order: 0 = groups dealt to the XCDs round-robin (default, see block_group), 1 = every XCD a contiguous range of groups (rounds 1-5),
                         // 2 = round-robin without the dK / dV kernel's two passes, 400 + n / 500 + n = sweep knobs (tools/attn_order_sweep.py)
};

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

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

// ---- tile movement -------------------------------------------------------------------------------------
// A [rows][64] bf16 slice of one (sequence, head) as a raw buffer: loads of rows >= `rows` return zeros (hardware bounds check), so
// the tile loop needs no row compares.  Everything that varies per lane or per tile goes into the VGPR offset (the range check of a
// raw buffer covers the VGPR + immediate offset only).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const bf16_t* base, long rs, int rows) {
  const int bytes = rows > 0 ? (int)(((long)(rows - 1) * rs + HD) * 2) : 0;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base), 0, bytes, 0x00020000);
}
// cooperative 64x64 tile load: thread -> (row = tid>>3 (+32), 16-byte chunk = tid&7); voff = byte offset of (tile row 0 + tid>>3, chunk)
__device__ __forceinline__ void tile_load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t step32, uint4 (&out)[2]) {
  const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
  const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, voff + step32, 0, 0);
  out[0] = make_uint4(a[0], a[1], a[2], a[3]);
  out[1] = make_uint4(b[0], b[1], b[2], b[3]);
}
__device__ __forceinline__ void tile_store(char* tile, int tid, const uint4 (&r)[2]) {
  const int chunk = tid & 7, rr = tid >> 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(tile + tile_off(rr + i * 32, chunk * 8)) = r[i];
}

// ---- two-elements-per-instruction fp32 helpers ------------------------------------------------------------
// two fp32 -> one packed bf16 pair with a single v_cvt_pk_bf16_f32 (element-wise casts compile to one conversion per element
// plus a v_perm to merge them)
__device__ __forceinline__ uint32_t cvt_pk(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
// (the compiler splits <2 x float> arithmetic that feeds scalar transcendental ops: force the packed forms)
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float vmax(float a, float b) {       // v_max_f32 without the canonicalising v_max x, x, x hipcc puts in front of fmaxf
  float d;
  asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float max3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32; x <= ~0 here
// hipcc does not pad hardware hazards around inline asm (cdna_hip_programming.md 5.7):
//   * a v_exp_f32 result needs one wait state before a non-transcendental VALU reads it -> the packed ops whose operands may come
//     straight from fast_exp2 start with s_nop 0 (the *_t variants);
//   * an MFMA result must not be read by an asm statement before the matrix pipe has written it back (up to 18 wait states after
//     the issue of an 8-pass MFMA).  mfma_settle() sits between a group of MFMAs and the first asm reader of their accumulators: the
//     "+v" operands order it after every MFMA of the group and before every reader, the nops inside cover the write-back.
__device__ __forceinline__ f32x2 pk_add_t(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("s_nop 0\n\tv_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_mul_t(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ void mfma_settle(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, f32x4& a4, f32x4& a5, f32x4& a6, f32x4& a7) {
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
}

__device__ __forceinline__ void mfma_settle4(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3) {
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
}

// wave priority: the wave that is in an MFMA phase (latency-bound: LDS fragment reads -> MFMA) is preferred by the issue arbiter over a wave of
// another block that is in its long VALU phase (ATTN_PRIO experiment)
__device__ __forceinline__ void prio_hi() {
#if ATTN_PRIO
  __builtin_amdgcn_s_setprio(ATTN_PRIO);
#endif
}
__device__ __forceinline__ void prio_lo() {
#if ATTN_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}

// ---- attention-probability dropout ------------------------------------------------------------------------
// One 32-bit hash serves the key pair (2j, 2j+1) of a row:
//     h(row, j) = mul24( (rowseed(b,h,q) ^ pairhash(j)) & 0xFFFFFF, C2 )        (low 32 bits of the 24 x 24 bit product)
//     draw(q, k) = signed 16-bit half (k & 1) of h(row, k >> 1);   element kept  <=>  draw >= Ts = p16 - 32768,
// i.e. P(drop) = p16 / 65536 exactly as before (inv_keep is derived from p16).  pairhash(j) = ((j & ~31) * C1) ^ ((j & 6) * C1) ^
// ((j & 0x19) * C1): in the forward / dQ kernels a lane's 16 keys of a tile are k = k0 + 4g + (16 kb + r), so the (j & 6) = 2g term is
// folded into the row seed once per kernel, the (j & ~31) = k0 / 2 term once per tile (scalar multiply), and the last term is an
// instruction literal -- one v_xor and one v_mul_u32_u24 per PAIR, no address arithmetic.  The decision never goes through
// v_cmp / v_cndmask in those kernels:
//   forward: e = sat16(Ts - 1 - draw) per half (v_pk_sub_i16 clamp) is negative <=> keep; v_pk_ashrrev_i16 15 turns it into
//            0xFFFF / 0 per half = an AND mask for the packed bf16 probability pair;
//   dQ:      d = sat16(draw - Ts) is negative <=> drop; the two sign bits, spread to 32 bits (v_ashrrev_i32 31 / v_bfe_i32 15,1),
//            clear the fp32 dP elements (v_bfi_b32) before the packed fma that forms dP / keep - delta;
//   dK/dV:   a lane owns ONE key and four rows, so the pair does not help; it evaluates its half with one 32-bit compare:
//            mul24(a, C2) << (k odd ? 0 : 16) moves the half into the top 16 bits, kept <=> (int32) >= Ts << 16.
// Same definition in all three kernels (forward mask == backward mask; tests extract it from the forward and replay it).
constexpr uint32_t DROP_C1 = 0x9E3779B1u, DROP_C2 = 0x00EBCA77u, DROP_M24 = 0x00FFFFFFu;
__device__ __forceinline__ uint32_t drop_rowseed(uint32_t seed, uint32_t rowid) { return v2s_hash32(seed ^ (rowid * 0x9E3779B1u)) & DROP_M24; }
__device__ __forceinline__ uint32_t drop_pairhash(uint32_t j) { return (((j & ~31u) * DROP_C1) ^ ((j & 6u) * DROP_C1) ^ ((j & 0x19u) * DROP_C1)) & DROP_M24; }
// AND mask (0xFFFF per kept half) for the packed pair whose hash input is a = rowseed ^ pairhash
__device__ __forceinline__ uint32_t drop_keepmask_pk(uint32_t a, uint32_t tm1pk) {
  const uint32_t h = __umul24(a, DROP_C2);
  uint32_t e, mk;
  asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(e) : "v"(tm1pk), "v"(h));
  asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(mk) : "v"(e));   // op_sel_hi: the HIGH lane also takes the low half of the inline
                                                                              // constant as its shift amount (its high half is 0)
  return mk;
}
// 32-bit "dropped" masks (all ones / zero) of the even-key (lo) and odd-key (hi) element of a pair
__device__ __forceinline__ void drop_dropmask32(uint32_t a, uint32_t tpk, uint32_t& lo, uint32_t& hi) {
  const uint32_t h = __umul24(a, DROP_C2);
  uint32_t d;
  asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(d) : "v"(h), "v"(tpk));
  hi = (uint32_t)((int32_t)d >> 31);
  lo = (uint32_t)__builtin_amdgcn_sbfe((int32_t)d, 15, 1);
}
__device__ __forceinline__ float clear_if(uint32_t mask, float x) {   // mask all ones -> 0, mask zero -> x  (v_bfi_b32)
  return __uint_as_float(~mask & __float_as_uint(x));
}

// ---- block-level staging (once per block) ------------------------------------------------------------------
constexpr int KV_TILE = 8192;                 // one [64][64] bf16 tile
constexpr int STAGE2 = 2 * KV_TILE;           // K|V (or Q|dO) of one pipeline stage
// bias window: WL = len64 + 128 floats (len64 = nominal key / query count rounded up to 64), four copies; copy s holds w[i+s] at index
// i.  CS = floats per copy = WL rounded up to 64, plus 16: copy s starts 16 banks after copy s-1, so the four copies that lanes with
// consecutive diagonals read (same 4-float group, different shift) fall on different banks.
__host__ __device__ __forceinline__ int bias_cs(int len64) { return ((len64 + 128 + 63) & ~63) + 16; }
// read of w[i0 .. i0+3] for any i0 >= 0: NC = 4 shifted copies -> one aligned ds_read_b128; NC = 2 (long sequences: half the LDS, so
// that the block count per CU does not drop) -> two aligned ds_read_b64 from the copy with the parity of i0
template <int NC>
__device__ __forceinline__ f32x4 bias_read4(const char* win, int CS, int i0) {
  if (NC == 4) {
    const int s = i0 & 3;
    return *reinterpret_cast<const f32x4*>(win + (s * CS + (i0 - s)) * 4);
  } else {
    const int s = i0 & 1;
    const char* b = win + (s * CS + (i0 - s)) * 4;
    const f32x2 lo = *reinterpret_cast<const f32x2*>(b), hi = *reinterpret_cast<const f32x2*>(b + 8);
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
  }
}
// window entry i <-> relative position d = i + dbase; value = bias_diag[h][d + Nq - 1] * mul (0 outside the table)
// REV: the window is stored mirrored (entry i holds relative position (WL - 1 - i) + dbase), so that a lane whose four elements sit at
// DEcreasing relative positions (the dK/dV kernel: four consecutive query rows of one key) reads them in order with one aligned read
template <int NC, bool REV = false>
__device__ __forceinline__ void bias_stage(char* win, int CS, int WL, int dbase, const float* __restrict__ diag_h, int ndiag, int nq, float mul,
                                           int tid) {
  float* b = reinterpret_cast<float*>(win);
  for (int i = tid; i < WL; i += 256) {
    const int gi = (REV ? WL - 1 - i : i) + dbase + nq - 1;
    const float w = (gi >= 0 && gi < ndiag) ? diag_h[gi] * mul : 0.f;
#pragma unroll
    for (int s = 0; s < NC; ++s)
      if (i - s >= 0) b[s * CS + i - s] = w;
  }
}
// key flags (0 keep, 1 masked, 2 out of range) of all len64 keys, then per 64-key tile: bit 0 = some key flagged, bit 1 = all flagged.
// Contains two block barriers.
__device__ __forceinline__ void flags_stage(uint8_t* flag, uint8_t* state, int len64, int nk_, const uint8_t* __restrict__ mask_b, int tid) {
  for (int k = tid; k < len64; k += 256) flag[k] = (k >= nk_) ? 2 : ((mask_b && mask_b[k] == 0) ? 1 : 0);
  __syncthreads();
  const int lane = tid & 63;
  for (int t = tid >> 6; t < (len64 >> 6); t += 4) {
    const unsigned long long any = __ballot(flag[t * 64 + lane] != 0);
    if (lane == 0) state[t] = (uint8_t)((any != 0ull ? 1 : 0) | (any == ~0ull ? 2 : 0));
  }
  __syncthreads();
}


// Key tiles the forward / dQ loops have to visit.  Non-causal attention with at least one visible key: every query row sees that key, so every row's
// statistics are real by the time the loop is past the LAST tile that holds a visible key, and each all-flagged tile after it would be skipped
// (`skip` below) -- after its K / V tiles were loaded, staged and waited for at a block barrier (measured: 27 % / 40 % of a computed tile in the
// forward / dQ kernel).  Those tiles are cut off the loop instead: same results bit for bit.  A sequence without any visible key keeps every tile
// (its rows are uniform over all keys, the reference's finfo.min arithmetic), and so do the causal variants (a row above the first visible key is
// not real, and the per-wave `future` test already ends their useful range).  Block-uniform by construction (reads the block's tile states).
template <bool CAUSAL>
__device__ __forceinline__ int tiles_to_visit(const uint8_t* state, int ntiles) {
  if (CAUSAL || !ATTN_TRIM) return ntiles;
  int t = ntiles;
  while (t > 0 && (state[t - 1] & 2)) --t;
  return t > 0 ? t : ntiles;
}

// Block -> (group, block within the group) for the three kernels; a group = the nper query / key blocks of one (sequence, head), which share its K / V
// (or Q / dO) tiles in L2.  The hardware deals workgroups to the 8 XCDs round-robin and IN ORDER: when the XCD whose turn it is has no free slot, the
// dispatch of every later block waits.  With each XCD walking its own contiguous range of groups (xcd_remap: XCD x owned sequences 4x .. 4x + 3 of a
// 32-sequence batch), a batch with different sequence lengths -- masked / skipped tiles are cheap -- makes the XCDs free their slots at different rates and
// the whole launch advances at the pace of the slowest one.  Round 6: the groups are dealt to the XCDs round-robin instead (group g -> XCD g mod 8, the
// blocks of a group still adjacent on that XCD), so that all eight XCDs work through the SAME sequences at the same time: encoder layer, lengths uniform
// in [0.7 N, N]: forward 172 -> 165, dQ 268 -> 258, dK/dV 318 -> 303 us; lengths in [0.4 N, N]: -8.6 % over the three; no padding: +-0 (profiles/r06_attn_dispatch_order.txt).
__device__ __forceinline__ void block_group(int bid, int nwg, int nper, int ngroups, int plain, int& grp, int& blk) {
  if (plain >= 400 && (ngroups & 7) == 0 && plain - 400 < nper) {      // two passes: the first (plain - 400) blocks of every group, then the others
    const int xcd = bid & 7, loc = bid >> 3, nh = plain - 400, g8 = ngroups >> 3;
    if (loc < nh * g8) { grp = (loc / nh) * 8 + xcd; blk = loc % nh; }
    else { const int r = loc - nh * g8, nr = nper - nh; grp = (r / nr) * 8 + xcd; blk = nh + r % nr; }
  } else if (plain != 1 && (ngroups & 7) == 0) {      // (plain == 2: round-robin only)
    const int xcd = bid & 7, loc = bid >> 3;
    grp = (loc / nper) * 8 + xcd; blk = loc - (loc / nper) * nper;
  } else {
    const int id = xcd_remap(bid, nwg);
    blk = id % nper; grp = id / nper;
  }
}

// Block-wide AND / MAX of a per-thread value through four LDS words at `scratch` (dynamic LDS that is not live at the call; the kernels' LDS budget is
// set as dynamic, a static array would add to it).  Three barriers: the words may be live before the call, and they are about to be overwritten after it
// (round 6: without the last barrier a wave already committing its first tile overwrote the words before a slower wave had read its answer).
__device__ __forceinline__ bool block_and(bool c, char* scratch, int lane, int wave) {
  const bool w = __all(c);
  __syncthreads();
  if (lane == 0) reinterpret_cast<int*>(scratch)[wave] = w ? 1 : 0;
  __syncthreads();
  const int4 v = *reinterpret_cast<const int4*>(scratch);
  __syncthreads();
  return (v.x & v.y & v.z & v.w) != 0;
}
__device__ __forceinline__ int block_max(int x, char* scratch, int lane, int wave) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = max(x, __shfl_xor(x, o, 64));
  __syncthreads();
  if (lane == 0) reinterpret_cast<int*>(scratch)[wave] = x;
  __syncthreads();
  const int4 v = *reinterpret_cast<const int4*>(scratch);
  __syncthreads();
  return max(max(v.x, v.y), max(v.z, v.w));
}
// Rows whose dO is all zero (pad rows of a dense batch: nothing downstream reads them, so their gradient is exactly zero -- modeling_t5.py computes
// them all the same).  For such a row dP = dO V^T = 0 and delta = rowsum(dO * O) = 0, hence dS = P * (dP - delta) = 0: it adds nothing to dQ, dK, dV or
// the bias gradient, whatever its probabilities are.  The dQ kernel reads every dO row anyway (delta): it marks the zero rows in bit 31 of the row-seed
// word of the row statistics, a block of 128 zero rows writes its zeros and leaves before the key loop, and the dK / dV kernel ends its query loop at
// the last row that is not marked.  Bit-identical results (the skipped terms are +-0 added to an accumulator); -0 counts as zero.
constexpr uint32_t ZROW_BIT = 0x80000000u;

// ====================================================================================== forward
// block = 4 waves x 32 query rows; loop over 64-key tiles.  Three blocks per CU (<= 168 VGPRs) instead of two: inside a wave the
// score MFMAs, the softmax VALU work and the PV MFMAs are serial, and only waves in different phases overlap them, so a third
// wave per SIMD is worth 8-13 % (tools/attn_ab.py).  The causal variants spill into their hot path at that budget and stay at two.
// LDS: [2 x (K tile | V tile)] [bias window] [key flags] [tile states]
template <bool TR, bool BIAS, bool CAUSAL, bool DROP, int NC>
__global__ __launch_bounds__(256, CAUSAL ? 2 : 3) void attn_fwd_kernel(const AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int nqb = (p.Nq + 127) >> 7;
  int qblk, bh;
  block_group(blockIdx.x, gridDim.x, nqb, p.B * p.H, p.order >= 500 ? p.order - 100 : (p.order >= 400 ? 0 : p.order), bh, qblk);
  const int h = bh % p.H, b = bh / p.H;
  const int Q0 = qblk * 128, wq0 = wave * 32;
  const int row0_ = p.seq_off ? p.seq_off[b] : 0;                          // packed (varlen) self-attention: first row of sequence b
  const int nq_ = p.seq_off ? p.seq_off[b + 1] - row0_ : p.Nq;
  const int* kso_ = p.kv_seq_off ? p.kv_seq_off : ((p.seq_off && !p.seq_q_only) ? p.seq_off : nullptr);     // row offsets of the key side
  const int krow0_ = kso_ ? kso_[b] : 0, nk_ = kso_ ? kso_[b + 1] - krow0_ : p.Nk;
  if (Q0 >= nq_) return;                                                   // block beyond the end of a short sequence

  const int len64 = (p.Nk + 63) & ~63;
  const int CS = BIAS ? bias_cs(len64) : 0;
  char* s_bias = smem + 2 * STAGE2;
  uint8_t* s_flag = reinterpret_cast<uint8_t*>(s_bias + NC * CS * 4);
  uint8_t* s_state = s_flag + len64;

  const bf16_t* qp = p.q + (p.seq_off ? (long)row0_ * p.q_rs : (long)b * p.q_bs) + h * HD;
  const bf16_t* kp = p.k + (kso_ ? (long)krow0_ * p.k_rs : (long)b * p.k_bs) + h * HD;
  const bf16_t* vp = p.v + (kso_ ? (long)krow0_ * p.v_rs : (long)b * p.v_bs) + h * HD;
  const __amdgpu_buffer_rsrc_t krs = tile_rsrc(kp, p.k_rs, nk_), vrs = tile_rsrc(vp, p.v_rs, nk_);
  uint32_t kvoff = (uint32_t)(((tid >> 3) * p.k_rs + (tid & 7) * 8) * 2), vvoff = (uint32_t)(((tid >> 3) * p.v_rs + (tid & 7) * 8) * 2);
  const uint32_t kstep32 = (uint32_t)(32 * p.k_rs * 2), vstep32 = (uint32_t)(32 * p.v_rs * 2);

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
  uint4 rk[2], rv[2];
  tile_load(krs, kvoff, kstep32, rk);
  tile_load(vrs, vvoff, vstep32, rv);

  if (BIAS) bias_stage<NC>(s_bias, CS, len64 + 128, -(Q0 + 127), p.bias_diag + (long)h * (p.Nq + p.Nk - 1), p.Nq + p.Nk - 1, p.Nq, 1.0f / p.scale, tid);
  flags_stage(s_flag, s_state, len64, nk_, p.key_mask ? p.key_mask + (long)b * p.Nk : nullptr, tid);

  float m[2] = {-INFINITY, -INFINITY}, lsum[2] = {0.f, 0.f};
  uint32_t rowseed[2] = {0u, 0u};
  if (DROP) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
      rowseed[qb] = drop_rowseed(v2s_salted(p.seed, p.salt), (uint32_t)((b * p.H + h) * p.Nq + Q0 + wq0 + qb * 16 + li)) ^ ((uint32_t)(2 * g) * DROP_C1);
  }
  f32x4 ot[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int db = 0; db < 4; ++db) ot[qb][db] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ntiles = tiles_to_visit<CAUSAL>(s_state, (nk_ + 63) >> 6);
  const float sc2 = p.scale * LOG2E;
  const int qmin = Q0 + wq0, qmax = qmin + 31;
  // lane part of the bias-window index of element (qb, kb, r = 0): (k0 + kb*16 + 4g) - (Q0 + qq) + (Q0 + 127)
  const int bidx0 = 4 * g + 127 - (wq0 + li);

  tile_store(smem, tid, rk);
  tile_store(smem + KV_TILE, tid, rv);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const char* sK = smem + (t & 1) * STAGE2;
    const char* sV = sK + KV_TILE;
    const int k0 = t * 64;
    if (t + 1 < ntiles) {
      kvoff += 2 * kstep32; vvoff += 2 * vstep32;
      tile_load(krs, kvoff, kstep32, rk);
      tile_load(vrs, vvoff, vstep32, rv);
    }
    const int tstate = s_state[t];
    const bool any_flag = (tstate & 1) != 0, all_flag = (tstate & 2) != 0;
    const bool future = CAUSAL && (k0 > qmax + p.causal_off);            // every element causally masked
    const bool edge = CAUSAL && (k0 + 63 > qmin + p.causal_off);         // some element causally masked
    const bool seen = __all((m[0] > REAL_MIN) && (m[1] > REAL_MIN));     // every row already saw a real key
    const bool skip = (all_flag || future) && seen;

    if (!skip) {
      // accumulators start from bias / scale: acc * (scale * log2e) is the biased score in the log2 domain
      f32x4 st[2][4];
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
          st[qb][kb] = BIAS ? bias_read4<NC>(s_bias, CS, k0 + kb * 16 + bidx0 - qb * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const bf16x8 kf = row_frag(sK, kb * 16, ks, lane);
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) st[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qb][ks], st[qb][kb], 0, 0, 0);
        }

      mfma_settle(st[0][0], st[0][1], st[0][2], st[0][3], st[1][0], st[1][1], st[1][2], st[1][3]);
      bf16x8 pf[2][2];
      const bool clean = !any_flag && !edge;
      // clean tiles keep the raw accumulators (x = acc * sc2 - m in one packed fma); tiles with masked elements go through the scaled
      // domain (x = s - m must be EXACTLY 0 for a masked element of a row whose keys are all masked)
      const float mult = clean ? sc2 : 1.0f;
      const f32x2 mult2 = {mult, mult};
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        float mx;
        if (clean) {
          mx = max3(st[qb][0][0], st[qb][0][1], st[qb][0][2]);
          mx = max3(mx, st[qb][0][3], st[qb][1][0]);
          mx = max3(mx, st[qb][1][1], st[qb][1][2]);
          mx = max3(mx, st[qb][1][3], st[qb][2][0]);
          mx = max3(mx, st[qb][2][1], st[qb][2][2]);
          mx = max3(mx, st[qb][2][3], st[qb][3][0]);
          mx = max3(mx, st[qb][3][1], st[qb][3][2]);
          mx = vmax(mx, st[qb][3][3]);
        } else {
          const int q = Q0 + wq0 + qb * 16 + li;
          mx = -INFINITY;
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) {
            const uint32_t f4 = *reinterpret_cast<const uint32_t*>(s_flag + k0 + kb * 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int kk = kb * 16 + 4 * g + r;
              float s = st[qb][kb][r] * sc2;
              uint32_t f = (f4 >> (8 * r)) & 0xffu;
              if (CAUSAL && (k0 + kk) > q + p.causal_off) f |= 1u;
              s = (f & 2u) ? -INFINITY : ((f & 1u) ? MASKED2 : s);
              st[qb][kb][r] = s;
              mx = fmaxf(mx, s);
            }
          }
        }
        mx = vmax(mx, __shfl_xor(mx, 16, 64));
        mx = vmax(mx, __shfl_xor(mx, 32, 64));
        const float mn = vmax(m[qb], mx * mult);
        const float alpha = fast_exp2(m[qb] - mn);
        m[qb] = mn;
        const f32x2 nmn2 = {-mn, -mn};
        f32x2 rs2 = {0.f, 0.f};
        const uint32_t rseed = rowseed[qb] ^ ((uint32_t)(k0 >> 1) * DROP_C1);
        uint32_t pw[4][2];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const f32x2 x01 = pk_fma(f32x2{st[qb][kb][0], st[qb][kb][1]}, mult2, nmn2);
          const f32x2 x23 = pk_fma(f32x2{st[qb][kb][2], st[qb][kb][3]}, mult2, nmn2);
          const f32x2 p01 = {fast_exp2(x01[0]), fast_exp2(x01[1])}, p23 = {fast_exp2(x23[0]), fast_exp2(x23[1])};
          rs2 = pk_add_t(rs2, p01);
          rs2 = pk_add_t(rs2, p23);
          pw[kb][0] = cvt_pk(p01[0], p01[1]);
          pw[kb][1] = cvt_pk(p23[0], p23[1]);
          if (DROP) {                                       // 1/(1-p) is applied once, to the output row
            pw[kb][0] &= drop_keepmask_pk(rseed ^ ((uint32_t)(kb * 8 + 0) * DROP_C1), p.tm1pk);
            pw[kb][1] &= drop_keepmask_pk(rseed ^ ((uint32_t)(kb * 8 + 1) * DROP_C1), p.tm1pk);
          }
        }
        lsum[qb] = lsum[qb] * alpha + (rs2[0] + rs2[1]);
        if (__any(alpha != 1.0f)) {                         // some row maximum of this wave moved: rescale the output accumulators
#pragma unroll
          for (int db = 0; db < 4; ++db) ot[qb][db] *= alpha;      // plain C (compiles to v_pk_mul_f32): these are MFMA accumulators
        }
        pf[qb][0] = __builtin_bit_cast(bf16x8, make_uint4(pw[0][0], pw[0][1], pw[1][0], pw[1][1]));
        pf[qb][1] = __builtin_bit_cast(bf16x8, make_uint4(pw[2][0], pw[2][1], pw[3][0], pw[3][1]));
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
    if (t + 1 < ntiles) {
      char* nx = smem + ((t + 1) & 1) * STAGE2;
      tile_store(nx, tid, rk);
      tile_store(nx + KV_TILE, tid, rv);
    }
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

// ====================================================================================== delta = rowsum(dO * O)  (stand-alone utility)
__global__ __launch_bounds__(256) void attn_delta_kernel(const AttnP p, float* __restrict__ delta) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;   // one thread per 8 elements of a (b,q,h) row
  const long total = (long)p.B * p.Nq * p.H * 8;
  float s = 0.f;
  int b = 0, q = 0, h = 0;
  if (t < total) {
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

// ====================================================================================== backward: dQ (+ dbias, + row statistics)
// same decomposition as the forward: wave = 32 query rows, loop over key tiles.
// Tiles that are fully masked / fully in the causal future contribute exactly zero to dQ and dbias whenever
// the row statistics come from a real key (m > REAL_MIN), and are skipped under that condition.
// The prologue also computes delta = rowsum(dO * O) of the block's rows (it holds the dO fragments anyway) and writes the per-row
// statistics the dK/dV kernel needs (p.rowstat), so v2s_attn_bwd launches the dK/dV kernel AFTER this one.
// LDS: [2 x (K tile | V tile)] [bias-gradient window: (Nk + 128) x 8 bytes] [bias window] [key flags] [tile states]
template <bool TR, bool BIAS, bool CAUSAL, bool DROP, int NC>
__global__ __launch_bounds__(256, ATTN_BWD_OCC) void attn_bwd_dq_kernel(const AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // scalar: the per-block bias-gradient routing below branches on it
  const int nqb = (p.Nq + 127) >> 7;
  int qblk, bh;
  block_group(blockIdx.x, gridDim.x, nqb, p.B * p.H, p.order >= 500 ? p.order - 100 : (p.order >= 400 ? 0 : p.order), bh, qblk);
  const int h = bh % p.H, b = bh / p.H;
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
  const int len64 = (p.Nk + 63) & ~63;
  const int CS = BIAS ? bias_cs(len64) : 0;
  unsigned long long* dbw = reinterpret_cast<unsigned long long*>(smem + 2 * STAGE2);
  char* s_bias = smem + 2 * STAGE2 + (BIAS ? ((p.Nk + 129) & ~1) * 8 : 0);
  uint8_t* s_flag = reinterpret_cast<uint8_t*>(s_bias + NC * CS * 4);
  uint8_t* s_state = s_flag + len64;
  const int ndb = p.Nk + 127;
  if (want_dbias) {
    for (int i = tid; i < ndb; i += 256) dbw[i] = 0ull;
  }
  f32x2 acc_lo = {0.f, 0.f}, acc_hi = {0.f, 0.f};    // bias-gradient mass of the two "far" buckets (no per-diagonal resolution needed)

  const bf16_t* qp = p.q + (p.seq_off ? (long)row0_ * p.q_rs : (long)b * p.q_bs) + h * HD;
  const bf16_t* dop = p.d_o + (p.seq_off ? (long)row0_ * p.do_rs : (long)b * p.do_bs) + h * HD;
  const bf16_t* op = p.o + (p.seq_off ? (long)row0_ * p.o_rs : (long)b * p.o_bs) + h * HD;
  const bf16_t* kp = p.k + (kso_ ? (long)krow0_ * p.k_rs : (long)b * p.k_bs) + h * HD;
  const bf16_t* vp = p.v + (kso_ ? (long)krow0_ * p.v_rs : (long)b * p.v_bs) + h * HD;
  const __amdgpu_buffer_rsrc_t krs = tile_rsrc(kp, p.k_rs, nk_), vrs = tile_rsrc(vp, p.v_rs, nk_);
  uint32_t kvoff = (uint32_t)(((tid >> 3) * p.k_rs + (tid & 7) * 8) * 2), vvoff = (uint32_t)(((tid >> 3) * p.v_rs + (tid & 7) * 8) * 2);
  const uint32_t kstep32 = (uint32_t)(32 * p.k_rs * 2), vstep32 = (uint32_t)(32 * p.v_rs * 2);
  uint4 rk[2], rv[2];
  tile_load(krs, kvoff, kstep32, rk);
  tile_load(vrs, vvoff, vstep32, rv);

  bf16x8 qf[2][2], dof[2][2];
  float m2[2], xmask[2], dl[2];
  const float lg2ik = DROP ? __log2f(p.inv_keep) : 0.f, rik = DROP ? 1.0f / p.inv_keep : 1.0f;
  uint32_t rowseed[2] = {0u, 0u};
  bool rows_real = true, blk_zero = true;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int q = Q0 + wq0 + qb * 16 + li;
    float dsum = 0.f;
    uint32_t nzw = 0u;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0), w = make_uint4(0, 0, 0, 0), o = make_uint4(0, 0, 0, 0);
      if (q < nq_) {
        v = *reinterpret_cast<const uint4*>(qp + (long)q * p.q_rs + ks * 32 + g * 8);
        w = *reinterpret_cast<const uint4*>(dop + (long)q * p.do_rs + ks * 32 + g * 8);
        o = *reinterpret_cast<const uint4*>(op + (long)q * p.o_rs + ks * 32 + g * 8);
      }
      nzw |= (w.x | w.y | w.z | w.w) & 0x7fff7fffu;
      qf[qb][ks] = __builtin_bit_cast(bf16x8, v);
      dof[qb][ks] = __builtin_bit_cast(bf16x8, w);
      float of[8], df[8];
      unpack8(o, of); unpack8(w, df);
#pragma unroll
      for (int j = 0; j < 8; ++j) dsum = fmaf(of[j], df[j], dsum);
    }
    dsum += __shfl_xor(dsum, 16, 64);            // delta = sum over the 64 columns of dO * O: the four lane groups hold 16 each
    dsum += __shfl_xor(dsum, 32, 64);
    if (ATTN_ZROWS) {
      nzw |= (uint32_t)__shfl_xor((int)nzw, 16, 64);
      nzw |= (uint32_t)__shfl_xor((int)nzw, 32, 64);
      blk_zero = blk_zero && (nzw == 0u);
    }
    // 1/l is folded into the exponent: P = exp2(s - (m + log2 l)).  Rows >= Nq: huge offset => P = 0 => dS = 0.  A row whose
    // keys are ALL masked (m <= REAL_MIN) keeps the reference's uniform distribution: its masked elements evaluate to
    // exp2(-log2 l) = 1/l (xmask), every other row's masked elements to 0.
    m2[qb] = 1.0e30f; xmask[qb] = -3.0e38f; dl[qb] = 0.f;
    bool real = true;
    const uint32_t rs24 = DROP ? drop_rowseed(v2s_salted(p.seed, p.salt), (uint32_t)((b * p.H + h) * p.Nq + q)) : 0u;
    if (q < nq_) {
      const long r = ((long)(b * p.H + h)) * p.Nq + q;
      const float mm = p.ml[r * 2], l2 = __log2f(p.ml[r * 2 + 1]);
      real = mm > REAL_MIN;
      m2[qb] = real ? mm + l2 : 0.f;
      xmask[qb] = real ? -3.0e38f : -l2;
      dl[qb] = dsum;
      // for the dK/dV kernel, with the keep scale 1/(1-p) folded in: it evaluates P' = exp2(x + log2 ik) = ik * P directly (kept elements of
      // Pd = P', no multiply) and dS = Pd * dP - P' * (delta / ik)
      if (g == 0) *reinterpret_cast<float4*>(p.rowstat + r * 4) = make_float4(-m2[qb] + lg2ik, real ? xmask[qb] : xmask[qb] + lg2ik, -dsum * rik,
                                                                              __uint_as_float(rs24 | ((ATTN_ZROWS && nzw == 0u) ? ZROW_BIT : 0u)));
    }
    rows_real = rows_real && real;
    rowseed[qb] = rs24 ^ ((uint32_t)(2 * g) * DROP_C1);
  }
  const bool seen = __all(rows_real);

  // all 128 rows of the block have a zero dO: dQ = 0, no bias gradient (the row statistics above are written for the dK / dV kernel all the same)
  const bool wave_zero = ATTN_ZROWS && ATTN_WAVEZ && __all(blk_zero);
  if (ATTN_ZROWS && block_and(blk_zero, smem, lane, wave)) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int q = Q0 + wq0 + qb * 16 + li;
      if (q < nq_) {
        bf16_t* dqp = p.dq + (p.seq_off ? (long)row0_ * p.dq_rs : (long)b * p.dq_bs) + (long)q * p.dq_rs + h * HD;
#pragma unroll
        for (int db = 0; db < 4; ++db) *reinterpret_cast<uint2*>(dqp + db * 16 + 4 * g) = make_uint2(0u, 0u);
      }
    }
    return;
  }

  if (BIAS) bias_stage<NC>(s_bias, CS, len64 + 128, -(Q0 + 127), p.bias_diag + (long)h * (p.Nq + p.Nk - 1), p.Nq + p.Nk - 1, p.Nq, 1.0f / p.scale, tid);
  flags_stage(s_flag, s_state, len64, nk_, p.key_mask ? p.key_mask + (long)b * p.Nk : nullptr, tid);

  f32x4 dqt[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int db = 0; db < 4; ++db) dqt[qb][db] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ntiles = tiles_to_visit<CAUSAL>(s_state, (nk_ + 63) >> 6);
  const float sc2 = p.scale * LOG2E;
  const f32x2 sc22 = {sc2, sc2};
  const float ik = DROP ? p.inv_keep : 1.0f;
  const f32x2 ik2 = {ik, ik};
  const int qmin = Q0 + wq0, qmax = qmin + 31;
  const int bidx0 = 4 * g + 127 - (wq0 + li);

  tile_store(smem, tid, rk);
  tile_store(smem + KV_TILE, tid, rv);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const char* sK = smem + (t & 1) * STAGE2;
    const char* sV = sK + KV_TILE;
    const int k0 = t * 64;
    if (t + 1 < ntiles) {
      kvoff += 2 * kstep32; vvoff += 2 * vstep32;
      tile_load(krs, kvoff, kstep32, rk);
      tile_load(vrs, vvoff, vstep32, rv);
    }
    const int tstate = s_state[t];
    const bool any_flag = (tstate & 1) != 0, all_flag = (tstate & 2) != 0;
    const bool future = CAUSAL && (k0 > qmax + p.causal_off);
    const bool edge = CAUSAL && (k0 + 63 > qmin + p.causal_off);
    const bool skip = ((all_flag || future) && seen) || wave_zero;      // (wave_zero: the wave's 32 rows have dO = 0 -> dS = 0: it only helps staging the tiles)

    if (!skip) {
      f32x4 st[2][4], dp[2][4];
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          st[qb][kb] = BIAS ? bias_read4<NC>(s_bias, CS, k0 + kb * 16 + bidx0 - qb * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
          dp[qb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      prio_hi();
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
      mfma_settle(st[0][0], st[0][1], st[0][2], st[0][3], st[1][0], st[1][1], st[1][2], st[1][3]);
      mfma_settle(dp[0][0], dp[0][1], dp[0][2], dp[0][3], dp[1][0], dp[1][1], dp[1][2], dp[1][3]);
      prio_lo();
#if ATTN_DQ_PREF
      // the transposed K fragments of the first half of the dQ product, requested before the softmax / dS phase (LDS latency under its VALU work)
      bf16x8 ktf0[ATTN_DQ_PREF == 2 ? 8 : 4];
#pragma unroll
      for (int db = 0; db < (ATTN_DQ_PREF == 2 ? 8 : 4); ++db) ktf0[db] = col_frag<TR>(sK, (db >> 2) * 32, (db & 3) * 16, lane);
      __builtin_amdgcn_sched_barrier(0);
#endif
      const bool clean = !any_flag && !edge;
      // bias-gradient routing (after the dS of the whole tile are formed, below): relative positions d = k - q of a 16x16 block
      // (qb, kb) span a 31-wide range; blocks entirely in a far bucket just sum their dS (1: far-low, 2: far-high), only the
      // near-diagonal blocks (3) resolve diagonals.  Most TILES lie entirely in one far bucket (troute != 3): decided once per tile.
      const int troute = !want_dbias ? 0 : ((k0 + 63 - qmin) <= p.far_lo ? 1 : ((k0 - qmax) >= p.far_hi ? 2 : 3));
      bf16x8 dsf[2][2];
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const int qq = wq0 + qb * 16 + li, q = Q0 + qq;
        const uint32_t rseed = rowseed[qb] ^ ((uint32_t)(k0 >> 1) * DROP_C1);
        const float ndl_q = -dl[qb], xm_q = xmask[qb], nm2_q = -m2[qb];
        const f32x2 ndl2 = {ndl_q, ndl_q}, nm22 = {nm2_q, nm2_q};
        // exponent arguments x = s * sc2 - (m + log2 l) of the 16 x 64 block first (ONE clean / masked decision), in place
        if (clean) {
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) {
            const f32x2 x01 = pk_fma(f32x2{st[qb][kb][0], st[qb][kb][1]}, sc22, nm22), x23 = pk_fma(f32x2{st[qb][kb][2], st[qb][kb][3]}, sc22, nm22);
            st[qb][kb] = f32x4{x01[0], x01[1], x23[0], x23[1]};
          }
        } else {
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) {
            const uint32_t f4 = *reinterpret_cast<const uint32_t*>(s_flag + k0 + kb * 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              uint32_t f = (f4 >> (8 * r)) & 0xffu;
              if (CAUSAL && (k0 + kb * 16 + 4 * g + r) > q + p.causal_off) f |= 1u;
              st[qb][kb][r] = (f & 2u) ? -INFINITY : ((f & 1u) ? xm_q : fmaf(st[qb][kb][r], sc2, nm2_q));
            }
          }
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const f32x2 p01 = {fast_exp2(st[qb][kb][0]), fast_exp2(st[qb][kb][1])}, p23 = {fast_exp2(st[qb][kb][2]), fast_exp2(st[qb][kb][3])};   // already divided by l
          f32x2 d01 = {dp[qb][kb][0], dp[qb][kb][1]}, d23 = {dp[qb][kb][2], dp[qb][kb][3]};
          if (DROP) {                                       // dropped elements: dP -> 0, so that u = -delta there
            uint32_t mlo, mhi;
            drop_dropmask32(rseed ^ ((uint32_t)(kb * 8 + 0) * DROP_C1), p.tpk, mlo, mhi);
            d01 = f32x2{clear_if(mlo, d01[0]), clear_if(mhi, d01[1])};
            drop_dropmask32(rseed ^ ((uint32_t)(kb * 8 + 1) * DROP_C1), p.tpk, mlo, mhi);
            d23 = f32x2{clear_if(mlo, d23[0]), clear_if(mhi, d23[1])};
          }
          const f32x2 s01 = pk_mul_t(p01, pk_fma(d01, ik2, ndl2)), s23 = pk_mul_t(p23, pk_fma(d23, ik2, ndl2));   // dS = P * (dP/(1-p) - delta)
          st[qb][kb] = f32x4{s01[0], s01[1], s23[0], s23[1]};
        }
      }
      // bias gradient.  Most tiles lie entirely in one far bucket (troute 1 / 2): their dS just sum up, no per-block decisions.
      if (troute == 1 || troute == 2) {
        f32x2 tsum = {0.f, 0.f};
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) {
            tsum = pk_add(tsum, f32x2{st[qb][kb][0], st[qb][kb][1]});
            tsum = pk_add(tsum, f32x2{st[qb][kb][2], st[qb][kb][3]});
          }
        if (troute == 1) acc_lo = pk_add(acc_lo, tsum);
        else acc_hi = pk_add(acc_hi, tsum);
      } else if (troute == 3) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          const int qq = wq0 + qb * 16 + li;
          const int qlo = Q0 + wq0 + qb * 16;            // rows of this block: qlo .. qlo+15
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) {
            const int klo = k0 + kb * 16;
            const int route = (klo + 15 - qlo) <= p.far_lo ? 1 : ((klo - (qlo + 15)) >= p.far_hi ? 2 : 3);   // wave-uniform: scalar branches
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
              const f32x2 bs = pk_add(f32x2{st[qb][kb][0], st[qb][kb][1]}, f32x2{st[qb][kb][2], st[qb][kb][3]});
              if (route == 1) acc_lo = pk_add(acc_lo, bs);
              else acc_hi = pk_add(acc_hi, bs);
            }
          }
        }
      }
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        dsf[qb][0] = __builtin_bit_cast(bf16x8, make_uint4(cvt_pk(st[qb][0][0], st[qb][0][1]), cvt_pk(st[qb][0][2], st[qb][0][3]),
                                                             cvt_pk(st[qb][1][0], st[qb][1][1]), cvt_pk(st[qb][1][2], st[qb][1][3])));
        dsf[qb][1] = __builtin_bit_cast(bf16x8, make_uint4(cvt_pk(st[qb][2][0], st[qb][2][1]), cvt_pk(st[qb][2][2], st[qb][2][3]),
                                                             cvt_pk(st[qb][3][0], st[qb][3][1]), cvt_pk(st[qb][3][2], st[qb][3][3])));
      }
      prio_hi();
#if ATTN_DQ_PREF
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const bf16x8 ktf = (kh == 0 || ATTN_DQ_PREF == 2) ? ktf0[kh * 4 + db] : col_frag<TR>(sK, kh * 32, db * 16, lane);
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) dqt[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf[qb][kh], dqt[qb][db], 0, 0, 0);
        }
#else
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const bf16x8 ktf = col_frag<TR>(sK, kh * 32, db * 16, lane);
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) dqt[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf[qb][kh], dqt[qb][db], 0, 0, 0);
        }
#endif
      prio_lo();
    }
    if (t + 1 < ntiles) {
      char* nx = smem + ((t + 1) & 1) * STAGE2;
      tile_store(nx, tid, rk);
      tile_store(nx + KV_TILE, tid, rv);
    }
    __syncthreads();
  }

#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int q = Q0 + wq0 + qb * 16 + li;
    if (q < nq_) {
      bf16_t* dqp = p.dq + (p.seq_off ? (long)row0_ * p.dq_rs : (long)b * p.dq_bs) + (long)q * p.dq_rs + h * HD;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        uint2 w;
        w.x = pack2bf(dqt[qb][db][0] * p.scale, dqt[qb][db][1] * p.scale);
        w.y = pack2bf(dqt[qb][db][2] * p.scale, dqt[qb][db][3] * p.scale);
        *reinterpret_cast<uint2*>(dqp + db * 16 + 4 * g) = w;
      }
    }
  }
#ifndef ATTN_DBIAS_FLUSH
#define ATTN_DBIAS_FLUSH 1      // 0: ablation build, the block's bias-gradient window is not added to global memory (results invalid)
#endif
  if (want_dbias && ATTN_DBIAS_FLUSH) {
    // dbw index i <-> (k - q) = i - (Q0 + 127);  global index = (k - q) + Nq - 1.  The far-bucket masses go to the
    // diagonals far_lo / far_hi themselves (same bucket): dbias_diag is meaningful after bucket reduction.
    float* dst = p.dbias_diag + (long)h * (p.Nq + p.Nk - 1);
    const float slo = wave_sum(acc_lo[0] + acc_lo[1]);
    const float shi = wave_sum(acc_hi[0] + acc_hi[1]);
    if (lane == 0) {
      const int glo = p.far_lo + p.Nq - 1, ghi = p.far_hi + p.Nq - 1;
      if (slo != 0.f && glo >= 0 && glo < p.Nq + p.Nk - 1) atomicAdd(dst + glo, slo);
      if (shi != 0.f && ghi >= 0 && ghi < p.Nq + p.Nk - 1) atomicAdd(dst + ghi, shi);
    }
    for (int i = tid; i < ndb; i += 256) {
      const int gi = i - (Q0 + 127) + p.Nq - 1;
      const float v = (float)((double)(long long)dbw[i] * (1.0 / 1099511627776.0));
      if (gi >= 0 && gi < p.Nq + p.Nk - 1 && v != 0.f) atomicAdd(dst + gi, v);
    }
  }
}

// ====================================================================================== backward: dK, dV
// wave = 16 * KBW keys (KBW key blocks), block = 64 * KBW keys, loop over 64-query tiles (Q and dO tiles in LDS).  KBW = 2: 224 registers,
// two blocks per CU; KBW = 1: half the accumulators / K, V fragments / score registers per wave -> three waves per SIMD, twice the LDS
// fragment reads and Q / dO tile traffic per MFMA (round 6, ATTN_DKV_KBW).
// LDS: [2 x (Q tile | dO tile | row statistics 4 x 64 floats | state)] [bias window]
constexpr int DKV_MS = STAGE2;                   // -(m + log2 l)[64], masked-element exponent[64], -delta[64], dropout row seed[64]
constexpr int DKV_STATE = DKV_MS + 4 * 64 * 4;   // int: every row of the tile has real statistics
constexpr int DKV_STAGE = DKV_STATE + 16;
constexpr int KBW = ATTN_DKV_KBW, DKV_BK = 64 * KBW;
// key of (key block kb, lane column li) inside the wave's 16 * KBW keys.  KBW = 2: the two key blocks hold the EVEN and the ODD key of the
// pairs (2 li, 2 li + 1), so that ONE dropout hash per query row serves both of a lane's elements of that row (its low / high 16-bit draw:
// drop_dropmask32, the same evaluation as the dQ kernel) instead of one hash, one shift and one compare per element (round 6: the mask
// was 40 of the 64 VALU instructions per 16 x 32 block of this kernel)
__device__ __forceinline__ int dkv_key(int kb, int li) { return KBW == 2 ? 2 * li + kb : kb * 16 + li; }
template <bool TR, bool BIAS, bool CAUSAL, bool DROP, int NC>
__global__ __launch_bounds__(256, ATTN_DKV_OCC) void attn_bwd_dkv_kernel(const AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int nkb = (p.Nk + DKV_BK - 1) / DKV_BK;
  const int* kso_ = p.kv_seq_off ? p.kv_seq_off : ((p.seq_off && !p.seq_q_only) ? p.seq_off : nullptr);     // row offsets of the key side
  int kblk, bh;
  // (dK / dV only, EMPIRICAL: on top of the round-robin deal, dispatching the first 7 key blocks of every group in a first pass and the others in a second is worth
  // another 3-4 % at every length tested -- N = 1000: 303 -> 294 us, 1100: 355 -> 341, 1200: 397 -> 380, 2000: 1004 -> 962, with or without padding; 5 of 6 at N = 768:
  // 188 -> 179; even first-pass counts and the same split in the forward / dQ kernels gain nothing: tools/attn_order_sweep.py, profiles/r06_attn_dispatch_order.txt)
  const int dkv_order = p.order != 0 ? (p.order >= 500 ? p.order - 100 : p.order) : (nkb >= 8 ? 407 : (nkb >= 6 ? 405 : 0));
  block_group(blockIdx.x, gridDim.x, nkb, p.B * p.H, dkv_order, bh, kblk);
  const int h = bh % p.H, b = bh / p.H;
  const int K0 = kblk * DKV_BK, wk0 = wave * 16 * KBW;
  const int row0_ = p.seq_off ? p.seq_off[b] : 0;
  const int nq_ = p.seq_off ? p.seq_off[b + 1] - row0_ : p.Nq;
  const int krow0_ = kso_ ? kso_[b] : 0, nk_ = kso_ ? kso_[b + 1] - krow0_ : p.Nk;
  if (K0 >= nk_) return;

  const int len64 = (p.Nq + 63) & ~63;
  const int CS = BIAS ? bias_cs(len64 + DKV_BK - 128) : 0;
  char* s_bias = smem + 2 * DKV_STAGE;

  const bf16_t* qp = p.q + (p.seq_off ? (long)row0_ * p.q_rs : (long)b * p.q_bs) + h * HD;
  const bf16_t* dop = p.d_o + (p.seq_off ? (long)row0_ * p.do_rs : (long)b * p.do_bs) + h * HD;
  const bf16_t* kp = p.k + (kso_ ? (long)krow0_ * p.k_rs : (long)b * p.k_bs) + h * HD;
  const bf16_t* vp = p.v + (kso_ ? (long)krow0_ * p.v_rs : (long)b * p.v_bs) + h * HD;
  const __amdgpu_buffer_rsrc_t qrs = tile_rsrc(qp, p.q_rs, nq_), dors = tile_rsrc(dop, p.do_rs, nq_);
  uint32_t qvoff = (uint32_t)(((tid >> 3) * p.q_rs + (tid & 7) * 8) * 2), dovoff = (uint32_t)(((tid >> 3) * p.do_rs + (tid & 7) * 8) * 2);
  const uint32_t qstep32 = (uint32_t)(32 * p.q_rs * 2), dostep32 = (uint32_t)(32 * p.do_rs * 2);
  uint4 rq[2], rdo[2];
  tile_load(qrs, qvoff, qstep32, rq);
  tile_load(dors, dovoff, dostep32, rdo);
  const float* rsp = p.rowstat + ((long)(b * p.H + h)) * p.Nq * 4;
  // per-row statistics of a query tile (written by the dQ kernel): rows >= Nq: P = 0
  float4 rstat = make_float4(-1.0e30f, -3.0e38f, 0.f, 0.f);
  if (tid < 64 && tid < nq_) rstat = *reinterpret_cast<const float4*>(rsp + (long)tid * 4);
  // last query row whose dO is not all zero (marked by the dQ kernel): 16 KB of the sequence's statistics, requested with everything else of the prologue
  int zlast = 0;
  if (ATTN_ZROWS) {
#pragma unroll 4
    for (int r = tid; r < nq_; r += 256)
      if (!(__float_as_uint(rsp[(long)r * 4 + 3]) & ZROW_BIT)) zlast = r + 1;
  }

  bf16x8 kf[KBW][2], vf[KBW][2];
  uint32_t kflag[KBW], kc[KBW], cmul[KBW];
#pragma unroll
  for (int kb = 0; kb < KBW; ++kb) {
    const int k = K0 + wk0 + dkv_key(kb, li);
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
    kc[kb] = drop_pairhash((uint32_t)k >> 1);
    cmul[kb] = (k & 1) ? 0u : 16u;                       // shift that moves this key's 16-bit half of the pair hash into the top half
  }
  const bool keys_clean = __all(kflag[0] == 0u && kflag[KBW - 1] == 0u);
  // Two ways for a block to have nothing to do (decided once, block-wide, through the still unused dynamic LDS; then: write the zeros and leave
  // instead of streaming the sequence's Q / dO through LDS with a barrier per tile for nothing):
  //  * no query row of the sequence has a non-zero dO (nq_eff == 0, see ZROW_BIT; otherwise the query loop ends at the last such row);
  //  * its keys are ALL masked (the padding tail of a dense batch) and every query row has real statistics: each tile would be skipped below
  //    (lengths uniform in [0.7 N, N]: one block in seven).
  const int nq_eff = ATTN_ZROWS ? block_max(zlast, smem, lane, wave) : nq_;
  bool dead = nq_eff == 0;
  if (!dead && ATTN_DKV_EARLY && block_and(kflag[0] != 0u && kflag[KBW - 1] != 0u, smem, lane, wave)) {
    bool real = true;
    for (int r = tid; r < nq_; r += 256) real = real && (rsp[(long)r * 4 + 1] < REAL_MIN);
    dead = block_and(real, smem, lane, wave);
  }
  if (dead) {
#pragma unroll
    for (int kb = 0; kb < KBW; ++kb) {
      const int k = K0 + wk0 + dkv_key(kb, li);
      if (k < nk_) {
        bf16_t* dkp = p.dk + (kso_ ? (long)krow0_ * p.dk_rs : (long)b * p.dk_bs) + (long)k * p.dk_rs + h * HD;
        bf16_t* dvp = p.dv + (kso_ ? (long)krow0_ * p.dv_rs : (long)b * p.dv_bs) + (long)k * p.dv_rs + h * HD;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          *reinterpret_cast<uint2*>(dkp + db * 16 + 4 * g) = make_uint2(0u, 0u);
          *reinterpret_cast<uint2*>(dvp + db * 16 + 4 * g) = make_uint2(0u, 0u);
        }
      }
    }
    return;
  }
  const int kmin = K0 + wk0, kmax = kmin + 16 * KBW - 1;
  f32x4 dkt[KBW][4], dvt[KBW][4];
#pragma unroll
  for (int kb = 0; kb < KBW; ++kb)
#pragma unroll
    for (int db = 0; db < 4; ++db) { dkt[kb][db] = f32x4{0.f, 0.f, 0.f, 0.f}; dvt[kb][db] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // bias window of this key block: entry i <-> relative position d = i + (K0 - len64 + 1)  (q < len64, k - K0 in [0, 128))
  if (BIAS) bias_stage<NC, true>(s_bias, CS, len64 + DKV_BK, K0 - len64 + 1, p.bias_diag + (long)h * (p.Nq + p.Nk - 1), p.Nq + p.Nk - 1, p.Nq, 1.0f / p.scale, tid);

  const int ntiles = (nq_eff + 63) >> 6;
  const float sc2 = p.scale * LOG2E;
  const f32x2 sc22 = {sc2, sc2};
  const float ik = DROP ? p.inv_keep : 1.0f;
  const f32x2 ik2 = {ik, ik};
  // window index of element (q, k): (k - q) - (K0 - len64 + 1) = i0 - r for a lane's four rows r = 0..3 of a block; the window is stored
  // MIRRORED (entry WL - 1 - i), so they are the four consecutive floats from WL - 1 - i0 on
  const int bidx0 = (len64 + DKV_BK - 1) - (wk0 + len64 - 1 - 4 * g);

  auto commit = [&](int s) {
    char* st = smem + s * DKV_STAGE;
    tile_store(st, tid, rq);
    tile_store(st + KV_TILE, tid, rdo);
    if (tid < 64) {
      float* ms = reinterpret_cast<float*>(st + DKV_MS);
      ms[tid] = rstat.x; ms[64 + tid] = rstat.y; ms[128 + tid] = rstat.z; ms[192 + tid] = __uint_as_float(__float_as_uint(rstat.w) & ~ZROW_BIT);
      // "every query row of this tile has real statistics": lets all-masked / all-future key blocks skip the tile
      const unsigned long long real = __ballot(rstat.y < REAL_MIN);
      if (tid == 0) reinterpret_cast<int*>(st + DKV_STATE)[0] = (real == ~0ull);
    }
  };
  commit(0);
  __syncthreads();

  const bool keys_all_masked = __all(kflag[0] != 0u && kflag[KBW - 1] != 0u);
#if ATTN_DKV_PIPE2
  // operands of the NEXT score phase, requested one phase ahead: Q / dO row fragments [ks][qi] of a 32-row half and its bias window entries
  bf16x8 pfq[2][2], pfd[2][2];
  f32x4 pbw[2][KBW];
  auto prefetch_half = [&](const char* tQ, const char* tDO, const int tq0, const int qh) {
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
#if ATTN_DKV_PIPE2 != 2
#pragma unroll
      for (int kb = 0; kb < KBW; ++kb)
        pbw[qi][kb] = BIAS ? bias_read4<NC>(s_bias, CS, bidx0 - dkv_key(kb, li) + tq0 + (2 * qh + qi) * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
#endif
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        pfq[ks][qi] = row_frag(tQ, (2 * qh + qi) * 16, ks, lane);
        pfd[ks][qi] = row_frag(tDO, (2 * qh + qi) * 16, ks, lane);
      }
    }
  };
  auto accumulate_tail = [&](const bf16x8 (&dot)[4], const bf16x8 (&qt)[4], const bf16x8 (&pdf)[KBW], const bf16x8 (&dsf)[KBW]) {
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int kb = 0; kb < KBW; ++kb) {
        dvt[kb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot[db], pdf[kb], dvt[kb][db], 0, 0, 0);
        dkt[kb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt[db], dsf[kb], dkt[kb][db], 0, 0, 0);
      }
  };
  prefetch_half(smem, smem + KV_TILE, 0, 0);
#endif
  bf16x8 pdf[KBW], dsf[KBW];
  bf16x8 dot[4], qt[4];
  for (int t = 0; t < ntiles; ++t) {
    const char* sQ = smem + (t & 1) * DKV_STAGE;
    const char* sDO = sQ + KV_TILE;
    const float* ms = reinterpret_cast<const float*>(sQ + DKV_MS);
    const int q0 = t * 64;
    if (t + 1 < ntiles && !(ATTN_ABL == 2 && t > 0)) {
      qvoff += 2 * qstep32; dovoff += 2 * dostep32;
      tile_load(qrs, qvoff, qstep32, rq);
      tile_load(dors, dovoff, dostep32, rdo);
      const int qn = q0 + 64 + tid;
      rstat = make_float4(-1.0e30f, -3.0e38f, 0.f, 0.f);
      if (tid < 64 && qn < nq_) rstat = *reinterpret_cast<const float4*>(rsp + (long)qn * 4);
    }

    const bool rows_real = reinterpret_cast<const int*>(sQ + DKV_STATE)[0] != 0;
    const bool future = CAUSAL && (kmin > q0 + 63 + p.causal_off);     // every (q, k) pair of this tile is causally masked
    const bool edge = CAUSAL && (kmax > q0 + p.causal_off);
    const bool skip = (keys_all_masked || future) && rows_real;

    if (!skip) {
      const bool clean = keys_clean && !edge && (q0 + 63 < nq_);
      // two halves of 32 query rows each (keeps the live score registers at 2 x KBW fragments per half)
      // S[q][key] and dP[q][key]:  D[row = q = qb*16 + 4g + r][col = key = kb*16 + li]; score accumulators start from bias / scale
      auto scores = [&](const int qh, f32x4 (&st)[2][KBW], f32x4 (&dp)[2][KBW]) {
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
          for (int kb = 0; kb < KBW; ++kb) {
            f32x4 init = f32x4{0.f, 0.f, 0.f, 0.f};
            if (BIAS) {
              init = bias_read4<NC>(s_bias, CS, bidx0 - dkv_key(kb, li) + q0 + (2 * qh + qi) * 16);
            }
            st[qi][kb] = init; dp[qi][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int qi = 0; qi < 2; ++qi) {
            const bf16x8 qfr = row_frag(sQ, (2 * qh + qi) * 16, ks, lane);
            const bf16x8 dfr = row_frag(sDO, (2 * qh + qi) * 16, ks, lane);
#pragma unroll
            for (int kb = 0; kb < KBW; ++kb) {
#if ATTN_ABL == 4       // ablation: no MFMA (the fragments still count as used), results invalid
              st[qi][kb][0] += __builtin_bit_cast(f32x4, qfr)[ks]; dp[qi][kb][0] += __builtin_bit_cast(f32x4, dfr)[ks];
#else
              st[qi][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[kb][ks], st[qi][kb], 0, 0, 0);
              dp[qi][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfr, vf[kb][ks], dp[qi][kb], 0, 0, 0);
#endif
            }
          }
      };
#if ATTN_DKV_PIPE2
      // the same products from operands that were requested one phase earlier (prefetch_half below): no LDS wait in front of the MFMAs
      auto scores_pf = [&](const int qh, f32x4 (&st)[2][KBW], f32x4 (&dp)[2][KBW]) {
#if ATTN_DKV_PIPE2 == 2
        // bias window entries requested here; the dP products (which start from zero) run first and cover their LDS latency
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
          for (int kb = 0; kb < KBW; ++kb)
            st[qi][kb] = BIAS ? bias_read4<NC>(s_bias, CS, bidx0 - dkv_key(kb, li) + q0 + (2 * qh + qi) * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
#else
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
          for (int kb = 0; kb < KBW; ++kb) st[qi][kb] = pbw[qi][kb];
#endif
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
          for (int kb = 0; kb < KBW; ++kb) dp[qi][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int qi = 0; qi < 2; ++qi)
#pragma unroll
            for (int kb = 0; kb < KBW; ++kb) dp[qi][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pfd[ks][qi], vf[kb][ks], dp[qi][kb], 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int qi = 0; qi < 2; ++qi)
#pragma unroll
            for (int kb = 0; kb < KBW; ++kb) st[qi][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pfq[ks][qi], kf[kb][ks], st[qi][kb], 0, 0, 0);
      };
#endif
      // P (dropped) -> dp registers become Pd ; st registers become dS ; packed to the B operands of the second products
      auto softmax_ds = [&](const int qh, f32x4 (&st)[2][KBW], f32x4 (&dp)[2][KBW], bf16x8 (&pdf)[KBW], bf16x8 (&dsf)[KBW]) {
        if (KBW == 2) mfma_settle(st[0][0], st[0][KBW - 1], st[1][0], st[1][KBW - 1], dp[0][0], dp[0][KBW - 1], dp[1][0], dp[1][KBW - 1]);
        else mfma_settle4(st[0][0], st[1][0], dp[0][0], dp[1][0]);
#pragma unroll
        for (int qi = 0; qi < (ATTN_ABL == 3 ? 0 : 2); ++qi) {      // (ablation 3: no softmax / dS arithmetic, results invalid)
          const int qb = 2 * qh + qi;
          const f32x4 nm4 = *reinterpret_cast<const f32x4*>(ms + qb * 16 + 4 * g);          // -(m + log2 l) + log2 ik of rows r = 0..3
          const f32x4 lv = *reinterpret_cast<const f32x4*>(ms + 64 + qb * 16 + 4 * g);      // exponent of a masked element
          const f32x4 nd4 = *reinterpret_cast<const f32x4*>(ms + 128 + qb * 16 + 4 * g);    // -delta / ik
          u32x4 sd4 = u32x4{0u, 0u, 0u, 0u};
          if (DROP) sd4 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint32_t*>(ms) + 192 + qb * 16 + 4 * g);
          const f32x2 nm01 = {nm4[0], nm4[1]}, nm23 = {nm4[2], nm4[3]}, nd01 = {nd4[0], nd4[1]}, nd23 = {nd4[2], nd4[3]};
          // exponent arguments of the 16 x 32 block first (ONE clean / masked decision), in place
          if (clean) {
#pragma unroll
            for (int kb = 0; kb < KBW; ++kb) {
              const f32x2 x01 = pk_fma(f32x2{st[qi][kb][0], st[qi][kb][1]}, sc22, nm01), x23 = pk_fma(f32x2{st[qi][kb][2], st[qi][kb][3]}, sc22, nm23);
              st[qi][kb] = f32x4{x01[0], x01[1], x23[0], x23[1]};
            }
          } else {
#pragma unroll
            for (int kb = 0; kb < KBW; ++kb) {
              const int k = K0 + wk0 + dkv_key(kb, li);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int q = q0 + qb * 16 + 4 * g + r;
                uint32_t f = kflag[kb];
                if (CAUSAL && k > q + p.causal_off) f |= 1u;
                st[qi][kb][r] = (f & 2u) ? -INFINITY : ((f & 1u) ? lv[r] : fmaf(st[qi][kb][r], sc2, nm4[r]));
              }
            }
          }
          // dropout: one hash per query row and key PAIR (KBW = 2: the lane's two elements of the row; drop masks all-ones / zero)
          uint32_t mlo[4] = {0u, 0u, 0u, 0u}, mhi[4] = {0u, 0u, 0u, 0u};
          if (DROP && KBW == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) drop_dropmask32(sd4[r] ^ kc[0], p.tpk, mlo[r], mhi[r]);
          }
#pragma unroll
          for (int kb = 0; kb < KBW; ++kb) {
            // P' = ik * P (the row statistics carry log2 ik: see the dQ kernel's prologue), already divided by l
            const f32x2 p01 = {fast_exp2(st[qi][kb][0]), fast_exp2(st[qi][kb][1])}, p23 = {fast_exp2(st[qi][kb][2]), fast_exp2(st[qi][kb][3])};
            f32x2 pd01 = p01, pd23 = p23;
            if (DROP && KBW == 2) {
              const uint32_t* mk = kb ? mhi : mlo;
              pd01 = f32x2{clear_if(mk[0], p01[0]), clear_if(mk[1], p01[1])}; pd23 = f32x2{clear_if(mk[2], p23[0]), clear_if(mk[3], p23[1])};
            } else if (DROP) {
              float pdv[4] = {pd01[0], pd01[1], pd23[0], pd23[1]};
#pragma unroll
              for (int r = 0; r < 4; ++r)
                pdv[r] = ((int32_t)(__umul24(sd4[r] ^ kc[kb], DROP_C2) << cmul[kb]) >= p.ts32) ? pdv[r] : 0.f;   // full-rate 24-bit multiply + shift
                                                                                                              // (a 32-bit v_mul_lo_u32 by a per-lane constant is quarter rate)
              pd01 = f32x2{pdv[0], pdv[1]}; pd23 = f32x2{pdv[2], pdv[3]};
            }
            // dS = Pd * dP - P' * (delta / ik)
            const f32x2 s01 = pk_fma(pd01, f32x2{dp[qi][kb][0], dp[qi][kb][1]}, pk_mul_t(p01, nd01));
            const f32x2 s23 = pk_fma(pd23, f32x2{dp[qi][kb][2], dp[qi][kb][3]}, pk_mul_t(p23, nd23));
            st[qi][kb] = f32x4{s01[0], s01[1], s23[0], s23[1]};
            dp[qi][kb] = f32x4{pd01[0], pd01[1], pd23[0], pd23[1]};
          }
        }
        // dV^T[d][key] += dO^T[d][q] * Pd[q][key] ; dK^T[d][key] += Q^T[d][q] * dS[q][key]
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb) {
          pdf[kb] = __builtin_bit_cast(bf16x8, make_uint4(cvt_pk(dp[0][kb][0], dp[0][kb][1]), cvt_pk(dp[0][kb][2], dp[0][kb][3]),
                                                           cvt_pk(dp[1][kb][0], dp[1][kb][1]), cvt_pk(dp[1][kb][2], dp[1][kb][3])));
          dsf[kb] = __builtin_bit_cast(bf16x8, make_uint4(cvt_pk(st[0][kb][0], st[0][kb][1]), cvt_pk(st[0][kb][2], st[0][kb][3]),
                                                           cvt_pk(st[1][kb][0], st[1][kb][1]), cvt_pk(st[1][kb][2], st[1][kb][3])));
        }
      };
      // dV^T[d][key] += dO^T[d][q] * Pd[q][key] ; dK^T[d][key] += Q^T[d][q] * dS[q][key]
      // (ATTN_DKV_PREF: the transposed Q / dO fragments of the second products are requested BEFORE the softmax / dS phase, so that their
      // LDS latency passes under its VALU work instead of in front of the MFMAs)
      auto colfrags = [&](const int qh, bf16x8 (&dot)[4], bf16x8 (&qt)[4]) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          dot[db] = col_frag<TR>(sDO, qh * 32, db * 16, lane);
          qt[db] = col_frag<TR>(sQ, qh * 32, db * 16, lane);
        }
      };
      auto accumulate = [&](const bf16x8 (&dot)[4], const bf16x8 (&qt)[4], const bf16x8 (&pdf)[KBW], const bf16x8 (&dsf)[KBW]) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
#pragma unroll
          for (int kb = 0; kb < KBW; ++kb) {
#if ATTN_ABL == 4
            dvt[kb][db][0] += __builtin_bit_cast(f32x4, dot[db])[kb] * __builtin_bit_cast(f32x4, pdf[kb])[db];
            dkt[kb][db][0] += __builtin_bit_cast(f32x4, qt[db])[kb] * __builtin_bit_cast(f32x4, dsf[kb])[db];
#else
            dvt[kb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot[db], pdf[kb], dvt[kb][db], 0, 0, 0);
            dkt[kb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt[db], dsf[kb], dkt[kb][db], 0, 0, 0);
#endif
          }
        }
      };
#if ATTN_DKV_PIPE2
      // software pipeline across the phases of the loop (round 6): every MFMA group runs on operands that were requested a phase earlier --
      // the score products on the Q / dO fragments and the bias window entries fetched under the PREVIOUS half's dV / dK products (prefetch_half,
      // after the tile's commit + barrier when that half belongs to the next tile), the dV / dK products on transposed fragments fetched
      // under the softmax phase
      {
        f32x4 st[2][KBW], dp[2][KBW];
        scores_pf(0, st, dp);
#if ATTN_DKV_PREF
        colfrags(0, dot, qt);
        __builtin_amdgcn_sched_barrier(0);
        softmax_ds(0, st, dp, pdf, dsf);
        __builtin_amdgcn_sched_barrier(0);
        prefetch_half(sQ, sDO, q0, 1);
#else
        softmax_ds(0, st, dp, pdf, dsf);
        __builtin_amdgcn_sched_barrier(0);
        colfrags(0, dot, qt);
        prefetch_half(sQ, sDO, q0, 1);
#endif
        accumulate(dot, qt, pdf, dsf);
      }
      {
        f32x4 st[2][KBW], dp[2][KBW];
        scores_pf(1, st, dp);
#if ATTN_DKV_PREF
        colfrags(1, dot, qt);
        __builtin_amdgcn_sched_barrier(0);
        softmax_ds(1, st, dp, pdf, dsf);
#else
        softmax_ds(1, st, dp, pdf, dsf);
        __builtin_amdgcn_sched_barrier(0);
        colfrags(1, dot, qt);
#endif
      }
#else
#pragma unroll
      for (int qh = 0; qh < 2; ++qh) {
        f32x4 st[2][KBW], dp[2][KBW];
        prio_hi();
        scores(qh, st, dp);
#if ATTN_DKV_PREF
        colfrags(qh, dot, qt);
        __builtin_amdgcn_sched_barrier(0);
        prio_lo();
        softmax_ds(qh, st, dp, pdf, dsf);
#else
        prio_lo();
        softmax_ds(qh, st, dp, pdf, dsf);
        colfrags(qh, dot, qt);
#endif
        prio_hi();
        accumulate(dot, qt, pdf, dsf);
        prio_lo();
      }
#endif
    }
#if ATTN_ABL == 2
    if (t == 0 && ntiles > 1) commit(1);        // ablation: stages written once, no per-tile LDS writes (results invalid)
#else
    if (t + 1 < ntiles) commit((t + 1) & 1);
#endif
#if ATTN_ABL != 1 && ATTN_ABL != 2
    __syncthreads();                            // (ablation 1 / 2: no per-tile barrier, results invalid)
#endif
#if ATTN_DKV_PIPE2
    // first half of the NEXT tile (its stage is visible now), under the second half's dV / dK products
    if (t + 1 < ntiles) prefetch_half(smem + ((t + 1) & 1) * DKV_STAGE, smem + ((t + 1) & 1) * DKV_STAGE + KV_TILE, q0 + 64, 0);
    if (!skip) accumulate_tail(dot, qt, pdf, dsf);
#endif
  }

#pragma unroll
  for (int kb = 0; kb < KBW; ++kb) {
    const int k = K0 + wk0 + dkv_key(kb, li);
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

// ====================================================================================== fp32 reference-grade kernels (option "fp32_io")
// Debug mode for parity work (<= 1e-4 against an fp32 reference): q / k / v / o / dO / dq / dk / dv are FP32 buffers with
// the same element strides, every product and the softmax run in fp32 FMA arithmetic (no MFMA, no bf16 anywhere).  Same semantics as
// the kernels above: score = scale * q.k + bias_diag[h][k - q + Nq - 1]; masked keys / the causal future are REPLACED by a huge
// negative constant (a row without a visible key is uniform over its keys), gradients flow straight through the replacement like the
// reference's `scores + mask` (modeling_t5.py:559); the dropout mask is the same counter-based function (drop_* above), so a masked
// run here can be compared with the MFMA kernels element for element.  Dense layout only (no seq_off packing).  Slow by design.
__device__ __forceinline__ bool f32_keep(uint32_t rowseed24, int k, int ts) {
  const uint32_t h = __umul24(rowseed24 ^ drop_pairhash((uint32_t)k >> 1), DROP_C2);
  const int draw = (k & 1) ? (int)(short)(h >> 16) : (int)(short)(h & 0xffffu);
  return draw >= ts;
}
__device__ __forceinline__ float f32_block_max(float v, float* red, int tid) {
  v = wave_max(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float f32_block_sum(float v, float* red, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
// scores of row q into prob[] (natural domain), returns the row maximum
__device__ __forceinline__ float f32_scores(const AttnP& p, const float* K, const float* sq, const float* bias_h, const uint8_t* mask_b, int q,
                                            float* prob, float* red, int tid) {
  float mx = -INFINITY;
  for (int k = tid; k < p.Nk; k += 256) {
    const float* kr = K + (long)k * p.k_rs;
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < HD; ++d) s = fmaf(sq[d], kr[d], s);
    s *= p.scale;
    if (bias_h) s += bias_h[k - q + p.Nq - 1];
    if ((mask_b && mask_b[k] == 0) || (p.causal && k > q + p.causal_off)) s = MASKED2;
    prob[k] = s;
    mx = fmaxf(mx, s);
  }
  return f32_block_max(mx, red, tid);
}

__global__ __launch_bounds__(256) void attn_f32_fwd_kernel(const AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* prob = reinterpret_cast<float*>(smem);            // [Nk]
  float* part = prob + ((p.Nk + 3) & ~3);                  // [4][64]
  __shared__ float sq[HD], red[4];
  const int tid = threadIdx.x;
  const int q = blockIdx.x % p.Nq, bh = blockIdx.x / p.Nq, h = bh % p.H, b = bh / p.H;
  const float* Q = reinterpret_cast<const float*>(p.q) + (long)b * p.q_bs + (long)q * p.q_rs + h * HD;
  const float* K = reinterpret_cast<const float*>(p.k) + (long)b * p.k_bs + h * HD;
  const float* V = reinterpret_cast<const float*>(p.v) + (long)b * p.v_bs + h * HD;
  if (tid < HD) sq[tid] = Q[tid];
  __syncthreads();
  const float* bias_h = p.bias_diag ? p.bias_diag + (long)h * (p.Nq + p.Nk - 1) : nullptr;
  const uint8_t* mask_b = p.key_mask ? p.key_mask + (long)b * p.Nk : nullptr;
  const float m = f32_scores(p, K, sq, bias_h, mask_b, q, prob, red, tid);
  float l = 0.f;
  for (int k = tid; k < p.Nk; k += 256) { const float e = __expf(prob[k] - m); prob[k] = e; l += e; }
  l = f32_block_sum(l, red, tid);
  const uint32_t rs = p.p16 ? drop_rowseed(v2s_salted(p.seed, p.salt), (uint32_t)((b * p.H + h) * p.Nq + q)) : 0u;
  const int ts = (int)p.p16 - 32768;
  const int d = tid & 63, sl = tid >> 6;
  float acc = 0.f;
  for (int k = sl; k < p.Nk; k += 4) {
    const float pd = (!p.p16 || f32_keep(rs, k, ts)) ? prob[k] : 0.f;
    acc = fmaf(pd, V[(long)k * p.v_rs + d], acc);
  }
  part[sl * HD + d] = acc;
  __syncthreads();
  if (tid < HD) {
    const float o = ((part[tid] + part[HD + tid]) + (part[2 * HD + tid] + part[3 * HD + tid])) * (p.p16 ? p.inv_keep : 1.0f) / l;
    reinterpret_cast<float*>(p.o)[(long)b * p.o_bs + (long)q * p.o_rs + h * HD + tid] = o;
  }
  if (tid == 0 && p.ml) {
    float* mp = p.ml + (((long)(b * p.H + h)) * p.Nq + q) * 2;
    mp[0] = m * LOG2E; mp[1] = l;
  }
}

// dQ (+ dbias) per query row; writes (m, l, delta) to rowstat for the dK/dV kernel
__global__ __launch_bounds__(256) void attn_f32_bwd_dq_kernel(const AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* prob = reinterpret_cast<float*>(smem);            // [Nk] p
  float* dsv = prob + ((p.Nk + 3) & ~3);                   // [Nk] dS
  float* part = dsv + ((p.Nk + 3) & ~3);                   // [4][64]
  __shared__ float sq[HD], sdo[HD], red[4];
  const int tid = threadIdx.x;
  const int q = blockIdx.x % p.Nq, bh = blockIdx.x / p.Nq, h = bh % p.H, b = bh / p.H;
  const float* Q = reinterpret_cast<const float*>(p.q) + (long)b * p.q_bs + (long)q * p.q_rs + h * HD;
  const float* DO = reinterpret_cast<const float*>(p.d_o) + (long)b * p.do_bs + (long)q * p.do_rs + h * HD;
  const float* K = reinterpret_cast<const float*>(p.k) + (long)b * p.k_bs + h * HD;
  const float* V = reinterpret_cast<const float*>(p.v) + (long)b * p.v_bs + h * HD;
  if (tid < HD) { sq[tid] = Q[tid]; sdo[tid] = DO[tid]; }
  __syncthreads();
  const float* bias_h = p.bias_diag ? p.bias_diag + (long)h * (p.Nq + p.Nk - 1) : nullptr;
  const uint8_t* mask_b = p.key_mask ? p.key_mask + (long)b * p.Nk : nullptr;
  const float m = f32_scores(p, K, sq, bias_h, mask_b, q, prob, red, tid);
  float l = 0.f;
  for (int k = tid; k < p.Nk; k += 256) { const float e = __expf(prob[k] - m); prob[k] = e; l += e; }
  l = f32_block_sum(l, red, tid);
  const uint32_t rs = p.p16 ? drop_rowseed(v2s_salted(p.seed, p.salt), (uint32_t)((b * p.H + h) * p.Nq + q)) : 0u;
  const int ts = (int)p.p16 - 32768;
  const float ik = p.p16 ? p.inv_keep : 1.0f;
  float dl = 0.f;
  for (int k = tid; k < p.Nk; k += 256) {
    const float* vr = V + (long)k * p.v_rs;
    float dp = 0.f;
#pragma unroll 8
    for (int d = 0; d < HD; ++d) dp = fmaf(sdo[d], vr[d], dp);
    dp = (!p.p16 || f32_keep(rs, k, ts)) ? dp * ik : 0.f;
    const float pk = prob[k] / l;
    prob[k] = pk;
    dsv[k] = dp;
    dl = fmaf(pk, dp, dl);
  }
  dl = f32_block_sum(dl, red, tid);                        // delta = sum_k P * dP_eff = rowsum(dO * O)
  for (int k = tid; k < p.Nk; k += 256) {
    const float ds = prob[k] * (dsv[k] - dl);
    dsv[k] = ds;
    if (p.dbias_diag) atomicAdd(p.dbias_diag + (long)h * (p.Nq + p.Nk - 1) + (k - q + p.Nq - 1), ds);
  }
  __syncthreads();
  const int d = tid & 63, sl = tid >> 6;
  float acc = 0.f;
  for (int k = sl; k < p.Nk; k += 4) acc = fmaf(dsv[k], K[(long)k * p.k_rs + d], acc);
  part[sl * HD + d] = acc;
  __syncthreads();
  if (tid < HD)
    reinterpret_cast<float*>(p.dq)[(long)b * p.dq_bs + (long)q * p.dq_rs + h * HD + tid] =
        ((part[tid] + part[HD + tid]) + (part[2 * HD + tid] + part[3 * HD + tid])) * p.scale;
  if (tid == 0) *reinterpret_cast<float4*>(p.rowstat + (((long)(b * p.H + h)) * p.Nq + q) * 4) = make_float4(m, l, dl, 0.f);
}

// dK, dV per key: every thread walks a slice of the query rows, 2 x 64 accumulators reduced over the block in rounds of 32 values
__global__ __launch_bounds__(256) void attn_f32_bwd_dkv_kernel(const AttnP p) {
  __shared__ float sk[HD], sv[HD], red[256][33];
  const int tid = threadIdx.x;
  const int k = blockIdx.x % p.Nk, bh = blockIdx.x / p.Nk, h = bh % p.H, b = bh / p.H;
  const float* Qb = reinterpret_cast<const float*>(p.q) + (long)b * p.q_bs + h * HD;
  const float* DOb = reinterpret_cast<const float*>(p.d_o) + (long)b * p.do_bs + h * HD;
  const float* Kr = reinterpret_cast<const float*>(p.k) + (long)b * p.k_bs + (long)k * p.k_rs + h * HD;
  const float* Vr = reinterpret_cast<const float*>(p.v) + (long)b * p.v_bs + (long)k * p.v_rs + h * HD;
  if (tid < HD) { sk[tid] = Kr[tid]; sv[tid] = Vr[tid]; }
  __syncthreads();
  const float* bias_h = p.bias_diag ? p.bias_diag + (long)h * (p.Nq + p.Nk - 1) : nullptr;
  const bool kmasked = p.key_mask && p.key_mask[(long)b * p.Nk + k] == 0;
  const int ts = (int)p.p16 - 32768;
  const float ik = p.p16 ? p.inv_keep : 1.0f;
  float dk[HD], dv[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
  for (int q = tid; q < p.Nq; q += 256) {
    const float* qr = Qb + (long)q * p.q_rs;
    const float* dor = DOb + (long)q * p.do_rs;
    const float4 st = *reinterpret_cast<const float4*>(p.rowstat + (((long)(b * p.H + h)) * p.Nq + q) * 4);     // m, l, delta
    float s = 0.f, dp = 0.f;
#pragma unroll 8
    for (int d = 0; d < HD; ++d) { s = fmaf(qr[d], sk[d], s); dp = fmaf(dor[d], sv[d], dp); }
    s *= p.scale;
    if (bias_h) s += bias_h[k - q + p.Nq - 1];
    if (kmasked || (p.causal && k > q + p.causal_off)) s = MASKED2;
    const float pk = __expf(s - st.x) / st.y;
    const bool keep = !p.p16 || f32_keep(drop_rowseed(v2s_salted(p.seed, p.salt), (uint32_t)((b * p.H + h) * p.Nq + q)), k, ts);
    const float pd = keep ? pk * ik : 0.f;
    const float ds = pk * ((keep ? dp * ik : 0.f) - st.z) * p.scale;
#pragma unroll
    for (int d = 0; d < HD; ++d) { dk[d] = fmaf(ds, qr[d], dk[d]); dv[d] = fmaf(pd, dor[d], dv[d]); }
  }
  float* dkp = reinterpret_cast<float*>(p.dk) + (long)b * p.dk_bs + (long)k * p.dk_rs + h * HD;
  float* dvp = reinterpret_cast<float*>(p.dv) + (long)b * p.dv_bs + (long)k * p.dv_rs + h * HD;
#pragma unroll
  for (int r = 0; r < 4; ++r) {                             // values r*32 .. r*32+31 of the 128 (dk | dv) per round
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; ++j) { const int v = r * 32 + j; red[tid][j] = v < HD ? dk[v] : dv[v - HD]; }
    __syncthreads();
    if (tid < 32) {
      float t = 0.f;
      for (int i = 0; i < 256; ++i) t += red[i][tid];
      const int v = r * 32 + tid;
      if (v < HD) dkp[v] = t; else dvp[v - HD] = t;
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
  V2S_CHECK(a->Nq <= 4096 && a->Nk <= 4096, V2S_ERR_SHAPE, "%s: at most 4096 queries / keys per sequence (Nq=%d Nk=%d)", who, a->Nq, a->Nk);
  V2S_CHECK(((a->q_rs | a->k_rs | a->v_rs | a->o_rs | a->q_bs | a->k_bs | a->v_bs | a->o_bs) % 8) == 0, V2S_ERR_ALIGN, "%s: strides must be multiples of 8 elements", who);
  V2S_CHECK((((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->o) & 15) == 0, V2S_ERR_ALIGN, "%s: pointers must be 16-byte aligned", who);
  V2S_CHECK(a->dropout_p >= 0.f && a->dropout_p < 1.f, V2S_ERR_ARG, "%s: dropout_p out of range", who);
  V2S_CHECK(a->scale > 0.f, V2S_ERR_ARG, "%s: scale must be positive", who);
  // the tile loads address rows through 32-bit byte offsets from the first row of a (sequence, head)
  V2S_CHECK((long)a->Nk * a->k_rs < (1L << 30) && (long)a->Nk * a->v_rs < (1L << 30) && (long)a->Nq * a->q_rs < (1L << 30), V2S_ERR_SHAPE,
            "%s: rows x row stride must stay below 2^30 elements per sequence", who);
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
  {                                               // thresholds of the 16-bit dropout draws (see drop_* in the kernels)
    const int ts = (int)p.p16 - 32768;
    p.tpk = ((uint32_t)ts & 0xFFFFu) * 0x10001u;
    p.tm1pk = ((uint32_t)(ts - 1) & 0xFFFFu) * 0x10001u;
    p.ts32 = (int)((uint32_t)ts << 16);
  }
  p.d_o = (const bf16_t*)a->d_o; p.do_bs = a->do_bs; p.do_rs = a->do_rs; p.rowstat = a->delta;
  p.dq = (bf16_t*)a->dq; p.dk = (bf16_t*)a->dk; p.dv = (bf16_t*)a->dv;
  p.dq_bs = a->dq_bs; p.dq_rs = a->dq_rs; p.dk_bs = a->dk_bs; p.dk_rs = a->dk_rs; p.dv_bs = a->dv_bs; p.dv_rs = a->dv_rs;
  p.dbias_diag = a->dbias_diag;
  p.seq_off = a->seq_off; p.seq_q_only = (a->seq_off && a->seq_q_only) ? 1 : 0; p.kv_seq_off = a->kv_seq_off; p.order = v2s_opt_attn_order();
  V2S_CHECK(!a->seq_off || a->seq_q_only || a->kv_seq_off || (a->Nq == a->Nk && !a->key_mask), V2S_ERR_ARG, "%s: seq_off (packed self-attention) needs Nq == Nk and no key_mask", who);
  V2S_CHECK(!a->kv_seq_off || !a->key_mask, V2S_ERR_ARG, "%s: kv_seq_off (packed keys) excludes key_mask: pad keys simply do not exist", who);
  // far buckets: disabled (every diagonal resolved) unless the caller states lo < hi
  if (a->bias_far_lo < a->bias_far_hi) { p.far_lo = a->bias_far_lo; p.far_hi = a->bias_far_hi; }
  else { p.far_lo = -(1 << 30); p.far_hi = (1 << 30); }
  if (bwd) {
    V2S_CHECK(a->d_o && a->ml && a->dq && a->dk && a->dv, V2S_ERR_ARG, "%s: backward needs d_o, ml, dq, dk, dv", who);
    V2S_CHECK(((a->do_rs | a->do_bs | a->dq_rs | a->dk_rs | a->dv_rs | a->dq_bs | a->dk_bs | a->dv_bs) % 8) == 0, V2S_ERR_ALIGN, "%s: grad strides must be multiples of 8", who);
    V2S_CHECK((long)a->Nq * a->do_rs < (1L << 30), V2S_ERR_SHAPE, "%s: rows x row stride must stay below 2^30 elements per sequence", who);
  }
  return V2S_OK;
}

// dynamic LDS of the three kernels (layouts at the kernels); nc = copies of the bias window
size_t lds_fwd(const AttnP& p, bool bias, int nc) {
  const int len64 = (p.Nk + 63) & ~63;
  return 2 * (size_t)STAGE2 + (bias ? (size_t)nc * bias_cs(len64) * 4 : 0) + len64 + 64;
}
size_t lds_dq(const AttnP& p, bool bias, int nc) {
  const int len64 = (p.Nk + 63) & ~63;
  return 2 * (size_t)STAGE2 + (bias ? (size_t)((p.Nk + 129) & ~1) * 8 + (size_t)nc * bias_cs(len64) * 4 : 0) + len64 + 64;
}
size_t lds_dkv(const AttnP& p, bool bias, int nc) {
  const int len64 = (p.Nq + 63) & ~63;
  return 2 * (size_t)DKV_STAGE + (bias ? (size_t)nc * bias_cs(len64 + DKV_BK - 128) * 4 : 0);
}
constexpr size_t LDS_PER_CU = 160 * 1024;

// compile-time specialisation dispatch: (tr_read, bias, causal, dropout[, window copies]).  Every instantiation may use the whole
// 160 KiB of LDS (long sequences with a bias window: hipFuncSetAttribute once per instantiation).
#define V2S_LAUNCH_ATTN(KERNEL, T_, B_, C_, D_, N_, grid, dyn, stream, p)                                                      \
  do {                                                                                                                         \
    static bool attr__ = false;                                                                                                \
    if (!attr__) {                                                                                                             \
      (void)hipFuncSetAttribute((const void*)KERNEL<T_, B_, C_, D_, N_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr__ = true;                                                                                                           \
    }                                                                                                                          \
    hipLaunchKernelGGL((KERNEL<T_, B_, C_, D_, N_>), grid, dim3(256), dyn, stream, p);                                          \
  } while (0)
#define V2S_LAUNCH_NC(KERNEL, T_, C_, D_, nc, grid, dyn, stream, p)                                          \
  do {                                                                                                       \
    if ((nc) == 2) V2S_LAUNCH_ATTN(KERNEL, T_, true, C_, D_, 2, grid, dyn, stream, p);                        \
    else V2S_LAUNCH_ATTN(KERNEL, T_, true, C_, D_, 4, grid, dyn, stream, p);                                  \
  } while (0)
#define V2S_DISPATCH4(KERNEL, tr, bias, causal, drop, nc, grid, dyn, stream, p)                               \
  do {                                                                                                        \
    const int key__ = ((tr) ? 8 : 0) | ((bias) ? 4 : 0) | ((causal) ? 2 : 0) | ((drop) ? 1 : 0);              \
    switch (key__) {                                                                                          \
      case 0: V2S_LAUNCH_ATTN(KERNEL, false, false, false, false, 4, grid, dyn, stream, p); break;            \
      case 1: V2S_LAUNCH_ATTN(KERNEL, false, false, false, true, 4, grid, dyn, stream, p); break;             \
      case 2: V2S_LAUNCH_ATTN(KERNEL, false, false, true, false, 4, grid, dyn, stream, p); break;             \
      case 3: V2S_LAUNCH_ATTN(KERNEL, false, false, true, true, 4, grid, dyn, stream, p); break;              \
      case 4: V2S_LAUNCH_NC(KERNEL, false, false, false, nc, grid, dyn, stream, p); break;                    \
      case 5: V2S_LAUNCH_NC(KERNEL, false, false, true, nc, grid, dyn, stream, p); break;                     \
      case 6: V2S_LAUNCH_NC(KERNEL, false, true, false, nc, grid, dyn, stream, p); break;                     \
      case 7: V2S_LAUNCH_NC(KERNEL, false, true, true, nc, grid, dyn, stream, p); break;                      \
      case 8: V2S_LAUNCH_ATTN(KERNEL, true, false, false, false, 4, grid, dyn, stream, p); break;             \
      case 9: V2S_LAUNCH_ATTN(KERNEL, true, false, false, true, 4, grid, dyn, stream, p); break;              \
      case 10: V2S_LAUNCH_ATTN(KERNEL, true, false, true, false, 4, grid, dyn, stream, p); break;             \
      case 11: V2S_LAUNCH_ATTN(KERNEL, true, false, true, true, 4, grid, dyn, stream, p); break;              \
      case 12: V2S_LAUNCH_NC(KERNEL, true, false, false, nc, grid, dyn, stream, p); break;                    \
      case 13: V2S_LAUNCH_NC(KERNEL, true, false, true, nc, grid, dyn, stream, p); break;                     \
      case 14: V2S_LAUNCH_NC(KERNEL, true, true, false, nc, grid, dyn, stream, p); break;                     \
      default: V2S_LAUNCH_NC(KERNEL, true, true, true, nc, grid, dyn, stream, p); break;                      \
    }                                                                                                         \
  } while (0)

}  // namespace

extern "C" int v2s_attn_fwd(const v2s_attn_args* a, void* stream) {
  AttnP p;
  if (int e = fill(p, a, "v2s_attn_fwd", false)) return e;
  if (v2s_opt_fp32_io()) {                // debug mode: fp32 buffers, fp32 arithmetic
    V2S_CHECK(!p.seq_off && !p.kv_seq_off, V2S_ERR_ARG, "v2s_attn_fwd: the fp32_io debug kernels take the dense layout only");
    const size_t dyn = ((size_t)((p.Nk + 3) & ~3) + 4 * HD) * sizeof(float);
    hipLaunchKernelGGL(attn_f32_fwd_kernel, dim3((unsigned)(p.B * p.H * p.Nq)), dim3(256), dyn, (hipStream_t)stream, p);
    V2S_LAUNCH_CHECK();
    return V2S_OK;
  }
  const int grid = ((p.Nq + 127) / 128) * p.H * p.B;
  const bool bias = p.bias_diag != nullptr;
  // bias window: four copies unless they would cost a block per CU (the kernel runs three blocks per CU, two when causal)
  const int nc = (bias && lds_fwd(p, bias, 4) * (p.causal ? 2 : 3) > LDS_PER_CU) ? 2 : 4;
  const size_t dyn = lds_fwd(p, bias, nc);
  V2S_CHECK(dyn <= 160 * 1024, V2S_ERR_SHAPE, "v2s_attn_fwd: Nk=%d too large for the LDS bias window", p.Nk);
  V2S_DISPATCH4(attn_fwd_kernel, v2s_opt_tr_read() != 0, bias, p.causal != 0, p.p16 != 0, nc, dim3(grid), dyn, (hipStream_t)stream, p);
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
  V2S_CHECK(a->delta != nullptr, V2S_ERR_ARG, "v2s_attn_bwd: the row-statistics workspace `delta` (fp32 [B][H][Nq][4]) is missing");
  V2S_CHECK(((uintptr_t)a->delta & 15) == 0, V2S_ERR_ALIGN, "v2s_attn_bwd: `delta` must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (v2s_opt_fp32_io()) {                // debug mode: fp32 buffers, fp32 arithmetic (dbias_diag is accumulated with float atomics)
    V2S_CHECK(!p.seq_off && !p.kv_seq_off, V2S_ERR_ARG, "v2s_attn_bwd: the fp32_io debug kernels take the dense layout only");
    const size_t dyn = ((size_t)2 * ((p.Nk + 3) & ~3) + 4 * HD) * sizeof(float);
    hipLaunchKernelGGL(attn_f32_bwd_dq_kernel, dim3((unsigned)(p.B * p.H * p.Nq)), dim3(256), dyn, s, p);
    V2S_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn_f32_bwd_dkv_kernel, dim3((unsigned)(p.B * p.H * p.Nk)), dim3(256), 0, s, p);
    V2S_LAUNCH_CHECK();
    return V2S_OK;
  }
  const bool tr = v2s_opt_tr_read() != 0, bias = p.bias_diag != nullptr, causal = p.causal != 0, drop = p.p16 != 0;
  const int gq = ((p.Nq + 127) / 128) * p.H * p.B;
  const int nc_q = (bias && lds_dq(p, bias, 4) * ATTN_BWD_OCC > LDS_PER_CU) ? 2 : 4, nc_kv = (bias && lds_dkv(p, bias, 4) * ATTN_DKV_OCC > LDS_PER_CU) ? 2 : 4;   // two blocks per CU
  const size_t dyn_q = lds_dq(p, bias, nc_q), dyn_kv = lds_dkv(p, bias, nc_kv);
  V2S_CHECK(dyn_q <= 160 * 1024 && dyn_kv <= 160 * 1024, V2S_ERR_SHAPE, "v2s_attn_bwd: Nq=%d Nk=%d too large for the LDS bias windows", p.Nq, p.Nk);
  const int part = v2s_opt_attn_bwd_part();      // profiling aid: 1 = dQ kernel only, 2 = dK/dV kernel only (needs the row statistics of an
                                                 // earlier dQ launch with the same arguments); 0 = both
  if (part != 2) {
    V2S_DISPATCH4(attn_bwd_dq_kernel, tr, bias, causal, drop, nc_q, dim3(gq), dyn_q, s, p);
    V2S_LAUNCH_CHECK();
  }
  if (part != 1) {
    const int gk = ((p.Nk + DKV_BK - 1) / DKV_BK) * p.H * p.B;
    V2S_DISPATCH4(attn_bwd_dkv_kernel, tr, bias, causal, drop, nc_kv, dim3(gk), dyn_kv, s, p);
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
