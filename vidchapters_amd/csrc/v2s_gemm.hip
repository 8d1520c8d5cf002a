// MFMA bf16 GEMM family for gfx950 (CDNA4): 128x128x64 block tile, 4 waves (2x2), each wave a 64x64
// sub-tile of 4x4 mfma_f32_16x16x32_bf16 fragments, fp32 accumulation.
//
// Replaces every nn.Linear on the Vid2Seq path and its autograd dgrad/wgrad
// (reference: model/vit.py:41,53,17,20; model/modeling_t5.py:304-311,528-536,581,1714).
//
// Data movement: global -> registers (16 B/lane, issued one K-tile ahead) -> LDS (XOR-swizzled so
// that the ds_read_b128 / ds_read_b64_tr_b16 fragment reads are bank-conflict free) -> MFMA.
// Operands stored with the contraction index NOT contiguous (dgrad's W, wgrad's dY and X) are kept
// in their natural [k][row] order in LDS and transposed on the way to the matrix core with the
// gfx950 LDS transpose read, so no operand is ever re-laid-out in HBM.
// The epilogue goes through LDS so that bias / activation / activation-derivative / dropout /
// residual / accumulate all run on 8-wide row-contiguous vectors with 16-byte global accesses.
#include <type_traits>
#include "v2s_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NTHREADS = 256;
constexpr int STAGE_BYTES = (BM * BK + BN * BK) * 2;  // 32 KiB
constexpr int A_BYTES = BM * BK * 2;                  // 16 KiB

struct GemmP {
  int M, N, K;
  const bf16_t* A; const bf16_t* B;
  long lda, ldb;
  void* C; long ldc;
  int c_f32, accumulate;
  float alpha;
  const float* bias;
  int act;
  bf16_t* pre;
  int dact;
  const bf16_t* z; long ldz;
  const bf16_t* residual; long ldr;
  uint32_t p16; float inv_keep; uint32_t seed;
  const uint32_t* salt;  // device word XOR-ed into seed (v2s_set_seed_salt) or NULL
  int tilesM, tilesN;
  int splitk, kper;      // split-K (wgrad): slice z covers K range [z*kper, min(K,(z+1)*kper)) and writes ws[z][M][N]
  float* ws;
  float rms_eps;         // > 0: fused RMSNorm row scale (skinny kernel only)
  int order;             // tile walk: 0 = row-major with adjacent K slices, GM > 0 = grouped (tile_coords)
  int row0;              // rows of this launch are rows row0.. of the caller's matrix (dropout mask index; v2s_gemm splits rows over two launches)
  int dbg;               // profiling aid (option "gemm_dbg"): 1 = epilogue without post-ops and global store, 2 = no epilogue (128x128 and 8-phase kernels; results invalid)
};

// swizzle of the [k][row] (transposed-operand) LDS image: XOR the 32-byte column chunk with bits of k
__device__ __forceinline__ int tr_g(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }


// work-item id -> (tile row, tile column, K slice).  order 0: slices of one tile adjacent, tiles row-major.  order 1: all tiles of one
// K slice adjacent, and inside a slice a grouped walk (GM tile rows deep, column by column), so that the ~32 blocks an XCD runs at
// any time share a few A row-slabs and B column-slabs in its L2 instead of touching 32 different ones.
__device__ __forceinline__ void tile_coords(const GemmP& p, int id0, int& tm, int& tn, int& slice) {
  if (p.order == 0) {
    slice = id0 % p.splitk;
    const int id = id0 / p.splitk;
    tm = id / p.tilesN; tn = id - tm * p.tilesN;
  } else {
    const int tiles = p.tilesM * p.tilesN;
    slice = id0 / tiles;
    const int t = id0 - slice * tiles;
    const int GM = p.order;
    const int gsz = GM * p.tilesN;
    const int grp = t / gsz, first = grp * GM;
    const int gm = min(p.tilesM - first, GM);
    const int r = t - grp * gsz;
    tm = first + r % gm; tn = r / gm;
  }
}

// ---- global -> register staging -------------------------------------------------------------------
template <bool T>
__device__ __forceinline__ void load_tile(const bf16_t* __restrict__ base, long ld, int row0, int R, int k0,
                                          int K, int tid, uint4 (&r)[4]) {
  if (!T) {  // memory is [row][k]
    const int chunk = tid & 7, rr = tid >> 3;
    const int k = k0 + chunk * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + rr + i * 32;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (row < R && k < K) v = *reinterpret_cast<const uint4*>(base + (long)row * ld + k);
      r[i] = v;
    }
  } else {  // memory is [k][row]
    const int c16 = tid & 15, kk = tid >> 4;
    const int row = row0 + c16 * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + kk + i * 16;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (k < K && row < R) v = *reinterpret_cast<const uint4*>(base + (long)k * ld + row);
      r[i] = v;
    }
  }
}

template <bool T>
__device__ __forceinline__ void store_tile(char* lds, int tid, const uint4 (&r)[4]) {
  if (!T) {
    const int chunk = tid & 7, rr = tid >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = rr + i * 32;
      *reinterpret_cast<uint4*>(lds + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)) = r[i];
    }
  } else {
    const int c16 = tid & 15, kk = tid >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = kk + i * 16;
      *reinterpret_cast<uint4*>(lds + k * 256 + (((c16 >> 1) ^ tr_g(k)) << 5) + ((c16 & 1) << 4)) = r[i];
    }
  }
}

// ---- LDS -> MFMA fragment ------------------------------------------------------------------------------
// fragment of 16 rows x 32 k: lane l holds row (l&15), k = ks*32 + (l>>4)*8 + j, j=0..7
template <bool T, bool TR>
__device__ __forceinline__ bf16x8 read_frag(const char* lds, int rowbase, int ks, int lane) {
  if (!T) {
    const int row = rowbase + (lane & 15);
    const int chunk = ks * 4 + (lane >> 4);
    const uint4 v = *reinterpret_cast<const uint4*>(lds + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
    return __builtin_bit_cast(bf16x8, v);
  } else if (TR) {
    // ds_read_b64_tr_b16: within a 16-lane group, lane i supplies the address of 4 contiguous bf16 of
    // k-row (i>>2) at columns (i&3)*4.., and receives column i of that 4x16 block (4 consecutive k).
    const int i = lane & 15;
    const int k = ks * 32 + (lane >> 4) * 8 + (i >> 2);
    const int m = rowbase + (i & 3) * 4;
    const char* p0 = lds + k * 256 + (((m >> 4) ^ tr_g(k)) << 5) + ((m & 15) << 1);
    const int k1 = k + 4;
    const char* p1 = lds + k1 * 256 + (((m >> 4) ^ tr_g(k1)) << 5) + ((m & 15) << 1);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((V2S_LDS s16x4*)(p0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((V2S_LDS s16x4*)(p1));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  } else {
    const int m = rowbase + (lane & 15);
    s16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = ks * 32 + (lane >> 4) * 8 + j;
      v[j] = *reinterpret_cast<const short*>(lds + k * 256 + (((m >> 4) ^ tr_g(k)) << 5) + ((m & 15) << 1));
    }
    return __builtin_bit_cast(bf16x8, v);
  }
}

// ---- epilogue shared by both main loops: accumulators -> LDS (fp32 [128][128]) -> vector post-ops -> global
// post-ops + store of ONE 8-wide row chunk (gm, gn..gn+7) whose raw accumulators are in v[]
// AHEAD: ``gop`` is the chunk of the epilogue's global operand (z if dact is set, else the residual) fetched ahead by the caller
template <bool AHEAD = false>
__device__ __forceinline__ void epilogue_chunk(const GemmP& p, float (&v)[8], int gm, int gn, int slice, uint4 gop = uint4{0, 0, 0, 0}) {
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] *= p.alpha;
  if (p.bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + gn);
    const float4 b1 = *reinterpret_cast<const float4*>(p.bias + gn + 4);
    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
  }
  if (p.pre) *reinterpret_cast<uint4*>(p.pre + (long)gm * p.ldc + gn) = pack8(v);
  if (p.act == V2S_ACT_RELU) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
  } else if (p.act == V2S_ACT_GELU) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
  }
  if (p.dact != V2S_ACT_NONE) {
    float zf[8];
    uint4 zr;
    if constexpr (AHEAD) zr = gop; else zr = *reinterpret_cast<const uint4*>(p.z + (long)gm * p.ldz + gn);
    unpack8(zr, zf);
    if (p.dact == V2S_ACT_RELU) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = zf[j] > 0.f ? v[j] : 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= dgelu_f(zf[j]);
    }
  }
  if (p.p16) {
    if (p.dact == V2S_ACT_RELU) {
      // z is the forward's dropout(relu(.)) output: z > 0 <=> the unit was active AND kept, so the select above already applied the
      // mask and only the 1/(1-p) scale is left -- no need to regenerate the hash (identical result)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= p.inv_keep;
    } else {
      v2s_drop8(v, (unsigned long long)(gm + p.row0) * (unsigned long long)p.N + gn, v2s_salted(p.seed, p.salt), p.p16, p.inv_keep);
    }
  }
  if (p.residual) {
    float rf[8];
    uint4 rr;
    if (AHEAD && p.dact == V2S_ACT_NONE) rr = gop; else rr = *reinterpret_cast<const uint4*>(p.residual + (long)gm * p.ldr + gn);
    unpack8(rr, rf);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += rf[j];
  }
  if (p.splitk > 1) {     // raw partial sums of this K slice; splitk_reduce_kernel applies alpha and accumulates into C
    float* wp = p.ws + ((long)slice * p.M + gm) * p.N + gn;
    *reinterpret_cast<float4*>(wp) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(wp + 4) = make_float4(v[4], v[5], v[6], v[7]);
    return;
  }
  if (p.c_f32) {
    float* cp = reinterpret_cast<float*>(p.C) + (long)gm * p.ldc + gn;
    float4 o0 = make_float4(v[0], v[1], v[2], v[3]), o1 = make_float4(v[4], v[5], v[6], v[7]);
    if (p.accumulate) {
      const float4 c0 = *reinterpret_cast<const float4*>(cp), c1 = *reinterpret_cast<const float4*>(cp + 4);
      o0.x += c0.x; o0.y += c0.y; o0.z += c0.z; o0.w += c0.w;
      o1.x += c1.x; o1.y += c1.y; o1.z += c1.z; o1.w += c1.w;
    }
    *reinterpret_cast<float4*>(cp) = o0;
    *reinterpret_cast<float4*>(cp + 4) = o1;
  } else {
    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + (long)gm * p.ldc + gn) = pack8(v);
  }
}

// epilogue of the 128x128 kernels: accumulators -> LDS (fp32 [128][128]) -> 8-wide row chunks.  The kernels compute the tile
// TRANSPOSED (mfma(B, A)): lane (g, li) of fragment (i, j) then holds row m = i*16 + li and the four CONSECUTIVE columns
// n = j*16 + 4g .. +3, i.e. one ds_write_b128 per fragment instead of four ds_write_b32 whose four lane groups shared their banks
// (16 instead of 64 LDS stores per lane).  The 16-byte chunks of a row are XOR-swizzled with (m & 7): the eight lanes one
// ds_write_b128 pass serves (eight consecutive rows, same column chunk) and the sixteen a ds_read_b128 pass serves (one row, sixteen
// chunks) then cover all banks.
// The epilogue's ONE global operand (the ReLU / GELU mask source z, or the residual): its eight 16-byte chunks per lane are requested
// before the accumulators are staged, so that their latency runs under the staging and the barrier -- instead of once per write-out
// iteration.  (Requested before the main loop they cost what they save, 59.0 vs 53.8 us on 32000x768x768; requested at the last
// K-step with that step's vmcnt(0) dropped, the changed loop shape costs the main loop 10-30 %.) (dec wo dgrad with the ReLU
// mask 58.6 -> 52.9 us, enc wo fwd with residual + dropout 171 -> 164, enc O fwd 59.1 -> 53.8; tools/gemm_lib_ab.py)
__device__ __forceinline__ bool epilogue_prefetch(const GemmP& p, int m0, int n0, int tid, uint4 (&gop)[8]) {
  const bf16_t* gsrc = p.dact != V2S_ACT_NONE ? p.z : p.residual;
  const long gld = p.dact != V2S_ACT_NONE ? p.ldz : p.ldr;
  const bool ahead = gsrc != nullptr && p.dbg == 0 && p.splitk == 1;
  if (ahead) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int c = tid + it * NTHREADS;
      const int gm = m0 + (c >> 4), gn = n0 + (c & 15) * 8;
      gop[it] = (gm < p.M && gn < p.N) ? *reinterpret_cast<const uint4*>(gsrc + (long)gm * gld + gn) : make_uint4(0, 0, 0, 0);
    }
  }
  return ahead;
}

__device__ __forceinline__ void gemm_epilogue(const GemmP& p, char* smem, const f32x4 (&acc)[4][4], int m0, int n0, int slice,
                                              int tid, int lane, int wm, int wn, const uint4 (&gop)[8], bool ahead) {
  float* cs = reinterpret_cast<float*>(smem);
  if (p.dbg == 2) {                               // ablation (option gemm_dbg): main loop only (accumulators kept live; results invalid)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = wm * 64 + i * 16 + (lane & 15);
      const int c = (wn * 64 + j * 16 + (lane >> 4) * 4) >> 2;            // 16-byte chunk of the row
      *reinterpret_cast<f32x4*>(cs + m * BN + ((c ^ (m & 7)) << 2)) = acc[i][j];
    }
  __syncthreads();
  if (ahead) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int c = tid + it * NTHREADS;
      const int row = c >> 4, cc = (c & 15) * 8;
      const int gm = m0 + row, gn = n0 + cc;
      if (gm >= p.M || gn >= p.N) continue;
      float v[8];
      const int c0 = (cc >> 2) ^ (row & 7), c1 = ((cc >> 2) + 1) ^ (row & 7);
      const float4 x0 = *reinterpret_cast<const float4*>(cs + row * BN + (c0 << 2));
      const float4 x1 = *reinterpret_cast<const float4*>(cs + row * BN + (c1 << 2));
      v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
      epilogue_chunk<true>(p, v, gm, gn, slice, gop[it]);
    }
    return;
  }
#pragma unroll 1
  for (int it = 0; it < 8; ++it) {
    const int c = tid + it * NTHREADS;
    const int row = c >> 4, cc = (c & 15) * 8;
    const int gm = m0 + row, gn = n0 + cc;
    if (gm >= p.M || gn >= p.N) continue;
    float v[8];
    const int c0 = (cc >> 2) ^ (row & 7), c1 = ((cc >> 2) + 1) ^ (row & 7);
    const float4 x0 = *reinterpret_cast<const float4*>(cs + row * BN + (c0 << 2));
    const float4 x1 = *reinterpret_cast<const float4*>(cs + row * BN + (c1 << 2));
    v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
    if (p.dbg == 1) {                             // ablation: everything but the post-ops and the global store
      asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
      continue;
    }
    epilogue_chunk(p, v, gm, gn, slice);
  }
}

template <bool TA, bool TB, bool TR>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(const GemmP p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int nwg = p.tilesM * p.tilesN * p.splitk;
  const int id0 = xcd_remap(blockIdx.x, nwg);
  int tm, tn, slice;
  tile_coords(p, id0, tm, tn, slice);
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = slice * p.kper, kend = min(p.K, kbeg + p.kper);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  const int nk = (kend - kbeg + BK - 1) / BK;
  load_tile<TA>(p.A, p.lda, m0, p.M, kbeg, kend, tid, ra);
  load_tile<TB>(p.B, p.ldb, n0, p.N, kbeg, kend, tid, rb);
  store_tile<TA>(smem, tid, ra);
  store_tile<TB>(smem + A_BYTES, tid, rb);
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    const char* sa = smem + (t & 1) * STAGE_BYTES;
    const char* sb = sa + A_BYTES;
    const bool more = (t + 1 < nk);
    if (more) {
      load_tile<TA>(p.A, p.lda, m0, p.M, kbeg + (t + 1) * BK, kend, tid, ra);
      load_tile<TB>(p.B, p.ldb, n0, p.N, kbeg + (t + 1) * BK, kend, tid, rb);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = read_frag<TA, TR>(sa, wm * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = read_frag<TB, TR>(sb, wn * 64 + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // transposed: see gemm_epilogue
    }
    if (more) {
      char* da = smem + ((t + 1) & 1) * STAGE_BYTES;
      store_tile<TA>(da, tid, ra);
      store_tile<TB>(da + A_BYTES, tid, rb);
    }
    __syncthreads();
  }

  uint4 gop[8];
  const bool ahead = epilogue_prefetch(p, m0, n0, tid, gop);
  gemm_epilogue(p, smem, acc, m0, n0, slice, tid, lane, wm, wn, gop, ahead);
}

// ---- LDS-DMA main loop (K a multiple of 64): tiles go global -> LDS directly (global_load_lds_dwordx4, no VGPR
// staging and no ds_write pass -- the LDS write port was the bottleneck of the register-staged loop).  The DMA writes
// 1 KiB per wave-instruction at (wave-uniform base + lane*16), so the XOR swizzle is applied to the per-lane SOURCE
// address; reads use the same involution (read_frag).  Rows beyond the matrix edge are clamped to the last valid row /
// chunk: they only feed output rows/columns that are never stored.
// The DMA is issued from inline asm on purpose: for the builtin, hipcc cannot prove that the LDS stage being written and
// the stage being read are disjoint and puts s_waitcnt vmcnt(0) in front of the first ds_read of every K-step, which
// serialises load and compute.  From asm the DMA is invisible to the compiler's counters; we wait ourselves
// (dma_wait) right before the barrier that publishes the stage.  M0 carries the wave-uniform LDS byte address.
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)p);
}
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst_uniform) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Per-lane SOURCE byte offset of 1 KiB chunk c of a tile, relative to (matrix base + the K offset of the tile): constant over the
// K loop, so the loop only advances a scalar base (saddr form of the DMA: address = SGPR pair + 32-bit VGPR offset).
template <bool T>
__device__ __forceinline__ uint32_t dma_lane_off(long ld, int row0, int R, int R8, int c, int lane) {
  if (!T) {                             // [128 rows][64 k]: chunk = 8 rows x 128 B
    const int row = c * 8 + (lane >> 3), pos = lane & 7;
    const int chunk = pos ^ ((row >> 1) & 7);
    int grow = row0 + row;
    grow = grow < R ? grow : R - 1;
    return (uint32_t)(((long)grow * ld + chunk * 8) * 2);
  } else {                              // [64 k][128 rows]: chunk = 4 k-rows x 256 B
    const int k = c * 4 + (lane >> 4), pos16 = lane & 15;
    const int c16 = ((((pos16 >> 1) ^ tr_g(k)) << 1) | (pos16 & 1));
    int grow = row0 + c16 * 8;
    grow = grow <= R8 - 8 ? grow : R8 - 8;
    return (uint32_t)(((long)k * ld + grow) * 2);
  }
}
// One wave's share of a stage: 4 x 1 KiB of A and 4 x 1 KiB of B, consecutive chunks (lds = stage base + wave * 4 KiB).  M0 (the
// wave-uniform LDS destination) is saved once, stepped with s_add between the loads and restored once; one wait state is needed
// between an M0 write and the LDS-DMA instruction that reads it.
__device__ __forceinline__ void dma_issue8(const uint32_t (&oa)[4], const uint32_t (&ob)[4], const void* ga, const void* gb, uint32_t lds) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %[keep], m0\n\t"
      "s_mov_b32 m0, %[lds]\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[a0], %[ga]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[a1], %[ga]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[a2], %[ga]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[a3], %[ga]\n\t"
      "s_add_u32 m0, m0, %[skip]\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[b0], %[gb]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[b1], %[gb]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[b2], %[gb]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[b3], %[gb]\n\t"
      "s_mov_b32 m0, %[keep]"
      : [keep] "=&s"(keep)
      : [lds] "s"(lds), [ga] "s"(ga), [gb] "s"(gb), [a0] "v"(oa[0]), [a1] "v"(oa[1]), [a2] "v"(oa[2]), [a3] "v"(oa[3]),
        [b0] "v"(ob[0]), [b1] "v"(ob[1]), [b2] "v"(ob[2]), [b3] "v"(ob[3]), [skip] "n"(A_BYTES - 3 * 1024)
      : "memory", "scc");
}

// main loop of the 128 x 128 LDS-DMA kernels: acc (transposed fragments, see gemm_epilogue) = A[m0.., kbeg..kend) x B[n0.., kbeg..kend)
template <bool TA, bool TB>
__device__ __forceinline__ void gemm_dma_mainloop(const GemmP& p, char* smem, const int m0, const int n0, const int kbeg, const int kend, f32x4 (&acc)[4][4]) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int M8 = (p.M + 7) & ~7, N8 = (p.N + 7) & ~7;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (kend - kbeg) / BK;
  const uint32_t sbase = __builtin_amdgcn_readfirstlane(lds_addr(smem) + wave * 4096);   // this wave's 4 chunks of the A tile of stage 0
  uint32_t oa[4], ob[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    oa[i] = dma_lane_off<TA>(p.lda, m0, p.M, M8, wave * 4 + i, lane);
    ob[i] = dma_lane_off<TB>(p.ldb, n0, p.N, N8, wave * 4 + i, lane);
  }
  const long stepA = TA ? (long)BK * p.lda * 2 : (long)BK * 2, stepB = TB ? (long)BK * p.ldb * 2 : (long)BK * 2;   // bytes per K-step
  const char* ga = reinterpret_cast<const char*>(p.A) + (long)(kbeg / BK) * stepA;
  const char* gb = reinterpret_cast<const char*>(p.B) + (long)(kbeg / BK) * stepB;
  dma_issue8(oa, ob, ga, gb, sbase);
  dma_wait();
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    const char* sa = smem + (t & 1) * STAGE_BYTES;
    const char* sb = sa + A_BYTES;
    if (t + 1 < nk) {
      ga += stepA; gb += stepB;
      dma_issue8(oa, ob, ga, gb, sbase + ((t + 1) & 1) * STAGE_BYTES);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = read_frag<TA, true>(sa, wm * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = read_frag<TB, true>(sb, wn * 64 + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // transposed: see gemm_epilogue
    }
    dma_wait();
    __syncthreads();
  }
}

template <bool TA, bool TB>
__device__ __forceinline__ void gemm_dma_body(const GemmP& p, const int bid, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = p.tilesM * p.tilesN * p.splitk;
  const int id0 = xcd_remap(bid, nwg);
  int tm, tn, slice;
  tile_coords(p, id0, tm, tn, slice);
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = slice * p.kper, kend = min(p.K, kbeg + p.kper);
  f32x4 acc[4][4];
  gemm_dma_mainloop<TA, TB>(p, smem, m0, n0, kbeg, kend, acc);
  uint4 gop[8];
  const bool ahead = epilogue_prefetch(p, m0, n0, tid, gop);
  gemm_epilogue(p, smem, acc, m0, n0, slice, tid, lane, wm, wn, gop, ahead);
}

template <bool TA, bool TB>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_dma_kernel(const GemmP p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
  gemm_dma_body<TA, TB>(p, blockIdx.x, smem);
}

// GROUPED weight gradients (round 4): up to V2S_GEMM_GROUP_MAX problems of ONE shape -- the same projection of the twelve decoder (or
// ViT) layers, whose operands live in separate allocations -- in one launch: block -> (problem, tile).  A 768 x 768 x 8192 weight gradient
// has 36 output tiles; alone it needs split-K (plus a reduce launch) to occupy the chip and runs at 370 TF/s, twelve of them fill it
// with whole-K tiles and no reduction (tools/gemm_group_ab.py).  The pointers travel in the kernel arguments.
constexpr int GEMM_GROUP_MAX = 16;
struct GemmGrp {
  GemmP p;
  const bf16_t* A[GEMM_GROUP_MAX];
  const bf16_t* B[GEMM_GROUP_MAX];
  void* C[GEMM_GROUP_MAX];
};
__global__ __launch_bounds__(NTHREADS, 2) void gemm_dma_grouped_kernel(const GemmGrp g) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
  const int per = g.p.tilesM * g.p.tilesN * g.p.splitk;
  const int gi = __builtin_amdgcn_readfirstlane((int)blockIdx.x / per);
  GemmP p = g.p;
  p.A = g.A[gi]; p.B = g.B[gi]; p.C = g.C[gi];
  gemm_dma_body<true, true>(p, (int)blockIdx.x - gi * per, smem);
}

// ---- tied LM head + label-smoothed cross entropy WITHOUT logits in memory (round 6; model/modeling_t5.py:1709-1721) ------------------------
// logits[r][c] = alpha * h[r] . E[c] are computed tile by tile by the 128 x 128 LDS-DMA main loop above and consumed in the epilogue:
//   MODE 0 (forward): each 128 x 128 tile is reduced to per-row partial statistics over its two 64-column halves -- (max, sum exp(v - max), sum v,
//          the target's logit if it lies there) -- one float4 per (row, half tile); lmhead_finish_kernel merges a row's halves
//          into its log-sum-exp and its smoothed loss (the same two numbers v2s_ce_fwd leaves per row);
//   MODE 1 (backward): the tile is RECOMPUTED and turned into d(logits) = (softmax - (1 - eps) onehot - eps / V) * gscale, stored as bf16
//          for the two GEMMs that consume it (d(hidden) = d(logits) E, d(E) += d(logits)^T h).
// N of the GemmP is the vocabulary padded to the tile (the arena keeps zero rows behind the embedding); columns >= V are excluded from the
// statistics and written as zeros.
struct HeadX {
  int V, tilesN;
  const long* labels;
  float4* part;          // MODE 0: [rows][2 * tilesN] (one entry per 64-column half of a column tile)
  const float* row;      // MODE 1: [rows][2] (log-sum-exp, loss) from lmhead_finish_kernel
  float eps;
  const float* gscale;   // MODE 1: device scalar d(total loss) / d(sum of row losses)
  bf16_t* dl; long ldd;  // MODE 1: d(logits) [rows][ldd]
};

template <int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void lmhead_ce_kernel(const GemmP p, const HeadX x) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int id0 = xcd_remap(blockIdx.x, p.tilesM * p.tilesN);
  int tm, tn, slice;
  tile_coords(p, id0, tm, tn, slice);
  const int m0 = tm * BM, n0 = tn * BN;
  f32x4 acc[4][4];
  gemm_dma_mainloop<false, false>(p, smem, m0, n0, 0, p.K, acc);
  // fragment (i, j) of a wave: lane (g = lane >> 4, li = lane & 15) holds row wm*64 + i*16 + li, columns wn*64 + j*16 + 4g .. +3 (see gemm_epilogue)
  const int g4 = (lane >> 4) * 4, li = lane & 15;
  if (MODE == 0) {
    // statistics straight from the accumulators -- no LDS staging, no barrier: a lane reduces its 16 columns of a row, the four lane groups that
    // share the row are merged with two xor-shuffles, and each wave writes ITS 64-column half: part[row][2 * tn + wn]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gm = m0 + wm * 64 + i * 16 + li;
      const bool live = gm < p.M;
      const int y = live ? (int)x.labels[gm] : -1;
      float mx = -INFINITY, tot = 0.f, tg = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int gn = n0 + wn * 64 + j * 16 + g4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[i][j][r] * p.alpha;
          acc[i][j][r] = v;
          const bool in = gn + r < x.V;
          mx = in ? fmaxf(mx, v) : mx;
          tot += in ? v : 0.f;
          tg += (gn + r == y) ? v : 0.f;
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mref = mx == -INFINITY ? 0.f : mx;          // (a half tile entirely beyond V: no valid column, sum 0)
      float se = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int gn = n0 + wn * 64 + j * 16 + g4;
#pragma unroll
        for (int r = 0; r < 4; ++r) se += (gn + r < x.V) ? __expf(acc[i][j][r] - mref) : 0.f;
      }
      se += __shfl_xor(se, 16, 64); tot += __shfl_xor(tot, 16, 64); tg += __shfl_xor(tg, 16, 64);
      se += __shfl_xor(se, 32, 64); tot += __shfl_xor(tot, 32, 64); tg += __shfl_xor(tg, 32, 64);
      if (live && lane < 16) x.part[(long)gm * (2 * x.tilesN) + 2 * tn + wn] = make_float4(mx, se, tot, tg);
    }
  } else {
    // d(logits) in registers, rounded to bf16, staged through LDS as a bf16 [128][128] tile (16-byte chunks XOR-swizzled with (row & 7)) so that the
    // global stores are whole 256-byte row segments
    const float sm = x.eps / x.V, g = x.gscale[0], keep = 1.f - x.eps;
    bf16_t* cs = reinterpret_cast<bf16_t*>(smem);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wm * 64 + i * 16 + li, gm = m0 + row;
      const bool live = gm < p.M;
      const int y = live ? (int)x.labels[gm] : -1;
      const float lse = y >= 0 ? x.row[(long)gm * 2] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cn = wn * 64 + j * 16 + g4, gn = n0 + cn;
        float f[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pr = __expf(acc[i][j][r] * p.alpha - lse) - sm;
          if (gn + r == y) pr -= keep;
          f[r] = (y >= 0 && gn + r < x.V) ? pr * g : 0.f;
        }
        const int c8 = cn >> 3;                                // 16-byte chunk of the row; this lane fills half of it
        *reinterpret_cast<uint2*>(cs + row * BN + ((c8 ^ (row & 7)) << 3) + (cn & 4)) = make_uint2(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]));
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int c = tid + it * NTHREADS;                        // 16 consecutive lanes = the 16 chunks of one row
      const int row = c >> 4, c8 = c & 15;
      const int gm = m0 + row, gn = n0 + c8 * 8;
      if (gm >= p.M || gn >= x.ldd) continue;
      *reinterpret_cast<uint4*>(x.dl + (long)gm * x.ldd + gn) = *reinterpret_cast<const uint4*>(cs + row * BN + ((c8 ^ (row & 7)) << 3));
    }
  }
}

// one wave per row: merge the (max, sum exp, sum, target) partials of the row's column tiles -> row_out[row] = (log-sum-exp, smoothed loss)
__global__ __launch_bounds__(256) void lmhead_finish_kernel(const float4* __restrict__ part, const long* __restrict__ labels, int rows, int tilesN, int V,
                                                            float eps, float* __restrict__ row_out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const long y = labels[row];
  if (y < 0) {
    if (lane == 0) { row_out[row * 2] = 0.f; row_out[row * 2 + 1] = 0.f; }
    return;
  }
  float m = -INFINITY, s = 0.f, tot = 0.f, tg = 0.f;
  for (int t = lane; t < 2 * tilesN; t += 64) {          // two 64-column halves per column tile
    const float4 q = part[(long)row * (2 * tilesN) + t];
    if (q.x == -INFINITY) continue;                       // a half entirely beyond V
    const float mn = fmaxf(m, q.x);
    s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + q.y * __expf(q.x - mn);
    m = mn; tot += q.z; tg += q.w;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    const float mn = fmaxf(m, m2);
    s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mn));
    m = mn;
    tot += __shfl_xor(tot, o, 64); tg += __shfl_xor(tg, o, 64);
  }
  if (lane == 0) {
    const float lse = m + logf(s);
    row_out[row * 2] = lse;
    row_out[row * 2 + 1] = (1.f - eps) * (lse - tg) + eps * (lse - tot / V);
  }
}

// =====================================================================================================================
// 256 x BN2 x 64 tile kernel (BN2 = 256 or 128), 8 waves, LDS-DMA double buffer (2 x 64 / 2 x 48 KiB), one block per CU.
// Twice (1.33x) the flops per staged byte of the 128x128 kernels: those are limited by bytes-in-flight x latency, not by
// the matrix pipe.  Wave grid 2(M) x 4(N) with 128x64 per wave (BN2=256) or 4 x 2 with 64x64 (BN2=128).
// Transposed operands use the same [k][row] LDS image as the small kernel with a row pitch of ROWS*2 bytes.
template <int ROWS, bool T>
__device__ __forceinline__ void dma_tile_big(const bf16_t* __restrict__ base, long ld, int row0, int R, int R8, int k0,
                                             uint32_t tile_addr, int wave, int lane) {
  constexpr int NCHUNK = ROWS * 64 * 2 / 1024;     // 1 KiB chunks in the tile (32 or 16)
  constexpr int PER_WAVE = NCHUNK / 8;
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int c = wave * PER_WAVE + i;
    const bf16_t* src;
    if (!T) {                             // [ROWS][64 k]: chunk = 8 rows x 128 B
      const int row = c * 8 + (lane >> 3), pos = lane & 7;
      const int chunk = pos ^ ((row >> 1) & 7);
      int grow = row0 + row;
      grow = grow < R ? grow : R - 1;
      src = base + (long)grow * ld + k0 + chunk * 8;
    } else {                              // [64 k][ROWS]: k-row pitch ROWS*2 bytes
      constexpr int SLOTS = ROWS / 8;     // 16-byte slots per k-row (32 or 16)
      constexpr int KPC = 64 / SLOTS;     // k-rows per 1 KiB chunk (2 or 4)
      const int k = c * KPC + lane / SLOTS, pos16 = lane % SLOTS;
      const int c16 = ((((pos16 >> 1) ^ tr_g(k)) << 1) | (pos16 & 1));
      int grow = row0 + c16 * 8;
      grow = grow <= R8 - 8 ? grow : R8 - 8;
      src = base + (long)(k0 + k) * ld + grow;
    }
    glds16(src, __builtin_amdgcn_readfirstlane(tile_addr + c * 1024));
  }
}

template <int ROWS, bool T>
__device__ __forceinline__ bf16x8 read_frag_big(const char* lds, int rowbase, int ks, int lane) {
  if (!T) {
    const int row = rowbase + (lane & 15);
    const int chunk = ks * 4 + (lane >> 4);
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)));
  } else {
    constexpr int PITCH = ROWS * 2;
    const int i = lane & 15;
    const int k = ks * 32 + (lane >> 4) * 8 + (i >> 2);
    const int m = rowbase + (i & 3) * 4;
    const int k1 = k + 4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((V2S_LDS s16x4*)(lds + k * PITCH + (((m >> 4) ^ tr_g(k)) << 5) + ((m & 15) << 1)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((V2S_LDS s16x4*)(lds + k1 * PITCH + (((m >> 4) ^ tr_g(k1)) << 5) + ((m & 15) << 1)));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
}

template <bool TA, bool TB, int BN2>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(const GemmP p) {
  constexpr int BM2 = 256;
  constexpr int A_B = BM2 * 64 * 2, B_B = BN2 * 64 * 2, STG = A_B + B_B;
  constexpr int WN = BN2 / 64, WM = 8 / WN;          // 4x2 or 2x4 ... (WM x WN) waves
  constexpr int MI = BM2 / WM / 16;                  // 16-row fragments per wave: 8 (BN2=256) or 4 (BN2=128)
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 * STG bytes
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int nwg = p.tilesM * p.tilesN * p.splitk;
  const int id0 = xcd_remap(blockIdx.x, nwg);
  int tm, tn, slice;
  tile_coords(p, id0, tm, tn, slice);
  const int m0 = tm * BM2, n0 = tn * BN2;
  const int kbeg = slice * p.kper, kend = min(p.K, kbeg + p.kper);
  const int M8 = (p.M + 7) & ~7, N8 = (p.N + 7) & ~7;

  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (kend - kbeg) / BK;
  const uint32_t sbase = lds_addr(smem);
  dma_tile_big<BM2, TA>(p.A, p.lda, m0, p.M, M8, kbeg, sbase, wave, lane);
  dma_tile_big<BN2, TB>(p.B, p.ldb, n0, p.N, N8, kbeg, sbase + A_B, wave, lane);
  dma_wait();
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    const char* sa = smem + (t & 1) * STG;
    const char* sb = sa + A_B;
    if (t + 1 < nk) {
      const uint32_t da = sbase + ((t + 1) & 1) * STG;
      dma_tile_big<BM2, TA>(p.A, p.lda, m0, p.M, M8, kbeg + (t + 1) * BK, da, wave, lane);
      dma_tile_big<BN2, TB>(p.B, p.ldb, n0, p.N, N8, kbeg + (t + 1) * BK, da + A_B, wave, lane);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 bfr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = read_frag_big<BN2, TB>(sb, wn * 64 + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const bf16x8 af = read_frag_big<BM2, TA>(sa, wm * (MI * 16) + i * 16, ks, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[j], acc[i][j], 0, 0, 0);
      }
    }
    dma_wait();
    __syncthreads();
  }

  // epilogue in passes of 64 tile rows (fp32 staging 64 x BN2 <= 64 KiB): pass ps takes fragments i in [ps*IPP, (ps+1)*IPP)
  float* cs = reinterpret_cast<float*>(smem);
  constexpr int IPP = 64 / WM / 16;                 // fragments per wave per pass: 2 (WM=2) or 1 (WM=4)
  constexpr int NPASS = MI / IPP;                   // 4
  constexpr int CPR = BN2 / 8;                      // 8-wide chunks per row
  constexpr int NIT = 64 * CPR / 512;               // 8-wide chunks per lane per pass: 4 | 2
  // one global operand (ReLU / GELU mask source, or the residual): the chunks of pass ps + 1 are requested before the write-out of
  // pass ps (those of pass 0 before the first staging), see gemm_epilogue
  const bf16_t* gsrc = p.dact != V2S_ACT_NONE ? p.z : p.residual;
  const long gld = p.dact != V2S_ACT_NONE ? p.ldz : p.ldr;
  const bool ahead = gsrc != nullptr;
  uint4 gop[NIT], gnext[NIT];
  auto fetch = [&](int ps, uint4 (&g)[NIT]) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = tid + it * 512;
      const int lr = c / CPR, cc = (c % CPR) * 8;
      const int gm = m0 + (lr / (IPP * 16)) * (MI * 16) + ps * (IPP * 16) + lr % (IPP * 16), gn = n0 + cc;
      g[it] = (gm < p.M && gn < p.N) ? *reinterpret_cast<const uint4*>(gsrc + (long)gm * gld + gn) : make_uint4(0, 0, 0, 0);
    }
  };
  if (ahead) fetch(0, gop);
#pragma clang loop unroll(full)                     // acc[] is indexed by ps: a rolled loop would put the accumulators in scratch
  for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
    for (int ii = 0; ii < IPP; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          cs[(wm * (IPP * 16) + ii * 16 + (lane >> 4) * 4 + r) * BN2 + wn * 64 + j * 16 + (lane & 15)] = acc[ps * IPP + ii][j][r];
    __syncthreads();
    if (ahead) {
      if (ps + 1 < NPASS) fetch(ps + 1, gnext);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = tid + it * 512;
        const int lr = c / CPR, cc = (c % CPR) * 8;
        const int gm = m0 + (lr / (IPP * 16)) * (MI * 16) + ps * (IPP * 16) + lr % (IPP * 16), gn = n0 + cc;
        if (gm < p.M && gn < p.N) {
          float v[8];
          const float4 x0 = *reinterpret_cast<const float4*>(cs + lr * BN2 + cc);
          const float4 x1 = *reinterpret_cast<const float4*>(cs + lr * BN2 + cc + 4);
          v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
          epilogue_chunk<true>(p, v, gm, gn, slice, gop[it]);
        }
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) gop[it] = gnext[it];
    } else {
#pragma unroll 1
      for (int c = tid; c < 64 * CPR; c += 512) {
        const int lr = c / CPR, cc = (c % CPR) * 8;                 // local row in the 64-row staging block
        const int w = lr / (IPP * 16), rr = lr % (IPP * 16);        // owning wave row, row inside its IPP*16 rows
        const int gm = m0 + w * (MI * 16) + ps * (IPP * 16) + rr, gn = n0 + cc;
        if (gm >= p.M || gn >= p.N) continue;
        float v[8];
        const float4 x0 = *reinterpret_cast<const float4*>(cs + lr * BN2 + cc);
        const float4 x1 = *reinterpret_cast<const float4*>(cs + lr * BN2 + cc + 4);
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
        epilogue_chunk(p, v, gm, gn, slice);
      }
    }
    __syncthreads();
  }
}


// =====================================================================================================================
// 256 x 128 x 32 tile kernel, 4 waves (2 x 2, 128 x 64 per wave), 3-stage LDS-DMA ring (3 x 24 KiB), TWO blocks per CU.
// The K=768 GEMMs of the step spend as long in prologue + epilogue as in the main loop; with one block per CU nothing runs on
// the matrix pipe meanwhile.  Two co-resident blocks overlap one block's epilogue / first loads with the other's MFMAs while
// keeping the 128x64 per-wave register tile (12 fragment reads per 32 MFMAs) of the 8-wave kernel.  Prefetch distance is two
// K-steps: the wait before the barrier leaves the newest stage in flight (s_waitcnt vmcnt(PER)).
// LDS image of a K-contiguous operand: [row][32 k] = 64 B rows, 16-byte slot g of row r stored at slot g ^ 2*((r>>2)&1): with
// ds_read_b128's lane groups ({0-3,12-15,20-27}, ...) every group then covers all 64 banks exactly once.
template <int N>
__device__ __forceinline__ void dma_wait_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int ROWS, bool T>
__device__ __forceinline__ void dma_tile_w4(const bf16_t* __restrict__ base, long ld, int row0, int R, int R8, int k0,
                                            uint32_t tile_addr, int wave, int lane) {
  constexpr int NCHUNK = ROWS * 32 * 2 / 1024;     // 1 KiB chunks in the tile (16 or 8)
  constexpr int PER_WAVE = NCHUNK / 4;
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int c = wave * PER_WAVE + i;
    const bf16_t* src;
    if (!T) {                             // chunk = 16 rows x 64 B
      const int row = c * 16 + (lane >> 2), pos = lane & 3;
      const int g = pos ^ (((row >> 2) & 1) << 1);
      int grow = row0 + row;
      grow = grow < R ? grow : R - 1;
      src = base + (long)grow * ld + k0 + g * 8;
    } else {                              // [32 k][ROWS]: k-row pitch ROWS*2 bytes
      constexpr int SLOTS = ROWS / 8;
      constexpr int KPC = 64 / SLOTS;
      const int k = c * KPC + lane / SLOTS, pos16 = lane % SLOTS;
      const int c16 = ((((pos16 >> 1) ^ tr_g(k)) << 1) | (pos16 & 1));
      int grow = row0 + c16 * 8;
      grow = grow <= R8 - 8 ? grow : R8 - 8;
      src = base + (long)(k0 + k) * ld + grow;
    }
    glds16(src, __builtin_amdgcn_readfirstlane(tile_addr + c * 1024));
  }
}

template <int ROWS, bool T>
__device__ __forceinline__ bf16x8 read_frag_w4(const char* lds, int rowbase, int lane) {
  if (!T) {
    const int r = lane & 15, g = lane >> 4;
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + (rowbase + r) * 64 + ((g ^ (((r >> 2) & 1) << 1)) << 4)));
  } else {
    return read_frag_big<ROWS, true>(lds, rowbase, 0, lane);
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_w4_kernel(const GemmP p) {
  constexpr int BM2 = 256, BN2 = 128, KT = 32, NST = 3;
  constexpr int A_B = BM2 * KT * 2, B_B = BN2 * KT * 2, STG = A_B + B_B;      // 16 + 8 KiB
  constexpr int PER = (BM2 + BN2) * KT * 2 / 1024 / 4;                          // DMA instructions per wave per stage (6)
  extern __shared__ __attribute__((aligned(16))) char smem[];   // NST * STG = 72 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = p.tilesM * p.tilesN * p.splitk;
  const int id0 = xcd_remap(blockIdx.x, nwg);
  int tm, tn, slice;
  tile_coords(p, id0, tm, tn, slice);
  const int m0 = tm * BM2, n0 = tn * BN2;
  const int kbeg = slice * p.kper, kend = min(p.K, kbeg + p.kper);
  const int M8 = (p.M + 7) & ~7, N8 = (p.N + 7) & ~7;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (kend - kbeg) / KT;
  const uint32_t sbase = lds_addr(smem);
  dma_tile_w4<BM2, TA>(p.A, p.lda, m0, p.M, M8, kbeg, sbase, wave, lane);
  dma_tile_w4<BN2, TB>(p.B, p.ldb, n0, p.N, N8, kbeg, sbase + A_B, wave, lane);
  if (nk > 1) {
    dma_tile_w4<BM2, TA>(p.A, p.lda, m0, p.M, M8, kbeg + KT, sbase + STG, wave, lane);
    dma_tile_w4<BN2, TB>(p.B, p.ldb, n0, p.N, N8, kbeg + KT, sbase + STG + A_B, wave, lane);
  }
  int st = 0, st2 = 2;                   // stage being computed, stage being filled (t + 2)
  for (int t = 0; t < nk; ++t) {
    if (t + 1 < nk) dma_wait_n<PER>(); else dma_wait_n<0>();
    __syncthreads();
    if (t + 2 < nk) {
      const uint32_t da = sbase + st2 * STG;
      dma_tile_w4<BM2, TA>(p.A, p.lda, m0, p.M, M8, kbeg + (t + 2) * KT, da, wave, lane);
      dma_tile_w4<BN2, TB>(p.B, p.ldb, n0, p.N, N8, kbeg + (t + 2) * KT, da + A_B, wave, lane);
    }
    const char* sa = smem + st * STG;
    const char* sb = sa + A_B;
    bf16x8 bfr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = read_frag_w4<BN2, TB>(sb, wn * 64 + j * 16, lane);
    bf16x8 af = read_frag_w4<BM2, TA>(sa, wm * 128, lane);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      bf16x8 afn = af;
      if (i + 1 < 8) afn = read_frag_w4<BM2, TA>(sa, wm * 128 + (i + 1) * 16, lane);     // one fragment ahead of the MFMAs
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[j], acc[i][j], 0, 0, 0);
      af = afn;
    }
    st = st == NST - 1 ? 0 : st + 1;
    st2 = st2 == NST - 1 ? 0 : st2 + 1;
  }
  __syncthreads();

  // epilogue: two passes of 128 tile rows through a 64 KiB fp32 staging block (pass ps takes fragments i in [4 ps, 4 ps + 4))
  float* cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          cs[(wm * 64 + ii * 16 + (lane >> 4) * 4 + r) * BN2 + wn * 64 + j * 16 + (lane & 15)] = acc[ps * 4 + ii][j][r];
    __syncthreads();
#pragma unroll 1
    for (int c = tid; c < 128 * 16; c += 256) {
      const int lr = c >> 4, cc = (c & 15) * 8;
      const int gm = m0 + (lr >> 6) * 128 + ps * 64 + (lr & 63), gn = n0 + cc;
      if (gm >= p.M || gn >= p.N) continue;
      float v[8];
      const float4 x0 = *reinterpret_cast<const float4*>(cs + lr * BN2 + cc);
      const float4 x1 = *reinterpret_cast<const float4*>(cs + lr * BN2 + cc + 4);
      v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
      epilogue_chunk(p, v, gm, gn, slice);
    }
    if (ps == 0) __syncthreads();
  }
}



// =====================================================================================================================
// 8-phase "ping-pong" kernel: 256 x BN2 output tile (BN2 = 256 | 128), 8 waves, K walked in stages of 32 through an LDS-DMA ring
// (4 x 32 KiB | 6 x 24 KiB) with COUNTED s_waitcnt vmcnt(N) -- loads stay in flight across barriers -- and the two halves of the
// block (waves 0-3 / 4-7: the two waves every SIMD holds) running half a phase apart: while one half issues its 16 MFMAs
// (s_setprio 1) the other half reads its fragments from LDS and issues the DMA of a later stage, so the matrix pipe of every SIMD
// always has a wave that is in its MFMA half-phase.  The one-barrier-per-K-step kernels above leave the pipe idle while both waves
// of a SIMD wait for the same barrier / the same vmcnt(0) (PMC r01: MFMA busy 36-59 %).
//
// Schedule (phase p = 1, 2, ..., "mem" = ds_reads + DMA issue + counted wait, "mfma" = 16 MFMAs; B0, B1, ... = s_barrier events):
//   half 0:        mem(1) B0 mfma(1) B1 mem(2) B2 mfma(2) B3 ...
//   half 1:  (B0)  mem(1) B1 mfma(1) B2 mem(2) B3 mfma(2) ...         one extra barrier up front, half 0 has one extra at the end
// LDS hazards under that stagger (derivation in DESIGN.md):
//   WAR  a ring slot last READ in phase q may be re-filled by a DMA issued in phase >= q + 2
//   RAW  a stage whose counted wait sits in mem(p') may be read in phase >= p' + 1
// BN2 = 256: wave tile 128 x 64 (8 x 4 fragments), 2 phases per stage; phase 2s+1 issues B(s+2), phase 2s+2 issues A(s+3) and
//            waits for stage s+1 with vmcnt(6) (A(s+2), B(s+2), A(s+3) stay in flight: four phases of flight time per stage).
// BN2 = 128: wave tile 64 x 64 (4 x 4 fragments), 1 phase per stage; phase s+1 issues stage s+4 and waits for stage s+1 with vmcnt(9).
// The accumulators are kept TRANSPOSED (mfma(B fragment, A fragment)): a lane then owns 4 consecutive columns of one row per
// fragment, so the epilogue stages through LDS with one ds_write_b128 per fragment instead of four ds_write_b32.
template <int BN2>
struct P8 {
  static constexpr int BM2 = 256, KT = 32;
  static constexpr int WN = BN2 / 64, WM = 8 / WN;          // 2 x 4 | 4 x 2 waves
  static constexpr int MI = BM2 / WM / 16;                  // 16-row fragments per wave: 8 | 4
  static constexpr int PPS = MI / 4;                        // phases per stage: 2 | 1
  static constexpr int A_B = BM2 * KT * 2, B_B = BN2 * KT * 2, STG = A_B + B_B;
  static constexpr int NST = BN2 == 256 ? 4 : 6;            // ring depth: 128 | 144 KiB
  static constexpr int DA = A_B / 1024 / 8, DB = B_B / 1024 / 8;      // DMA instructions per wave: A part 2, B part 2 | 1
  static constexpr int LDS_BYTES = NST * STG;
};

// per-lane SOURCE byte offset of 1 KiB chunk c of a [ROWS][32 k] stage image (same image as dma_tile_w4 / read_frag_w4), relative
// to (matrix base + the K offset of the stage): constant over the K loop, the loop advances a scalar base (saddr DMA form)
template <int ROWS, bool T>
__device__ __forceinline__ uint32_t p8_lane_off(long ld, int row0, int R, int R8, int c, int lane) {
  if (!T) {                             // chunk = 16 rows x 64 B
    const int row = c * 16 + (lane >> 2), pos = lane & 3;
    const int g = pos ^ (((row >> 2) & 1) << 1);
    int grow = row0 + row;
    grow = grow < R ? grow : R - 1;
    return (uint32_t)(((long)grow * ld + g * 8) * 2);
  } else {                              // [32 k][ROWS]: k-row pitch ROWS*2 bytes
    constexpr int SLOTS = ROWS / 8, KPC = 64 / SLOTS;
    const int k = c * KPC + lane / SLOTS, pos16 = lane % SLOTS;
    const int c16 = ((((pos16 >> 1) ^ tr_g(k)) << 1) | (pos16 & 1));
    int grow = row0 + c16 * 8;
    grow = grow <= R8 - 8 ? grow : R8 - 8;
    return (uint32_t)(((long)k * ld + grow) * 2);
  }
}
__device__ __forceinline__ void p8_dma2(uint32_t o0, uint32_t o1, const void* g, uint32_t lds) {
  uint32_t keep;
  asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o0], %[g]\n\t"
               "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o1], %[g]\n\ts_mov_b32 m0, %[keep]"
               : [keep] "=&s"(keep) : [lds] "s"(lds), [g] "s"(g), [o0] "v"(o0), [o1] "v"(o1) : "memory", "scc");
}
__device__ __forceinline__ void p8_dma1(uint32_t o0, const void* g, uint32_t lds) {
  uint32_t keep;
  asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o0], %[g]\n\ts_mov_b32 m0, %[keep]"
               : [keep] "=&s"(keep) : [lds] "s"(lds), [g] "s"(g), [o0] "v"(o0) : "memory", "scc");
}
#define P8_BARRIER()                      \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)

template <bool TA, bool TB, int BN2>
__global__ __launch_bounds__(512, 2) void gemm_p8_kernel(const GemmP p) {
  using G = P8<BN2>;
  constexpr int BM2 = G::BM2, WN = G::WN, MI = G::MI, PPS = G::PPS, A_B = G::A_B, STG = G::STG, NST = G::NST, DA = G::DA, DB = G::DB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = wave >> 2;                     // waves w and w + 4 share a SIMD: they must be in different halves
  const int nwg = p.tilesM * p.tilesN * p.splitk;
  const int id0 = xcd_remap(blockIdx.x, nwg);
  int tm, tn, slice;
  tile_coords(p, id0, tm, tn, slice);
  const int m0 = tm * BM2, n0 = tn * BN2;
  const int kbeg = slice * p.kper, kend = min(p.K, kbeg + p.kper);
  const int M8 = (p.M + 7) & ~7, N8 = (p.N + 7) & ~7;
  const int nst = (kend - kbeg) / 32;             // >= 4 (dispatcher)

  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint32_t oa[DA], ob[DB];
#pragma unroll
  for (int i = 0; i < DA; ++i) oa[i] = p8_lane_off<BM2, TA>(p.lda, m0, p.M, M8, wave * DA + i, lane);
#pragma unroll
  for (int i = 0; i < DB; ++i) ob[i] = p8_lane_off<BN2, TB>(p.ldb, n0, p.N, N8, wave * DB + i, lane);
  const long stepA = TA ? 32L * p.lda * 2 : 64L, stepB = TB ? 32L * p.ldb * 2 : 64L;        // bytes per stage
  const char* ga = reinterpret_cast<const char*>(p.A) + (long)(kbeg / 32) * stepA;          // source of the next A part to issue
  const char* gb = reinterpret_cast<const char*>(p.B) + (long)(kbeg / 32) * stepB;
  const uint32_t sbase = lds_addr(smem);
  const uint32_t dstA = __builtin_amdgcn_readfirstlane(sbase + wave * DA * 1024);
  const uint32_t dstB = __builtin_amdgcn_readfirstlane(sbase + A_B + wave * DB * 1024);
  int slotA = 0, slotB = 0;                       // ring slot (byte offset) of the next A / B part to issue
  auto issueA = [&]() {
    p8_dma2(oa[0], oa[1], ga, dstA + slotA);
    ga += stepA; slotA = slotA + STG == NST * STG ? 0 : slotA + STG;
  };
  auto issueB = [&]() {
    if constexpr (DB == 2) p8_dma2(ob[0], ob[1], gb, dstB + slotB); else p8_dma1(ob[0], gb, dstB + slotB);
    gb += stepB; slotB = slotB + STG == NST * STG ? 0 : slotB + STG;
  };

  bf16x8 af[4], bfr[4];
  int rslot = 0;                                  // ring slot (byte offset) of the stage being read
  const int arow = wm * (MI * 16), brow = wn * 64;

  if constexpr (PPS == 2) {
    issueA(); issueB(); issueA(); issueB(); issueA();           // A0 B0 A1 B1 A2   (nst >= 4)
    dma_wait_n<6>();                                            // stage 0 landed; A1 B1 A2 in flight
  } else {
    issueA(); issueB(); issueA(); issueB(); issueA(); issueB(); issueA(); issueB();     // stages 0..3
    dma_wait_n<9>();
  }
  P8_BARRIER();
  if (half == 1) P8_BARRIER();

  for (int s = 0; s < nst; ++s) {
    const char* sa = smem + rslot;
    const char* sb = sa + A_B;
    if constexpr (PPS == 2) {
      // ---- phase 2s+1: fragments 0-3 of A, all of B; issue B(s+2)
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = read_frag_w4<BN2, TB>(sb, brow + j * 16, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = read_frag_w4<BM2, TA>(sa, arow + i * 16, lane);
      if (s + 2 < nst) issueB();
      P8_BARRIER();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      P8_BARRIER();
      // ---- phase 2s+2: fragments 4-7 of A; issue A(s+3); stage s+1 must have landed before the next phase reads it
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = read_frag_w4<BM2, TA>(sa, arow + 64 + i * 16, lane);
      if (s + 3 < nst) { issueA(); dma_wait_n<6>(); }
      else if (s + 2 < nst) dma_wait_n<4>();
      else dma_wait_n<0>();
      P8_BARRIER();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[4 + i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      P8_BARRIER();
    } else {
      // ---- phase s+1: the whole stage; issue stage s+4; stage s+1 must have landed before the next phase reads it
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = read_frag_w4<BN2, TB>(sb, brow + j * 16, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = read_frag_w4<BM2, TA>(sa, arow + i * 16, lane);
      if (s + 4 < nst) { issueA(); issueB(); dma_wait_n<9>(); }
      else if (s + 3 < nst) dma_wait_n<6>();
      else if (s + 2 < nst) dma_wait_n<3>();
      else dma_wait_n<0>();
      P8_BARRIER();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      P8_BARRIER();
    }
    rslot = rslot + STG == NST * STG ? 0 : rslot + STG;
  }
  if (half == 0) P8_BARRIER();                    // re-align the two halves: every MFMA and every LDS read of the block is done

  // epilogue: passes of WM*32 tile rows through an fp32 staging block [WM*32][BN2 + 4] (pitch 65 x 16 B: the eight lanes of a
  // ds_write_b128 group hit eight different 16-byte bank slots); transposed accumulators: lane = row (lane & 15) of fragment i,
  // columns 4*(lane >> 4) .. +3 of fragment j
  constexpr int WMc = G::WM, PB = BN2 + 4, NPASS = MI / 2, CPR = BN2 / 8, ROWS_PASS = WMc * 32;
  float* cs = reinterpret_cast<float*>(smem);
  if (p.dbg == 2) {                               // ablation: main loop only (accumulators kept live)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  // one global operand (ReLU / GELU mask source, or the residual): the chunks of pass ps + 1 are requested before the write-out of
  // pass ps (those of pass 0 before the first staging), see gemm_epilogue
  constexpr int NIT = ROWS_PASS * CPR / 512;        // 8-wide chunks per lane per pass
  const bf16_t* gsrc = p.dact != V2S_ACT_NONE ? p.z : p.residual;
  const long gld = p.dact != V2S_ACT_NONE ? p.ldz : p.ldr;
  const bool ahead = gsrc != nullptr && p.dbg == 0;
  uint4 gop[NIT], gnext[NIT];
  auto fetch = [&](int ps, uint4 (&g)[NIT]) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = tid + it * 512;
      const int lr = c / CPR, cc = (c % CPR) * 8;
      const int gm = m0 + (lr >> 5) * (MI * 16) + ps * 32 + (lr & 31), gn = n0 + cc;
      g[it] = (gm < p.M && gn < p.N) ? *reinterpret_cast<const uint4*>(gsrc + (long)gm * gld + gn) : make_uint4(0, 0, 0, 0);
    }
  };
  if (ahead) fetch(0, gop);
#pragma clang loop unroll(full)                     // acc[] is indexed by ps
  for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(cs + (wm * 32 + ii * 16 + (lane & 15)) * PB + wn * 64 + j * 16 + (lane >> 4) * 4) = acc[ps * 2 + ii][j];
    __syncthreads();
    if (ahead) {
      if (ps + 1 < NPASS) fetch(ps + 1, gnext);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = tid + it * 512;
        const int lr = c / CPR, cc = (c % CPR) * 8;
        const int gm = m0 + (lr >> 5) * (MI * 16) + ps * 32 + (lr & 31), gn = n0 + cc;
        if (gm < p.M && gn < p.N) {
          float v[8];
          const float4 x0 = *reinterpret_cast<const float4*>(cs + lr * PB + cc);
          const float4 x1 = *reinterpret_cast<const float4*>(cs + lr * PB + cc + 4);
          v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
          epilogue_chunk<true>(p, v, gm, gn, slice, gop[it]);
        }
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) gop[it] = gnext[it];
    } else {
#pragma unroll 1
      for (int c = tid; c < ROWS_PASS * CPR; c += 512) {
        const int lr = c / CPR, cc = (c % CPR) * 8;
        const int gm = m0 + (lr >> 5) * (MI * 16) + ps * 32 + (lr & 31), gn = n0 + cc;
        if (gm >= p.M || gn >= p.N) continue;
        float v[8];
        const float4 x0 = *reinterpret_cast<const float4*>(cs + lr * PB + cc);
        const float4 x1 = *reinterpret_cast<const float4*>(cs + lr * PB + cc + 4);
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
        if (p.dbg == 1) {                           // ablation: everything but the global store
          asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
          continue;
        }
        epilogue_chunk(p, v, gm, gn, slice);
      }
    }
    if (ps + 1 < NPASS) __syncthreads();
  }
}

// =====================================================================================================================
// Persistent form of the 8-phase kernel with a DEFERRED epilogue (256 x 256 tiles, bf16 output, epilogues without global operands).
// Ablation of gemm_p8_kernel on the wide K = 768 shapes of the step (profiles/r02_gemm_p8_ablation_v1.txt): of 150 us, 29 us are the
// output stores -- every CU finishes its tile at the same moment and 256 CUs x 128 KiB hit HBM as one burst -- and 22 us the LDS
// staging in front of them; the matrix pipe idles through both.  Here a block walks its tiles in a loop and the output of tile t
// leaves the chip DURING the main loop of tile t+1:
//   * after the last MFMA of a tile the accumulators are rounded to bf16 (ReLU and the dropout scale applied in registers: the same
//     single rounding as the synchronous epilogue) into 64 "held" registers, the accumulators restart at zero;
//   * in stages 0..8 of the next tile, one 32-row slab of the held tile per stage goes through a 2 x 16 KiB LDS staging block
//     (dumped by the four waves that own it, read back as 16-byte row chunks by all eight) and is stored with the dropout mask
//     applied on the way;
//   * the DMA stream does not stop at a tile edge: the first stages of the next tile are requested during the last phases of the
//     current one (no prologue except for the block's first tile), and past the block's last tile it re-requests that tile's
//     last stage into free ring slots so that every counted wait keeps the same constant.
// vmcnt bookkeeping: the stores are VMEM operations of the same in-order counter as the DMAs.  Per lane, program order is
//   ... B(s+1) | A(s+2) | st st B(s+2) | A(s+3) wait ...    (st = the two stores of an odd phase while a write-out is running)
// so the wait that retires stage s+1 leaves 8 operations in flight while stores are interleaved (tile-local stages 1..8) and 6
// otherwise.  Stores are issued by every wave of the block in those phases whether or not a previous tile exists (first tile: all
// offsets out of range, dropped by the buffer bounds check), so the constants do not depend on data.
// Measured dead ends (profiles/r02_gemm_p8d_ablation.txt): the slabs are written back to back ON PURPOSE.  A counted wait behind a
// store also waits for that store to be acknowledged (one in-order queue); spread over the whole tile every store stalled the ring on
// its own (+35 %), back to back their latencies overlap.  The memory half-phases have no slack -- whatever they take beyond the other
// half's 16 MFMAs is paid twice per stage -- so the write-out steps are kept in straight-line, fully specialised code.
// The kernel counts its VMEM operations by hand: build.sh fails the build if the compiler ever spills a register of it to scratch.
__device__ __forceinline__ uint2 p8_pack4(const f32x4& v) {
  uint2 r;
  r.x = pack2bf(v[0], v[1]); r.y = pack2bf(v[2], v[3]);
  return r;
}
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <bool TA, bool TB, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_p8d_kernel(const GemmP p) {
  using G = P8<256>;
  constexpr int BM2 = 256, BN2 = 256, A_B = G::A_B, STG = G::STG, NST = G::NST, RING = NST * STG;
  extern __shared__ __attribute__((aligned(16))) char smem[];        // ring (128 KiB) + 2 x 16 KiB write-out staging
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int half = wm;
  const int ntiles = p.tilesM * p.tilesN;
  const int G_ = gridDim.x;
  const int nmy = (ntiles - (int)blockIdx.x + G_ - 1) / G_;          // tiles of this block: ids blockIdx.x + k * gridDim.x
  const int M8 = (p.M + 7) & ~7, N8 = (p.N + 7) & ~7;
  const int nst = p.K / 32;                                          // >= 9 (dispatcher)

  f32x4 acc[8][4];
  uint2 held[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; held[i][j] = make_uint2(0u, 0u); }

  // ---- DMA issue streams (A and B advance separately: A runs one unit ahead of B)
  const long stepA = TA ? 32L * p.lda * 2 : 64L, stepB = TB ? 32L * p.ldb * 2 : 64L;
  const uint32_t sbase = lds_addr(smem);
  const uint32_t dstA = __builtin_amdgcn_readfirstlane(sbase + wave * 2048);
  const uint32_t dstB = __builtin_amdgcn_readfirstlane(sbase + A_B + wave * 2048);
  uint32_t oa[2], ob[2];
  const char* ga; const char* gb;
  long incA = stepA, incB = stepB;
  int sA = 0, sB = 0, kA = 0, kB = 0, slotA = 0, slotB = 0;
  auto setupA = [&](int k) {
    int tm, tn, sl;
    tile_coords(p, xcd_remap((int)blockIdx.x + k * G_, ntiles), tm, tn, sl);
    oa[0] = p8_lane_off<BM2, TA>(p.lda, tm * BM2, p.M, M8, wave * 2, lane);
    oa[1] = p8_lane_off<BM2, TA>(p.lda, tm * BM2, p.M, M8, wave * 2 + 1, lane);
    ga = reinterpret_cast<const char*>(p.A);
  };
  auto setupB = [&](int k) {
    int tm, tn, sl;
    tile_coords(p, xcd_remap((int)blockIdx.x + k * G_, ntiles), tm, tn, sl);
    ob[0] = p8_lane_off<BN2, TB>(p.ldb, tn * BN2, p.N, N8, wave * 2, lane);
    ob[1] = p8_lane_off<BN2, TB>(p.ldb, tn * BN2, p.N, N8, wave * 2 + 1, lane);
    gb = reinterpret_cast<const char*>(p.B);
  };
  auto issueA = [&]() {
    p8_dma2(oa[0], oa[1], ga, dstA + slotA);
    slotA = slotA + STG == RING ? 0 : slotA + STG;
    ga += incA;
    if (++sA == nst) {
      if (kA + 1 < nmy) { ++kA; sA = 0; setupA(kA); }
      else { ga -= stepA; incA = 0; sA = -(1 << 30); }                         // exhausted: keep re-requesting the last stage
    }
  };
  auto issueB = [&]() {
    p8_dma2(ob[0], ob[1], gb, dstB + slotB);
    slotB = slotB + STG == RING ? 0 : slotB + STG;
    gb += incB;
    if (++sB == nst) {
      if (kB + 1 < nmy) { ++kB; sB = 0; setupB(kB); }
      else { gb -= stepB; incB = 0; sB = -(1 << 30); }
    }
  };

  // ---- write-out of the previous tile
  const long cbytes = (long)p.M * p.ldc * 2;
  const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)cbytes, 0x00020000);
  char* stg = smem + RING;
  int pm0 = p.M, pn0 = 0;                                             // tile origin of the held tile; pm0 = M: nothing held, every store out of range
  const int arow = wm * 128, brow = wn * 64;
  auto dump_unit = [&](auto Uc) {                                     // the four waves of half U/4: their fragments 2(U&3), 2(U&3)+1 -> staging U&1
    constexpr int U = decltype(Uc)::value;
    char* buf = stg + (U & 1) * 16384;
    int ln = lane;
    asm volatile("" : "+v"(ln));                                      // opaque: keeps the address arithmetic out of the loop-invariant (always live) set
    // row r = ii*16 + (ln & 15); 16-byte chunk (wn*8 + j*2 + (ln >> 5)) ^ (ln & 15); half (ln >> 4) & 1
    const int base = (ln & 15) * 512 + (((ln >> 4) & 1) << 3);
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ch = ((wn * 8 + j * 2 + (ln >> 5)) ^ (ln & 15));
        *reinterpret_cast<uint2*>(buf + ii * 16 * 512 + base + (ch << 4)) = held[2 * (U & 3) + ii][j];
      }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto store_unit = [&](auto Uc) {                                    // every thread: two 16-byte row chunks of slab U
    constexpr int U = decltype(Uc)::value;
    const char* buf = stg + (U & 1) * 16384;
    int td = tid;
    asm volatile("" : "+v"(td));                                      // opaque (see dump_unit)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c = td + q * 512, r = c >> 5, cc = c & 31;
      u32x4 x = *reinterpret_cast<const u32x4*>(buf + r * 512 + ((cc ^ (r & 15)) << 4));
      const int gm = pm0 + U * 32 + r, gn = pn0 + cc * 8;
      if (p.p16) {
        const uint32_t m = v2s_keep8((unsigned long long)(gm + p.row0) * (unsigned long long)p.N + gn, v2s_salted(p.seed, p.salt), p.p16);
        x[0] &= ((m & 1u) ? 0xffffu : 0u) | ((m & 2u) ? 0xffff0000u : 0u);
        x[1] &= ((m & 4u) ? 0xffffu : 0u) | ((m & 8u) ? 0xffff0000u : 0u);
        x[2] &= ((m & 16u) ? 0xffffu : 0u) | ((m & 32u) ? 0xffff0000u : 0u);
        x[3] &= ((m & 64u) ? 0xffffu : 0u) | ((m & 128u) ? 0xffff0000u : 0u);
      }
      const bool ok = gm < p.M && gn < p.N;
      const uint32_t off = ok ? (uint32_t)(((long)gm * p.ldc + gn) * 2) : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b128(x, crsrc, (int)off, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto convert = [&]() {                                              // accumulators -> held (bf16), accumulators restart at zero
    const bool relu = p.act == V2S_ACT_RELU;
    const float sc = p.inv_keep;                                      // 1 without dropout
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v = acc[i][j];
        if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        v[0] *= sc; v[1] *= sc; v[2] *= sc; v[3] *= sc;
        held[i][j] = p8_pack4(v);
        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
  };

  bf16x8 af[4], bfr[4];
  int rslot = 0;
  auto mfma_lo = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  auto mfma_hi = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[4 + i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  // one stage = two phases; S = tile-local stage index for the write-out steps (0..8), -1 = none
  auto stage = [&](auto Sc) {
    constexpr int S = decltype(Sc)::value;
    const char* sa = smem + rslot;
    const char* sb = sa + A_B;
    int ln = lane;
    asm volatile("" : "+v"(ln));                  // opaque: fragment addresses are recomputed per stage instead of living in registers for the whole kernel
    if constexpr (S >= 1 && S <= 8) store_unit(std::integral_constant<int, (S >= 1 ? S - 1 : 0)>{});
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = read_frag_w4<BN2, TB>(sb, brow + j * 16, ln);
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = read_frag_w4<BM2, TA>(sa, arow + i * 16, ln);
    if constexpr (S >= 0 && S <= 7) {
      if (half == (S >> 2)) dump_unit(std::integral_constant<int, (S >= 0 ? S : 0)>{});
      issueB();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the dump is in LDS before this wave reaches the barrier
    } else {
      issueB();
    }
    P8_BARRIER();
    mfma_lo();
    P8_BARRIER();
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = read_frag_w4<BM2, TA>(sa, arow + 64 + i * 16, ln);
    issueA();
    if constexpr (S >= 1 && S <= 8) dma_wait_n<8>(); else dma_wait_n<6>();
    P8_BARRIER();
    mfma_hi();
    P8_BARRIER();
    rslot = rslot + STG == RING ? 0 : rslot + STG;
  };

  setupA(0); setupB(0);
  issueA(); issueB(); issueA(); issueB(); issueA();
  dma_wait_n<6>();
  P8_BARRIER();
  if (half == 1) P8_BARRIER();

  for (int k = 0; k < nmy; ++k) {
    stage(std::integral_constant<int, 0>{}); stage(std::integral_constant<int, 1>{}); stage(std::integral_constant<int, 2>{});
    stage(std::integral_constant<int, 3>{}); stage(std::integral_constant<int, 4>{}); stage(std::integral_constant<int, 5>{});
    stage(std::integral_constant<int, 6>{}); stage(std::integral_constant<int, 7>{}); stage(std::integral_constant<int, 8>{});
#pragma unroll 1
    for (int s = 9; s < nst; ++s) stage(std::integral_constant<int, -1>{});
    convert();
    int tm, tn, sl;
    tile_coords(p, xcd_remap((int)blockIdx.x + k * G_, ntiles), tm, tn, sl);
    pm0 = tm * BM2; pn0 = tn * BN2;
  }
  // ---- drain: the block's last tile is still in the held registers
  if (half == 0) P8_BARRIER();
  dma_wait_n<0>();
  __syncthreads();
  if constexpr (ABL == 1) return;                 // ablation build (option gemm_dbg 3): no drain
  auto drain = [&](auto Uc) {
    constexpr int U = decltype(Uc)::value;
    if (half == (U >> 2)) dump_unit(Uc);
    __syncthreads();
    store_unit(Uc);
  };
  drain(std::integral_constant<int, 0>{}); drain(std::integral_constant<int, 1>{}); drain(std::integral_constant<int, 2>{});
  drain(std::integral_constant<int, 3>{}); drain(std::integral_constant<int, 4>{}); drain(std::integral_constant<int, 5>{});
  drain(std::integral_constant<int, 6>{}); drain(std::integral_constant<int, 7>{});
}

// =====================================================================================================================
// (Round 4's two structural experiments lived here -- a persistent 128 x 128 kernel with dedicated write-out waves, "gemm_ps", and a 4-wave 256 x 192
// kernel with 128 x 96 wave tiles in AGPRs, "gemm_wt" -- both bit-identical to the default dispatch and neither faster (DESIGN.md 8a-r4 keeps the
// ablations and counters).  Removed in round 6: the generated gemm_a4p kernel is what that line of work led to.)

// =====================================================================================================================
// 16-byte load of a weight slice that is read ONCE per decode step (SKINNY_NT: non-temporal hint, the line is not kept in the caches the
// activations and the KV cache live in)
#ifndef SKINNY_NT
#define SKINNY_NT 0
#endif
// 1: a scheduling barrier between the load group and the MFMA group of a K iteration.  Without it hipcc interleaves them and REUSES the registers of the
// first step for the last one (57 VGPRs used): the loads of step 3 are issued only after step 1 has arrived -- two or three dependent memory round trips
// per launch in a kernel that is one latency chain (A/B builds: 0)
#ifndef SKINNY_LOADS_FIRST
#define SKINNY_LOADS_FIRST 1
#endif
// diagnostic build (tools/skinny_stamps.py): thread 0 of the first and of the last block stamp the shader clock (s_memtime) and the 100 MHz
// reference clock (s_memrealtime) at the kernel's phase boundaries into the caller's workspace; never defined in the product build
#ifndef SKINNY_STAMPS
#define SKINNY_STAMPS 0
#endif
#if SKINNY_STAMPS
#define SK_STAMP(i_) do { st_r[i_] = __builtin_amdgcn_s_memrealtime(); } while (0)       // (uniform values: they stay in SGPRs, the kernel's VGPR count and occupancy must not change)
#else
#define SK_STAMP(i_) do { } while (0)
#endif
__device__ __forceinline__ uint4 ld_stream(const bf16_t* ptr) {
#if SKINNY_NT
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_;
  const u32x4_ v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_*>(ptr));
  return make_uint4(v[0], v[1], v[2], v[3]);
#else
  return *reinterpret_cast<const uint4*>(ptr);
#endif
}

// Skinny GEMM for cached decoding (M <= 64 rows: one token per live sequence / beam): C[M][N] = A[M][K] . B[N][K]^T.
// The weight matrix B is the only real traffic (read once); the 128-wide tiles above would occupy 6..24 CUs for it.  Here a block
// owns 16 output columns: its 4 waves split K four ways, each streaming its [16][K/4] weight slab straight from HBM into MFMA
// B-fragments (16 B per lane, no LDS) against the L2-resident activations, then the four partial 64x16 tiles are summed through
// LDS and the usual epilogue runs on 8-wide chunks.  Grid = N/16 blocks (48..200 for the decoder projections, 2013 for the LM head).
// MT = 16-row fragments per block: 4 (a block covers all 64 rows; blockIdx.y = 0) or 1 (blockIdx.y walks the row fragments: four
// times the blocks, a quarter of the activation loads per wave -- the long-K projections, where 48 blocks x one latency-bound load
// chain each left the chip idle: decoder wo 64x768x3072 18.7 us -> see profiles/r02_decode_step.txt).
// NT = 16-column fragments per block (beam search: M = batch x beams = 256 rows made the activation re-reads -- every 16-column
// block streams all M x K of A through L2 -- the larger stream; a 32 x 32 block halves both).
template <int WAVES, int STEPS, int MT, int NT = 1>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_kernel(const GemmP p) {
  constexpr int PITCH = NT == 1 ? 17 : NT * 16 + 4;
  __shared__ float part[WAVES][MT * 16][PITCH];
#if SKINNY_STAMPS
  unsigned long long st_r[8] = {};
#endif
  SK_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * (NT * 16), m0 = blockIdx.y * (MT * 16);
  const int kq = p.K / WAVES;                    // multiple of 32 * STEPS (checked by the dispatcher)
  const int kbeg = wave * kq;
  const int r = lane & 15, kc = (lane >> 4) * 8;
  const bf16_t* bp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int brow = n0 + t * 16 + r; brow = brow < p.N ? brow : p.N - 1;
    bp[t] = p.B + (long)brow * p.ldb + kbeg + kc;
  }
  const bf16_t* ap[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int m = m0 + i * 16 + r; m = m < p.M ? m : p.M - 1;
    ap[i] = p.A + (long)m * p.lda + kbeg + kc;
  }
  // a thread owns at most one 8-wide output chunk: its residual operand is requested NOW, so that the epilogue's one dependent global
  // load (an L2 round trip on the critical path of a launch that is all latency) hides behind the weight stream
  static_assert(MT * 16 * NT * 2 <= WAVES * 64, "one output chunk per thread");
  const bool res_ahead = p.residual != nullptr && p.dact == V2S_ACT_NONE && p.dbg != 4;      // (gemm_dbg = 4: A/B switch, same results)
  uint4 res_chunk = uint4{0, 0, 0, 0};
  if (res_ahead && tid < MT * 16 * NT * 2) {
    const int rm = m0 + tid / (NT * 2), rn = n0 + (tid % (NT * 2)) * 8;
    if (rm < p.M && rn < p.N) res_chunk = *reinterpret_cast<const uint4*>(p.residual + (long)rm * p.ldr + rn);
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool rms = p.rms_eps > 0.f;
  SK_STAMP(1);                                // arguments read, addresses formed
  float sq[MT];                               // fused RMSNorm: sum of squares of this lane's A elements, per 16-row fragment
#pragma unroll
  for (int i = 0; i < MT; ++i) sq[i] = 0.f;
  // STEPS K-steps (of 32) are loaded back to back before their MFMAs: (NT + MT) x STEPS 16-byte loads in flight per lane
  for (int k = 0; k < kq; k += 32 * STEPS) {
    uint4 bq[STEPS][NT], aq[STEPS][MT];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
#pragma unroll
      for (int t = 0; t < NT; ++t) bq[s][t] = ld_stream(bp[t] + k + s * 32);
#pragma unroll
      for (int i = 0; i < MT; ++i) aq[s][i] = *reinterpret_cast<const uint4*>(ap[i] + k + s * 32);
    }
#if SKINNY_LOADS_FIRST
    __builtin_amdgcn_sched_barrier(0);          // every load of the iteration is issued before its first MFMA (see SKINNY_LOADS_FIRST)
#endif
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aq[s][i]), __builtin_bit_cast(bf16x8, bq[s][t]), acc[i][t], 0, 0, 0);
    if (rms) {
#pragma unroll
      for (int s = 0; s < STEPS; ++s)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          float f[8];
          unpack8(aq[s][i], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) sq[i] = fmaf(f[j], f[j], sq[i]);
        }
    }
  }
  __shared__ float rowsq[WAVES][MT * 16];
#if SKINNY_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  SK_STAMP(2);                                // K loop done (every load arrived, MFMAs issued)
  if (rms) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float v = sq[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 16) rowsq[wave][i * 16 + lane] = v;
    }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) part[wave][i * 16 + (lane >> 4) * 4 + q][t * 16 + (lane & 15)] = acc[i][t][q];
  SK_STAMP(3);                                // partial tiles written to LDS
  __syncthreads();
  SK_STAMP(4);                                // barrier passed
  for (int e = tid; e < MT * 16 * NT * 2; e += WAVES * 64) {
    const int ml = e / (NT * 2), m = m0 + ml, c = (e % (NT * 2)) * 8;
    const int gn = n0 + c;
    if (m < p.M && gn < p.N) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) t += part[w][ml][c + j];
        v[j] = t;
      }
      if (rms) {
        float ss = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) ss += rowsq[w][ml];
        const float rstd = rsqrtf(ss / (float)p.K + p.rms_eps);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= rstd;
      }
#if SKINNY_STAMPS
      if (e == 0) SK_STAMP(5);                // the eight partial sums added, row scale applied
#endif
      if (res_ahead) epilogue_chunk<true>(p, v, m, gn, 0, res_chunk);
      else epilogue_chunk<false>(p, v, m, gn, 0);
    }
  }
#if SKINNY_STAMPS
  if (p.ws && tid == 0) {          // every block: entry and exit on the reference clock, from word 32 on
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.ws) + 32 + 2 * (blockIdx.y * gridDim.x + blockIdx.x);
    o[0] = st_r[0]; o[1] = __builtin_amdgcn_s_memrealtime();
  }
  if (p.ws && tid == 0 && ((blockIdx.x == 0 && blockIdx.y == 0) || (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1))) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SK_STAMP(6);                              // output chunk stored (written back as far as vmcnt tells)
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.ws) + ((blockIdx.x == 0 && blockIdx.y == 0) ? 0 : 16);
    for (int i = 0; i < 7; ++i) o[8 + i] = st_r[i];
  }
#endif
}

// LM head of a cached decode step: M <= 64 rows against N ~ 32k columns -- 49 MB of weights, the largest single read of the step.
// The K-split kernel above re-reads the activations per 16-column block (2013 blocks x 98 KB through L2) and runs 8 dependent rounds
// of blocks: 39.5 us = 1.25 TB/s (profiles/r02_decode_step.txt).  Here one block per CU stages the 64 x K activations in LDS once;
// each wave then owns whole 16-column tiles over the full contraction (no cross-wave reduction, no block barrier after the fill):
// the tile's K/32 weight fragments (24 x 16 B per lane for K = 768) are requested together BEFORE the activation fill, so the HBM
// stream starts at launch, and the next tile's are requested as soon as the MFMAs have consumed the registers.
template <int KS>   // K = 32 * KS
__global__ __launch_bounds__(512) void gemm_skinny_wide_kernel(const GemmP p) {
  extern __shared__ __align__(16) unsigned char wide_smem[];
  constexpr int K = KS * 32, PITCH = K + 8;          // row pitch = 2K + 16 bytes: the 16 rows of a fragment read hit distinct banks
  bf16_t* As = reinterpret_cast<bf16_t*>(wide_smem);
  float* rowsq = reinterpret_cast<float*>(wide_smem + 64 * PITCH * 2);
  float* stage = rowsq + 64;                         // 8 waves x [16][17]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kc = (lane >> 4) * 8;
  const int tiles = (p.N + 15) / 16, stride = gridDim.x * 8;
  int tile = blockIdx.x * 8 + wave;
  uint4 bq[KS];
  auto load_b = [&](int t) {
    int brow = t * 16 + r; brow = brow < p.N ? brow : p.N - 1;
    const bf16_t* bp = p.B + (long)brow * p.ldb + kc;
#pragma unroll
    for (int s = 0; s < KS; ++s) bq[s] = ld_stream(bp + s * 32);
  };
  if (tile < tiles) load_b(tile);
  const bool rms = p.rms_eps > 0.f;
  {
    const int row = tid >> 3, j = tid & 7;
    const int m = row < p.M ? row : p.M - 1;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < K / 64; ++i) {
      const int c = (j + 8 * i) * 8;
      const uint4 v = *reinterpret_cast<const uint4*>(p.A + (long)m * p.lda + c);
      *reinterpret_cast<uint4*>(As + row * PITCH + c) = v;
      if (rms) {
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) sq = fmaf(f[e], f[e], sq);
      }
    }
    sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
    if (j == 0) rowsq[row] = sq;
  }
  __syncthreads();
  float* st = stage + wave * (16 * 17);
  for (; tile < tiles; tile += stride) {
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(As + (i * 16 + r) * PITCH + s * 32 + kc);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, bq[s]), acc[i], 0, 0, 0);
      }
    const int n0 = tile * 16;
    if (tile + stride < tiles) load_b(tile + stride);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // wave-private transpose of one 16 x 16 fragment: DS operations of a wave complete in order, the waitcnt is also the
      // compiler barrier between the writes and the reads of the other lanes
#pragma unroll
      for (int q = 0; q < 4; ++q) st[((lane >> 4) * 4 + q) * 17 + (lane & 15)] = acc[i][q];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane < 32) {
        const int ml = lane >> 1, m = i * 16 + ml, c = (lane & 1) * 8, gn = n0 + c;
        if (m < p.M && gn < p.N) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = st[ml * 17 + c + e];
          if (rms) {
            const float rstd = rsqrtf(rowsq[m] / (float)p.K + p.rms_eps);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= rstd;
          }
          epilogue_chunk(p, v, m, gn, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

// C[m][n] (+)= alpha * sum_z ws[z][m][n]   (deterministic: fixed slice order)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, long ldc, int M,
                                                            int N, int S, float alpha, int accumulate, int c_bf16) {
  const int n4 = N >> 2;
  const long total = (long)M * n4;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int m = (int)(t / n4), c = (int)(t - (long)m * n4) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < S; ++z) {
      const float4 v = *reinterpret_cast<const float4*>(ws + ((long)z * M + m) * N + c);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float4 o = make_float4(s.x * alpha, s.y * alpha, s.z * alpha, s.w * alpha);
    if (c_bf16) {          // bf16 output (round 6: the LM head's d(hidden) leaves its split-K reduction rounded once, no separate cast launch)
      *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(C) + (long)m * ldc + c) = make_uint2(pack2bf(o.x, o.y), pack2bf(o.z, o.w));
      continue;
    }
    float* cp = C + (long)m * ldc + c;
    if (accumulate) {
      const float4 c0 = *reinterpret_cast<const float4*>(cp);
      o.x += c0.x; o.y += c0.y; o.z += c0.z; o.w += c0.w;
    }
    *reinterpret_cast<float4*>(cp) = o;
  }
}

// out[r][c] = W[r][c] * w[c]
__global__ __launch_bounds__(256) void scale_cols_kernel(const bf16_t* __restrict__ W, const float* __restrict__ w, bf16_t* __restrict__ out,
                                                         long total8, int cols8) {
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total8; t += (long)gridDim.x * 256) {
    const int c = (int)(t % cols8) * 8;
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(W + t * 8), f);
    const float4 w0 = *reinterpret_cast<const float4*>(w + c), w1 = *reinterpret_cast<const float4*>(w + c + 4);
    f[0] *= w0.x; f[1] *= w0.y; f[2] *= w0.z; f[3] *= w0.w; f[4] *= w1.x; f[5] *= w1.y; f[6] *= w1.z; f[7] *= w1.w;
    *reinterpret_cast<uint4*>(out + t * 8) = pack8(f);
  }
}

// ---- column sums (bias gradients) ------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ X, long ldx, int M, int N,
                                                     float* __restrict__ out, int rows_per_block) {
  // block handles 64 columns x rows_per_block rows; thread (c8 = tid&7 -> 8 columns, r = tid>>3)
  __shared__ float red[32][65];
  const int tid = threadIdx.x;
  const int c8 = tid & 7, rr = tid >> 3;
  const int n0 = blockIdx.x * 64 + c8 * 8;
  const int mbeg = blockIdx.y * rows_per_block, mend = min(M, mbeg + rows_per_block);
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (n0 < N) {
    // eight rows' loads in flight per thread (the rolled loop waited for every 16-byte load before it asked for the next one: 32 dependent round trips
    // per block, 12 us for a 3200-row ViT tensor); the additions keep their order
    int m = mbeg + rr;
    for (; m + 7 * 32 < mend; m += 8 * 32) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const uint4*>(X + (long)(m + u * 32) * ldx + n0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float f[8];
        unpack8(v[u], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += f[j];
      }
    }
    for (; m < mend; m += 32) {
      float f[8];
      unpack8(*reinterpret_cast<const uint4*>(X + (long)m * ldx + n0), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += f[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rr][c8 * 8 + j] = s[j];
  __syncthreads();
  if (tid < 64) {
    float t = 0.f;
    for (int r = 0; r < 32; ++r) t += red[r][tid];
    const int n = blockIdx.x * 64 + tid;
    if (n < N) atomicAdd(out + n, t);
  }
}

#include "v2s_gemm_a4.h"

}  // namespace

static int num_cus() {
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, n = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return ncu;
}

// Tile width of the 8-phase kernel for this shape, 0 = keep the older kernels.  (Rules from tools/gemm_p8_ab.py, profiles/r02_gemm_p8_ab.txt.)
static int p8_auto(const v2s_gemm_args* a, bool deferred_ok) {
  // measured, variants interleaved in one process (tools/gemm_p8_ablate.py, gemm_p8_ab.py; profiles/r02_gemm_p8*.txt), us old / 8-phase
  // synchronous / deferred:  32000x2304x768 132 / 136 / 127,  32000x3072x768 relu+dropout 192 / 182 / 174,  plain 174 / 162 / 148,
  // 65536x2048x2048 526 / 490 / 501,  8192^3 966 / 792 / 852,  t5-large 16000x3072x1024 118 / 113 / 102, 16000x1024x4096 141 / 122 / -;
  // few-tile shapes (N = 768: 375 tiles on 256 CUs; the 8192- and 3200-row decoder / ViT shapes) stay on the 128 x 128 kernels.
  if (a->M < 256 || a->N < 256) return 0;
  const long t256 = (long)((a->M + 255) / 256) * ((a->N + 255) / 256);
  if (t256 < 512) {
    // under two rounds of 256 x 256 tiles the 128 x 128 kernels win on tail quantisation -- unless the tiles happen to fill whole
    // rounds (t5-large: 16000 x 1024 -> 252 tiles, 32000 x 1024 -> 500): 16000x1024x4096 141 -> 122 us, 16000x1024x3072 (dgrad) 104 -> 90
    const int ncu = num_cus();
    const long rounds = (t256 + ncu - 1) / ncu;
    const bool full_rounds = v2s_opt_gemm_p8() != 5 && t256 * 100 >= rounds * ncu * 92 && a->K >= 1024 && !a->transA;
    if (full_rounds) return 256;
    // split-K weight gradients with a long contraction and >= 24 output tiles (tools/gemm_wgrad_ab.py, round 4; us 4-wave ring or 8-wave
    // 256-row kernel / 8-phase): 2304x768x32000 124.9 / 120.5, 768x3072x32000 155.2 / 145.1, 3072x768x32000 159.6 / 158.4; smaller outputs
    // (768x768, 1536x768) and the K = 8192 / 3200 decoder and ViT shapes lose 5-30 %
    if (a->transA && a->workspace && t256 >= 24 && a->K >= 16384) return 256;
    return (a->transA && t256 >= 256 && a->K >= 4096) ? 256 : 0;
  }
  if (deferred_ok && a->K <= 1536) return 256;          // p8_decide turns deferred_ok into the deferred form
  if (a->K >= 1024 || a->transA) return 256;
  // LM-head logits (fp32 output, N = 32200): 2048x32200x768 136 -> 126 us, 1024 rows 71 -> 66 (tools/gemm_vendor_ab.py shapes, round 3)
  if (a->c_dtype == V2S_F32 && a->N >= 8192 && !a->transB) return 256;
  return 0;
}

// Shapes the 4-wave asm-scheduled 256 x 256 kernels take by default (gemm_a4 = 1).  Measured with tools/gemm_a4_ab.py, variants interleaved in
// one process (profiles/r05_b_gemm_a4p_ab.txt; us default / persistent a4p / vendor): 32000x2304x768 139 / 108 / 108, 32000x768x768 52 / 44 / 47,
// 32000x3072x768 168 / 135 / 134, 32000x768x3072 150 / 139 / 125, dgrads 32000x768x2304 122 / 104 / 103, 32000x3072x768 172 / 127 / 154,
// 32000x768x3072 154 / 133 / 163, 8192x2304x768 45 / 42, 8192x3072x768 52 / 45; with fewer tiles than CUs the 128 x 128 kernels win
// (8192x768x768 18.8 / 20.2, 8192x768x3072 47 / 59).  The one-tile form (synchronous epilogue, any epilogue) loses to the default dispatch
// everywhere (1 block per CU: nothing runs beside its epilogue) and is only taken when forced.
// Epilogues the persistent deferred-write-out kernel has: 0 = none of them, 1 = plain bf16, 2 = the ReLU-mask dgrad (dact = RELU with z laid out like C,
// optional 1 / (1 - p) scale of the forward's dropout: out = z > 0 ? acc * scale : 0)
// 3 = ReLU (forward, act = RELU), 4 = ReLU + dropout (the FFN's wi forward; the mask of v2s_keep8 regenerated in the kernel: ldc == N, K >= 640)
static int a4p_epilogue(const v2s_gemm_args* a) {
  if (a->transA || a->c_dtype != V2S_BF16 || a->accumulate || a->bias || a->pre || a->residual || a->alpha != 1.0f ||
      a->M < 256 || a->N < 512 || (a->N % 8) != 0)
    return 0;
  if (a->act == V2S_ACT_RELU) {
    if (a->transB || a->dact != V2S_ACT_NONE || v2s_opt_gemm_a4_relu() == 0) return 0;
    if (a->dropout_p == 0.f) return a->K >= 384 ? 3 : 0;
    const uint32_t p16 = (uint32_t)(a->dropout_p * 65536.0f + 0.5f);
    return (a->K >= 640 && a->ldc == a->N && (long)a->M * a->N < (1L << 35) && p16 >= 1 && p16 <= 65535) ? 4 : 0;
  }
  if (a->act != V2S_ACT_NONE) return 0;
  if (a->dact == V2S_ACT_NONE) return (a->dropout_p == 0.f && a->K >= 384) ? 1 : 0;
  // Measured (profiles/r05_a4p_dact_ab.txt, r05_step_ab_a4.txt): the encoder wo dgrad 32000x3072x768 with its mask operand 251.5 -> 189.7 us alone,
  // but +0.5 ... +0.8 ms per train step on three boxes (co-scheduling with the weight-gradient stream, cold operands and the burst of mask-operand
  // loads were each tested as the cause and ruled out: DESIGN.md 8a-r5).  So only when forced (gemm_a4 = 2) or asked for (gemm_a4 = 5).
  const int mode = v2s_opt_gemm_a4();
  // (not in place: ragged M / N are covered by an OVERLAPPING last tile row / column, whose second visit would read z after the first wrote the
  // masked gradient over it -- the generic kernels are in-place safe, so C == z keeps them)
  if (a->dact == V2S_ACT_RELU && a->transB && a->z && a->z != a->C && a->ldz == a->ldc && a->K >= 512 && (mode == 2 || mode == 5)) return 2;
  return 0;
}

static bool a4_auto(const v2s_gemm_args* a, long t256) {
  const bool persistent_ok = a4p_epilogue(a) != 0;
  if (a->transA)      // split-K weight gradients with a long contraction (tools/gemm_wgrad_ab.py, profiles/r05_c_gemm_a4_wgrad_ab.txt; us before / a4):
                      // 2304x768x32000 134 / 112, 3072x768x32000 159 / 144, 768x3072x32000 157 / 145, 1536x768x35200 91 / 86; 768x768x32000 (9 tiles) 54 / 56
    return v2s_opt_gemm_a4() == 4 && a->workspace != nullptr && a->c_dtype == V2S_F32 && a->K >= 16384 && t256 >= 18;
  return !a->transA && persistent_ok && t256 >= 256;
}

// Which form of the 8-phase kernel runs this problem: p8 = tile width (0 = none), p8d = deferred-epilogue persistent form.
static void p8_decide(const v2s_gemm_args* a, bool tr, int& p8, bool& p8d) {
  p8 = 0; p8d = false;
  const int p8_mode = v2s_opt_gemm_p8();
  // legal: K a multiple of 32 with >= 4 stages, the tr-read path, 32-bit per-lane source offsets
  const bool p8_ok = p8_mode != 0 && tr && (a->K % 32) == 0 && a->K >= 128 && a->M >= 128 && a->N >= 64 &&
                     (a->transA ? 32 * a->lda + a->M : (long)a->M * a->lda) < (1L << 30) &&
                     (a->transB ? 32 * a->ldb + a->N : (long)a->N * a->ldb) < (1L << 30);
  if (!p8_ok) return;
  // deferred-epilogue persistent form: bf16 output, epilogue without global operands (ReLU / dropout only), no split-K
  const bool p8d_ok = !a->transA && a->c_dtype == V2S_BF16 && !a->accumulate && !a->bias && !a->pre && !a->residual && a->dact == V2S_ACT_NONE &&
                      (a->act == V2S_ACT_NONE || a->act == V2S_ACT_RELU) && a->alpha == 1.0f && a->K >= 288 && a->N >= 256 && a->M >= 256 &&
                      (long)a->M * a->ldc * 2 < (1L << 31);
  if (p8_mode == 2) p8 = 256;
  else if (p8_mode == 3) p8 = 128;
  else if (p8_mode == 4) { p8 = 256; p8d = p8d_ok; }
  else { p8 = p8_auto(a, p8d_ok); p8d = p8 == 256 && p8d_ok && a->K <= 1536; }
}


static thread_local const char* g_last_gemm = "";
extern "C" const char* v2s_last_gemm_kernel(void) { return g_last_gemm; }
static int gemm_impl(const v2s_gemm_args* a, void* stream, int row0, int p8_force);

// Entry point.  (gemm_impl can run a row range of a larger problem -- row0 keeps the dropout mask index global; a split of the rows
// into whole rounds of 256 x 256 tiles plus a remainder on the 128 x 128 kernels was measured and did not pay: 132.1 vs 133.1 us
// on 32000 x 2304 x 768, nothing on the N = 768 shapes.)
extern "C" int v2s_gemm(const v2s_gemm_args* a, void* stream) { return gemm_impl(a, stream, 0, -1); }

extern "C" int v2s_gemm_grouped(const v2s_gemm_args* a, int32_t count, const void* const* A, const void* const* B, void* const* C, void* stream) {
  V2S_CHECK(a && A && B && C, V2S_ERR_ARG, "v2s_gemm_grouped: null args");
  V2S_CHECK(count >= 1 && count <= GEMM_GROUP_MAX, V2S_ERR_ARG, "v2s_gemm_grouped: 1..%d problems per call (got %d)", GEMM_GROUP_MAX, count);
  V2S_CHECK(a->transA && a->transB && a->c_dtype == V2S_F32 && !a->bias && !a->act && !a->dact && !a->pre && !a->residual && a->dropout_p == 0.f &&
                a->rms_eps <= 0.f, V2S_ERR_ARG, "v2s_gemm_grouped: weight-gradient form only (transA = transB = 1, fp32 C, plain epilogue)");
  V2S_CHECK(a->M > 0 && a->N > 0 && a->K > 0 && (a->K % BK) == 0 && (a->M % 8) == 0 && (a->N % 8) == 0, V2S_ERR_SHAPE,
            "v2s_gemm_grouped: M, N multiples of 8 and K a multiple of %d (got %d %d %d)", BK, a->M, a->N, a->K);
  V2S_CHECK((a->lda % 8) == 0 && (a->ldb % 8) == 0 && (a->ldc % 8) == 0 && a->lda >= a->M && a->ldb >= a->N && a->ldc >= a->N, V2S_ERR_ALIGN,
            "v2s_gemm_grouped: leading dimensions must be multiples of 8 covering the rows");
  V2S_CHECK(64 * a->lda + a->M < (1L << 31) && 64 * a->ldb + a->N < (1L << 31), V2S_ERR_SHAPE, "v2s_gemm_grouped: operand rows too long for 32-bit lane offsets");
  GemmGrp g;
  GemmP& p = g.p;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.A = nullptr; p.B = nullptr; p.lda = a->lda; p.ldb = a->ldb; p.C = nullptr; p.ldc = a->ldc; p.c_f32 = 1; p.accumulate = a->accumulate;
  p.alpha = a->alpha; p.bias = nullptr; p.act = 0; p.pre = nullptr; p.dact = 0; p.z = nullptr; p.ldz = 0; p.residual = nullptr; p.ldr = 0;
  p.p16 = 0; p.inv_keep = 1.f; p.seed = 0; p.salt = nullptr; p.rms_eps = 0.f;
  p.order = v2s_opt_gemm_order(); p.dbg = 0; p.row0 = 0;
  p.tilesM = (a->M + BM - 1) / BM; p.tilesN = (a->N + BN - 1) / BN; p.splitk = 1; p.kper = a->K; p.ws = nullptr;
  for (int i = 0; i < count; ++i) {
    V2S_CHECK(A[i] && B[i] && C[i] && ((((uintptr_t)A[i] | (uintptr_t)B[i] | (uintptr_t)C[i]) & 15) == 0), V2S_ERR_ALIGN, "v2s_gemm_grouped: problem %d: null or misaligned pointer", i);
    g.A[i] = (const bf16_t*)A[i]; g.B[i] = (const bf16_t*)B[i]; g.C[i] = C[i];
  }
  for (int i = count; i < GEMM_GROUP_MAX; ++i) { g.A[i] = g.A[0]; g.B[i] = g.B[0]; g.C[i] = g.C[0]; }
  g_last_gemm = "gemm_dma_grouped_kernel";
  hipLaunchKernelGGL(gemm_dma_grouped_kernel, dim3((unsigned)(p.tilesM * p.tilesN * count)), dim3(NTHREADS), 0, (hipStream_t)stream, g);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

static int gemm_impl(const v2s_gemm_args* a, void* stream, int row0, int p8_force) {
  V2S_CHECK(a != nullptr, V2S_ERR_ARG, "v2s_gemm: null args");
  V2S_CHECK(a->M > 0 && a->N > 0 && a->K > 0, V2S_ERR_SHAPE, "v2s_gemm: non-positive shape %d %d %d", a->M, a->N, a->K);
  // Ragged sizes (e.g. vocab 32100 or 612): rows are always exact (predicated); a dimension that is walked in
  // 8-element chunks may be ragged only if the caller padded the row pitch to a multiple of 8 (pad content of
  // A must be finite/zero when K is ragged: it is multiplied by zero-filled B rows).
  const long N8 = (a->N + 7) / 8 * 8, K8 = (a->K + 7) / 8 * 8, M8 = (a->M + 7) / 8 * 8;
  V2S_CHECK((a->lda % 8) == 0 && (a->ldb % 8) == 0 && (a->ldc % 8) == 0, V2S_ERR_ALIGN, "v2s_gemm: leading dims must be multiples of 8");
  V2S_CHECK(a->ldc >= N8, V2S_ERR_SHAPE, "v2s_gemm: ldc (%ld) must cover N rounded up to 8 (%ld)", (long)a->ldc, N8);
  // K is walked in 8-element chunks only by NON-transposed operands; transposed ones predicate k exactly
  V2S_CHECK((a->K % 8) == 0 || (a->transB && (a->transA || a->lda >= K8)), V2S_ERR_SHAPE,
            "v2s_gemm: ragged K (%d) needs transB=1 and (transA=1 or lda >= %ld with a zero/finite row pad)", a->K, K8);
  V2S_CHECK(!a->transA || a->lda >= M8, V2S_ERR_SHAPE, "v2s_gemm: transA needs lda >= M rounded up to 8");
  V2S_CHECK(!a->transB || a->ldb >= N8, V2S_ERR_SHAPE, "v2s_gemm: transB needs ldb >= N rounded up to 8");
  V2S_CHECK((a->N % 8) == 0 || (!a->bias && !a->residual && !a->z && !a->pre), V2S_ERR_SHAPE,
            "v2s_gemm: ragged N (%d) is supported for plain epilogues only", a->N);
  V2S_CHECK((((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C) & 15) == 0, V2S_ERR_ALIGN, "v2s_gemm: pointers must be 16-byte aligned");
  V2S_CHECK(a->c_dtype == V2S_BF16 || a->c_dtype == V2S_F32, V2S_ERR_DTYPE, "v2s_gemm: bad c_dtype %d", a->c_dtype);
  V2S_CHECK(!(a->accumulate && a->c_dtype != V2S_F32), V2S_ERR_DTYPE, "v2s_gemm: accumulate needs fp32 C");
  V2S_CHECK(!(a->dact != V2S_ACT_NONE && a->z == nullptr), V2S_ERR_ARG, "v2s_gemm: dact needs z");
  V2S_CHECK(a->dropout_p >= 0.f && a->dropout_p < 1.f, V2S_ERR_ARG, "v2s_gemm: dropout_p out of range");
  V2S_CHECK(!(a->transA && !a->transB), V2S_ERR_ARG, "v2s_gemm: (transA=1, transB=0) is not instantiated");

  GemmP p;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.A = (const bf16_t*)a->A; p.B = (const bf16_t*)a->B; p.lda = a->lda; p.ldb = a->ldb;
  p.C = a->C; p.ldc = a->ldc; p.c_f32 = (a->c_dtype == V2S_F32); p.accumulate = a->accumulate;
  p.alpha = a->alpha; p.bias = a->bias; p.act = a->act; p.pre = (bf16_t*)a->pre; p.dact = a->dact;
  p.z = (const bf16_t*)a->z; p.ldz = a->ldz; p.residual = (const bf16_t*)a->residual; p.ldr = a->ldr;
  p.p16 = (uint32_t)(a->dropout_p * 65536.0f + 0.5f);
  p.inv_keep = p.p16 ? 1.0f / (1.0f - (float)p.p16 / 65536.0f) : 1.0f;
  p.seed = a->dropout_seed;
  p.salt = v2s_seed_salt();
  p.rms_eps = a->rms_eps;
  p.order = v2s_opt_gemm_order();
  p.dbg = v2s_opt_gemm_dbg();
  p.row0 = row0;
  hipStream_t s = (hipStream_t)stream;
  const bool tr = v2s_opt_tr_read() != 0;
  const bool plain_split = a->workspace && (a->c_dtype == V2S_F32 || !a->accumulate) && !a->bias && !a->act && !a->dact && !a->residual && !a->pre &&
                           a->dropout_p == 0.f && a->K >= 1024 && (a->N % 8) == 0;
  // decode-step projections: M = batch x beams rows (<= 512).  Past 64 rows the 49 MB LM head goes back to the general tiles
  // (enough of them there), the layer projections stay here: 256 x 768 is 12 tiles of 128 x 128 on 256 CUs.
  const bool skinny_rows = a->M <= 64 || (a->decode != 0 && a->M <= 512 && a->N < 8192);
  if (skinny_rows && !a->transA && !a->transB && (a->K % 128) == 0 && v2s_opt_gemm_skinny() != 0) {
    if (a->N >= 8192 && (a->K == 512 || a->K == 768 || a->K == 1024) && v2s_opt_gemm_skinny() != 2) {
      p.tilesM = 1; p.tilesN = (a->N + 15) / 16; p.splitk = 1; p.kper = a->K; p.ws = nullptr;
      g_last_gemm = "gemm_skinny_wide_kernel";
      const int ncu = num_cus();
      const int want = (p.tilesN + 7) / 8;
      const dim3 grid((unsigned)(want < ncu ? want : ncu));
      const size_t lds = (size_t)64 * (a->K + 8) * 2 + 64 * 4 + 8 * 16 * 17 * 4;
#define V2S_WIDE(KS_) do { static bool attr = false; \
        if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_skinny_wide_kernel<KS_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        hipLaunchKernelGGL((gemm_skinny_wide_kernel<KS_>), grid, dim3(512), lds, s, p); } while (0)
      if (a->K == 512) V2S_WIDE(16); else if (a->K == 768) V2S_WIDE(24); else V2S_WIDE(32);
#undef V2S_WIDE
      V2S_LAUNCH_CHECK();
      return V2S_OK;
    }
    p.tilesM = 1; p.tilesN = (a->N + 15) / 16; p.splitk = 1; p.kper = a->K; p.ws = nullptr;
#if SKINNY_STAMPS
    p.ws = (float*)a->workspace;
#endif
    g_last_gemm = "gemm_skinny_kernel";
    // row fragments per block: one (grid.y walks them: four times the blocks, a quarter of the activation loads per wave) measured
    // faster on every decoder projection (greedy B = 64: 1.39 ms/step with four fragments per block, 1.17 with one;
    // profiles/r02_decode_ab_skinny_mt.txt).  gemm_skinny = 2 keeps the four-fragment blocks, 4 = one fragment and four waves.
    const int mode = v2s_opt_gemm_skinny();
    int waves = ((a->K % 256) == 0 && mode != 4) ? 8 : 4;
    // row fragments per block: the fewest (most blocks, least activation traffic per wave) that keep the grid within ~4 blocks per CU
    int mt = 1, nt = 1;
    const long cap = mode >= 20000 ? mode - 20000 : 1024;      // A/B (tools/decode_ab.py): gemm_skinny = 20000 + the largest grid taken with one row fragment per block
    while (mt < 4 && (long)p.tilesN * ((a->M + 16 * mt - 1) / (16 * mt)) > cap) mt *= 2;
    if (a->M <= 64 && (mode == 2 || a->N >= 8192)) mt = 4;
    if (a->M > 64) { mt = 2; nt = 2; }       // beam rows: 32 x 32 blocks (4-beam step 2.26 -> 2.03 ms; profiles/r02_decode_ab_skinny_mt.txt)
    // A/B (tools/decode_ab.py): gemm_skinny = <waves><mt><nt> for M > 64, 1<waves><mt><nt> for M <= 64
    if (a->M > 64 && mode >= 100 && mode < 1000) { waves = mode / 100; mt = (mode / 10) % 10; nt = mode % 10; }
    if (a->M <= 64 && a->N < 8192 && mode >= 1000 && mode < 20000) { waves = (mode / 100) % 10; mt = (mode / 10) % 10; nt = mode % 10; }
    if ((a->K % (waves * 32)) != 0) waves = 4;
    const int nsteps = a->K / waves / 32;
    p.tilesN = (a->N + 16 * nt - 1) / (16 * nt);
    const dim3 grid((unsigned)p.tilesN, (unsigned)((a->M + mt * 16 - 1) / (mt * 16)));
    bool launched = true;
#define V2S_SKINNY(W_, S_, MT_, NT_) hipLaunchKernelGGL((gemm_skinny_kernel<W_, S_, MT_, NT_>), grid, dim3(W_ * 64), 0, s, p)
#define V2S_SKINNY_S4(W_, MT_, NT_) do { if (nsteps % 4 == 0) V2S_SKINNY(W_, 4, MT_, NT_); else if (nsteps % 3 == 0) V2S_SKINNY(W_, 3, MT_, NT_); \
      else if (nsteps % 2 == 0) V2S_SKINNY(W_, 2, MT_, NT_); else V2S_SKINNY(W_, 1, MT_, NT_); } while (0)
    if (nt == 1 && mt == 1) {
      if (waves == 8) {
        if (nsteps % 12 == 0) V2S_SKINNY(8, 12, 1, 1); else if (nsteps % 8 == 0) V2S_SKINNY(8, 8, 1, 1); else if (nsteps % 6 == 0) V2S_SKINNY(8, 6, 1, 1);
        else V2S_SKINNY_S4(8, 1, 1);
      } else {
        if (nsteps % 12 == 0) V2S_SKINNY(4, 12, 1, 1); else if (nsteps % 8 == 0) V2S_SKINNY(4, 8, 1, 1); else if (nsteps % 6 == 0) V2S_SKINNY(4, 6, 1, 1);
        else V2S_SKINNY_S4(4, 1, 1);
      }
    } else if (nt == 1 && mt == 2) {
      if (waves == 8) { if (nsteps % 6 == 0) V2S_SKINNY(8, 6, 2, 1); else V2S_SKINNY_S4(8, 2, 1); }
      else { if (nsteps % 6 == 0) V2S_SKINNY(4, 6, 2, 1); else V2S_SKINNY_S4(4, 2, 1); }
    } else if (nt == 1 && mt == 4) {
      if (waves == 8) V2S_SKINNY_S4(8, 4, 1); else V2S_SKINNY_S4(4, 4, 1);
    } else if (nt == 2 && mt == 2) {
      if (waves == 8) { if (nsteps % 6 == 0) V2S_SKINNY(8, 6, 2, 2); else V2S_SKINNY_S4(8, 2, 2); }
      else { if (nsteps % 6 == 0) V2S_SKINNY(4, 6, 2, 2); else V2S_SKINNY_S4(4, 2, 2); }
    } else if (nt == 2 && mt == 4 && waves == 4) {
      V2S_SKINNY_S4(4, 4, 2);
    } else if (nt == 4 && mt == 2 && waves == 4) {
      V2S_SKINNY_S4(4, 2, 4);
    } else if (nt == 2 && mt == 1) {
      if (waves == 8) { if (nsteps % 6 == 0) V2S_SKINNY(8, 6, 1, 2); else V2S_SKINNY_S4(8, 1, 2); } else V2S_SKINNY_S4(4, 1, 2);
    } else launched = false;
    V2S_CHECK(launched, V2S_ERR_ARG, "v2s_gemm: gemm_skinny = %d names a block shape that is not built", mode);
#undef V2S_SKINNY_S4
#undef V2S_SKINNY
    V2S_LAUNCH_CHECK();
    return V2S_OK;
  }
  V2S_CHECK(a->rms_eps <= 0.f, V2S_ERR_ARG, "v2s_gemm: rms_eps (fused RMSNorm) is only available on the decode path (M <= 64, or decode = 1 and M <= 512 with N < 8192; K %% 128 == 0)");
  // tile choice: the 256-row kernel (one 8-wave block per CU) when K % 64 == 0 and it yields enough tiles, else 128x128
  int bm = BM, bn = BN;
  const int big_mode = v2s_opt_gemm_big();
  // measured on the step's shapes with every kernel on the LDS-DMA loop (tools/gemm_tile_ab.py, us, 256-row tiles vs 128x128 DMA):
  // NT 32000x3072x768 190 vs 199, LM head 8192x32200x768 498 vs 530 -> 256x256; but 32000x768x3072 203 vs 179, 32000x768x768 63 vs 59,
  // 35200x1536x768 109 vs 101, 8192x2304x768 54 vs 43, 8192x3072x768 63 vs 53, 32000x2304x768 137 vs 139 -> 128x128 (two blocks per
  // CU, twice as many tiles: far less tail quantisation).  So plain NT takes the 256x256 tile only for wide outputs with >= 1200 tiles
  // (>= 4.7 rounds of the chip), dgrad (transB) likewise (32000x3072x768 261 vs 268, but 8192x3072x768 74 vs 60); wide long-K weight
  // gradients keep it.
  if (big_mode && tr && (a->K % BK) == 0 && a->M >= 256 && a->N >= 128) {
    // weight gradients with a short contraction keep the 128x128 tiles even for wide outputs (768x2048x3200: 359 vs 270 TF/s)
    const bool wide = a->N >= 1024 && big_mode != 2 && !(a->transA && a->K < 8192);
    const int bn2 = wide ? 256 : 128;
    const long t2 = (long)((a->M + 255) / 256) * ((a->N + bn2 - 1) / bn2);
    const bool transposed = a->transA || a->transB;
    // forward and dgrad: wide output AND enough 256x256 tiles (dgrad 8192x3072x768: 74 vs 60 us -> 128x128); weight gradients: wide
    // (a dgrad whose epilogue reads the activation mask stays on the 128x128 tiles: 32000x3072x768 with the ReLU mask 245 vs 227 us)
    const bool use256 = a->transA ? wide : (wide && t2 >= 1200 && a->dact == V2S_ACT_NONE);
    if (big_mode == 2) { if (t2 >= 240) { bm = 256; bn = bn2; } }
    else if (use256 && (t2 >= 240 || (plain_split && t2 >= 8))) { bm = 256; bn = bn2; }
  }
  // 4-wave 256x128x32 kernel, two blocks per CU.  With the split-K cost model below, measured (tools/gemm_bench.py wgrad, TF/s,
  // w4 / 128x128 DMA / 8-wave 256-row): qkv 2304x768x32000 756/678/-, wi 3072x768x32000 798/761/-, o 768x768x32000 476/545/-,
  // wo 768x3072x32000 865/693/871, ViT K=3200 shapes 290/389/-, decoder K=8192 shapes 553-625/656-723/-, cross-kv 680/837/-:
  // w4 only for the big-output long-contraction weight gradients, never for NT/dgrad shapes (see the NT table above)
  bool w4 = false;
  const bool w4_ok = tr && (a->K % 32) == 0 && a->M >= 128 && a->N >= 64;
  if (w4_ok && (big_mode >= 3 || (big_mode == 1 && a->transA && a->transB && a->N < 1024 &&
                                  (long)a->M * a->N >= 2304L * 768L && a->K >= 16384))) { w4 = true; bm = 256; bn = 128; }
  // 8-phase ping-pong kernel (p8_decide): 256 x 256 | 256 x 128 tiles, synchronous or deferred epilogue
  int p8 = 0;
  bool p8d = false;
  if (p8_force != 0) p8_decide(a, tr, p8, p8d);
  if (p8) { bm = 256; bn = p8; w4 = false; }
  // 4-wave asm-scheduled 256 x 256 kernel (gemm_a4_kernel): forward / dgrad shapes, any epilogue, never split-K
  bool a4 = false, a4p = false;
  {
    const int amode = v2s_opt_gemm_a4();
    const long t256 = (long)((a->M + 255) / 256) * ((a->N + 255) / 256);
    const bool a_ok = amode != 0 && p8_force != 0 && tr && (a->K % 128) == 0 && a->M >= 256 && a->N >= 256 && (a->N % 8) == 0 &&
                      (a->transA ? (a->M % 8) == 0 : !(plain_split && t256 < 512)) &&
                      (a->transA ? 32 * a->lda + a->M : (long)a->M * a->lda) < (1L << 30) &&
                      (a->transB ? 32 * a->ldb + a->N : (long)a->N * a->ldb) < (1L << 30);
    if (a_ok && (amode == 2 || amode == 3 || ((amode == 1 || amode == 4 || amode == 5) && v2s_opt_gemm_p8() == 1 && a4_auto(a, t256)))) {
      a4 = true; bm = 256; bn = 256; p8 = 0; p8d = false; w4 = false;
      // persistent form with the deferred write-out: plain bf16 epilogue, whole tiles (gemm_a4 = 3: never)
      a4p = amode != 3 && a4p_epilogue(a) != 0 && !plain_split && (long)a->M * a->ldc * 2 < (1L << 31) && t256 < 65536;
    }
  }
  p.tilesM = (a->M + bm - 1) / bm; p.tilesN = (a->N + bn - 1) / bn;
  // split-K: weight-gradient GEMMs have few output tiles (768x768 -> 36) but a huge contraction (all tokens);
  // slice K so that the chip is filled.  Only for fp32 outputs with a plain epilogue; partials go to the workspace.
  p.splitk = 1; p.kper = a->K; p.ws = (float*)a->workspace;
  const int tiles = p.tilesM * p.tilesN;
  const int want = (bm == 256) ? 512 : 768;
  if (plain_split && tiles < want) {
    int sp = (want + tiles - 1) / tiles;
    const int maxs = a->K / 512;
    if (v2s_opt_gemm_split() != 0) {
      // all blocks of a split GEMM are equally long, so time ~ rounds x (K/sp + per-block overhead) + reduction(sp): pick the
      // slice count that minimises it (block slots: 256 for the one-block-per-CU kernels, 512 for the two-per-CU ones)
      const int slots = (bm == 256 && !w4) ? 256 : 512;
      long best = -1;
      for (int c = 1; c <= maxs; ++c) {
        const long rounds = ((long)tiles * c + slots - 1) / slots;
        const long kp = ((a->K + c - 1) / c + BK - 1) / BK * BK;
        const long cost = rounds * (kp + 384) + 16L * c;
        if (best < 0 || cost < best) { best = cost; sp = c; }
      }
    }
    if (sp > maxs) sp = maxs;
    const long per_slice = (long)a->M * a->N * 4;
    if ((long)sp * per_slice > a->workspace_bytes) sp = (int)(a->workspace_bytes / per_slice);
    if (sp > 1) {
      const int kq = a4 ? 128 : BK;                     // the a4 loop walks K in iterations of 128
      int kper = ((a->K + sp - 1) / sp + kq - 1) / kq * kq;
      p.kper = kper;
      p.splitk = (a->K + kper - 1) / kper;
      p.alpha = 1.0f;           // alpha and the accumulate are applied by the reduction
    }
  }
  const unsigned nblocks = (unsigned)(p.tilesM * p.tilesN * p.splitk);
  if (a4p && p.splitk == 1) {
    static bool attr_a4p = false;
    if (!attr_a4p) {
      (void)hipFuncSetAttribute((const void*)gemm_a4p_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, A4P_LDS);
      (void)hipFuncSetAttribute((const void*)gemm_a4p_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, A4P_LDS);
      (void)hipFuncSetAttribute((const void*)gemm_a4p_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, A4P_LDS);
      (void)hipFuncSetAttribute((const void*)gemm_a4p_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, A4P_LDS);
      (void)hipFuncSetAttribute((const void*)gemm_a4p_kernel<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, A4P_LDS);
      attr_a4p = true;
    }
    int ncu = num_cus();
    const int nt = p.tilesM * p.tilesN;
    if (v2s_opt_gemm_a4_grid() > 0 && v2s_opt_gemm_a4_grid() < ncu) ncu = v2s_opt_gemm_a4_grid();
    if (v2s_opt_gemm_a4_grid() < 0 && nt > ncu) {       // balanced: the fewest blocks that finish in the same number of rounds (375 tiles: 188 blocks x 2 instead of
      const int rounds = (nt + ncu - 1) / ncu;          // 119 x 2 + 137 x 1; the other CUs are left to the concurrent streams)
      ncu = (nt + rounds - 1) / rounds;
    }
    const dim3 grid((unsigned)(nt < ncu ? nt : ncu)), block(256);
    // tile walk of the persistent kernel: row-major while W (N x K) stays in an XCD's L2, groups of 4 tile rows beyond (option gemm_a4_walk forces GM)
    p.order = v2s_opt_gemm_a4_walk() > 0 ? v2s_opt_gemm_a4_walk() : (p.tilesN >= 16 ? 4 : 1);
    const int epi = a4p_epilogue(a);
    g_last_gemm = epi == 2 ? "gemm_a4p_kernel<true, 1>" : epi == 3 ? "gemm_a4p_kernel<false, 2>" : epi == 4 ? "gemm_a4p_kernel<false, 3>" :
                  (a->transB ? "gemm_a4p_kernel<true, 0>" : "gemm_a4p_kernel<false, 0>");
    if (epi == 2) hipLaunchKernelGGL((gemm_a4p_kernel<true, 1>), grid, block, A4P_LDS, s, p);
    else if (epi == 3) hipLaunchKernelGGL((gemm_a4p_kernel<false, 2>), grid, block, A4P_LDS, s, p);
    else if (epi == 4) hipLaunchKernelGGL((gemm_a4p_kernel<false, 3>), grid, block, A4P_LDS, s, p);
    else if (a->transB) hipLaunchKernelGGL((gemm_a4p_kernel<true, 0>), grid, block, A4P_LDS, s, p);
    else hipLaunchKernelGGL((gemm_a4p_kernel<false, 0>), grid, block, A4P_LDS, s, p);
  } else if (a4) {
    static bool attr_a4 = false;
    if (!attr_a4) {
      (void)hipFuncSetAttribute((const void*)gemm_a4_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, A4_LDS);
      (void)hipFuncSetAttribute((const void*)gemm_a4_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, A4_LDS);
      (void)hipFuncSetAttribute((const void*)gemm_a4_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, A4_LDS);
      attr_a4 = true;
    }
    const dim3 grid(nblocks), block(256);
    g_last_gemm = a->transA ? "gemm_a4_kernel<true, true>" : (a->transB ? "gemm_a4_kernel<false, true>" : "gemm_a4_kernel<false, false>");
    if (a->transA) hipLaunchKernelGGL((gemm_a4_kernel<true, true>), grid, block, A4_LDS, s, p);
    else if (a->transB) hipLaunchKernelGGL((gemm_a4_kernel<false, true>), grid, block, A4_LDS, s, p);
    else hipLaunchKernelGGL((gemm_a4_kernel<false, false>), grid, block, A4_LDS, s, p);
  } else if (p8d) {
    static bool attr8d = false;
    const int ncu = num_cus();
    if (!attr8d) {
      (void)hipFuncSetAttribute((const void*)gemm_p8d_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
      (void)hipFuncSetAttribute((const void*)gemm_p8d_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
      attr8d = true;
    }
    const int nt = p.tilesM * p.tilesN;
    const dim3 grid((unsigned)(nt < ncu ? nt : ncu)), block(512);
    g_last_gemm = a->transB ? "gemm_p8d_kernel<false, true>" : "gemm_p8d_kernel<false, false>";
    if (p.dbg == 3 && !a->transB) {                 // ablation build of the NT variant: 3 = no drain, 4 = no write-out at all (results invalid)
      (void)hipFuncSetAttribute((const void*)gemm_p8d_kernel<false, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
      (void)hipFuncSetAttribute((const void*)gemm_p8d_kernel<false, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
      (void)hipFuncSetAttribute((const void*)gemm_p8d_kernel<false, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
      hipLaunchKernelGGL((gemm_p8d_kernel<false, false, 1>), grid, block, 163840, s, p);
    } else if (!a->transB) hipLaunchKernelGGL((gemm_p8d_kernel<false, false>), grid, block, 163840, s, p);
    else hipLaunchKernelGGL((gemm_p8d_kernel<false, true>), grid, block, 163840, s, p);
  } else if (p8) {
    static bool attr8 = false;
    if (!attr8) {
      (void)hipFuncSetAttribute((const void*)gemm_p8_kernel<false, false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, P8<256>::LDS_BYTES);
      (void)hipFuncSetAttribute((const void*)gemm_p8_kernel<false, true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, P8<256>::LDS_BYTES);
      (void)hipFuncSetAttribute((const void*)gemm_p8_kernel<true, true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, P8<256>::LDS_BYTES);
      (void)hipFuncSetAttribute((const void*)gemm_p8_kernel<false, false, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, P8<128>::LDS_BYTES);
      (void)hipFuncSetAttribute((const void*)gemm_p8_kernel<false, true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, P8<128>::LDS_BYTES);
      (void)hipFuncSetAttribute((const void*)gemm_p8_kernel<true, true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, P8<128>::LDS_BYTES);
      attr8 = true;
    }
    const dim3 grid(nblocks), block(512);
    static const char* names8[2][3] = {{"gemm_p8_kernel<false, false, 256>", "gemm_p8_kernel<false, true, 256>", "gemm_p8_kernel<true, true, 256>"},
                                       {"gemm_p8_kernel<false, false, 128>", "gemm_p8_kernel<false, true, 128>", "gemm_p8_kernel<true, true, 128>"}};
    g_last_gemm = names8[p8 == 256 ? 0 : 1][!a->transB ? 0 : (!a->transA ? 1 : 2)];
#define V2S_P8(TA_, TB_)                                                                                               \
    do {                                                                                                               \
      if (p8 == 256) hipLaunchKernelGGL((gemm_p8_kernel<TA_, TB_, 256>), grid, block, P8<256>::LDS_BYTES, s, p);       \
      else hipLaunchKernelGGL((gemm_p8_kernel<TA_, TB_, 128>), grid, block, P8<128>::LDS_BYTES, s, p);                 \
    } while (0)
    if (!a->transA && !a->transB) V2S_P8(false, false);
    else if (!a->transA && a->transB) V2S_P8(false, true);
    else V2S_P8(true, true);
#undef V2S_P8
  } else if (w4) {
    static bool attr4 = false;
    if (!attr4) {
      (void)hipFuncSetAttribute((const void*)gemm_w4_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
      (void)hipFuncSetAttribute((const void*)gemm_w4_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
      (void)hipFuncSetAttribute((const void*)gemm_w4_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
      attr4 = true;
    }
    const dim3 grid(nblocks), block(256);
    g_last_gemm = !a->transB ? "gemm_w4_kernel<false, false>" : (!a->transA ? "gemm_w4_kernel<false, true>" : "gemm_w4_kernel<true, true>");
    if (!a->transA && !a->transB) hipLaunchKernelGGL((gemm_w4_kernel<false, false>), grid, block, 73728, s, p);
    else if (!a->transA && a->transB) hipLaunchKernelGGL((gemm_w4_kernel<false, true>), grid, block, 73728, s, p);
    else hipLaunchKernelGGL((gemm_w4_kernel<true, true>), grid, block, 73728, s, p);
  } else if (bm == 256) {
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute((const void*)gemm_big_kernel<false, false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      (void)hipFuncSetAttribute((const void*)gemm_big_kernel<false, true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      (void)hipFuncSetAttribute((const void*)gemm_big_kernel<true, true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      (void)hipFuncSetAttribute((const void*)gemm_big_kernel<false, false, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
      (void)hipFuncSetAttribute((const void*)gemm_big_kernel<false, true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
      (void)hipFuncSetAttribute((const void*)gemm_big_kernel<true, true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
      attr_done = true;
    }
    const size_t dyn = 2 * (size_t)(256 * 64 * 2 + bn * 64 * 2);
    const dim3 grid(nblocks), block(512);
    {
      static const char* names[2][3] = {{"gemm_big_kernel<false, false, 256>", "gemm_big_kernel<false, true, 256>", "gemm_big_kernel<true, true, 256>"},
                                        {"gemm_big_kernel<false, false, 128>", "gemm_big_kernel<false, true, 128>", "gemm_big_kernel<true, true, 128>"}};
      g_last_gemm = names[bn == 256 ? 0 : 1][!a->transB ? 0 : (!a->transA ? 1 : 2)];
    }
#define V2S_BIG(TA_, TB_)                                                                              \
    do {                                                                                               \
      if (bn == 256) hipLaunchKernelGGL((gemm_big_kernel<TA_, TB_, 256>), grid, block, dyn, s, p);     \
      else hipLaunchKernelGGL((gemm_big_kernel<TA_, TB_, 128>), grid, block, dyn, s, p);               \
    } while (0)
    if (!a->transA && !a->transB) V2S_BIG(false, false);
    else if (!a->transA && a->transB) V2S_BIG(false, true);
    else V2S_BIG(true, true);
#undef V2S_BIG
  } else {
  const dim3 grid(nblocks), block(NTHREADS);
  // measured: the DMA loop wins for the transposed-operand variants (dgrad +5..18 %, wgrad +2..9 %) and, since the grouped tile walk,
  // for plain NT shapes too (tools/gemm_dma_ab.py: 8192x768x3072 61 -> 49 us, 3200x768x2048 35 -> 30 us, 8192x768x768 24 -> 21 us)
  const bool dma = v2s_opt_gemm_dma() != 0 && tr && (a->transA || a->transB || v2s_opt_gemm_dma() == 2) && (a->K % BK) == 0 && (p.kper % BK) == 0 &&
                   a->M >= 8 && a->N >= 8 &&    // gemm_dma: 2 (default) = every variant, 1 = transposed-operand variants only
                   (a->transA ? 64 * a->lda + a->M : (long)a->M * a->lda) < (1L << 31) &&     // per-lane source offsets are 32-bit byte offsets
                   (a->transB ? 64 * a->ldb + a->N : (long)a->N * a->ldb) < (1L << 31);
  if (dma) g_last_gemm = !a->transB ? "gemm_dma_kernel<false, false>" : (!a->transA ? "gemm_dma_kernel<false, true>" : "gemm_dma_kernel<true, true>");
  else g_last_gemm = !a->transB ? "gemm_kernel<false, false, true>" : (!a->transA ? (tr ? "gemm_kernel<false, true, true>" : "gemm_kernel<false, true, false>")
                                                                                    : (tr ? "gemm_kernel<true, true, true>" : "gemm_kernel<true, true, false>"));
  if (dma) {
    if (!a->transA && !a->transB) hipLaunchKernelGGL((gemm_dma_kernel<false, false>), grid, block, 0, s, p);
    else if (!a->transA && a->transB) hipLaunchKernelGGL((gemm_dma_kernel<false, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_dma_kernel<true, true>), grid, block, 0, s, p);
  } else if (!a->transA && !a->transB) {
    hipLaunchKernelGGL((gemm_kernel<false, false, true>), grid, block, 0, s, p);
  } else if (!a->transA && a->transB) {
    if (tr) hipLaunchKernelGGL((gemm_kernel<false, true, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_kernel<false, true, false>), grid, block, 0, s, p);
  } else {
    if (tr) hipLaunchKernelGGL((gemm_kernel<true, true, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_kernel<true, true, false>), grid, block, 0, s, p);
  }
  }
  V2S_LAUNCH_CHECK();
  if (p.splitk > 1) {
    long blocks = ((long)a->M * (a->N / 4) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p.ws, (float*)a->C, (long)a->ldc, a->M, a->N,
                       p.splitk, a->alpha, a->accumulate, a->c_dtype == V2S_BF16 ? 1 : 0);
    V2S_LAUNCH_CHECK();
  }
  return V2S_OK;
}

// ---- tied LM head + label-smoothed cross entropy, logits never written (lmhead_ce_kernel above) --------------------------------------------------
namespace {
__global__ __launch_bounds__(1024) void lmhead_reduce_kernel(const float* __restrict__ row_out, const long* __restrict__ labels, int rows,
                                                             float* __restrict__ loss_sum, float* __restrict__ count) {
  __shared__ float a[1024], c[1024];      // deterministic single-block reduction (same order as v2s_ce_fwd's)
  float s = 0.f, n = 0.f;
  for (int r = threadIdx.x; r < rows; r += 1024)
    if (labels[r] >= 0) { s += row_out[r * 2 + 1]; n += 1.f; }
  a[threadIdx.x] = s; c[threadIdx.x] = n;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) { a[threadIdx.x] += a[threadIdx.x + o]; c[threadIdx.x] += c[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { *loss_sum += a[0]; *count += c[0]; }
}
int lmhead_fill(GemmP& p, const char* who, const void* h, int64_t ldh, const void* E, int32_t rows, int32_t V, int32_t Vpad, int32_t d, float alpha) {
  V2S_CHECK(h && E && rows > 0 && V > 0 && Vpad >= V && (Vpad % 8) == 0 && d >= 64 && (d % 64) == 0 && (ldh % 8) == 0 && ldh >= d, V2S_ERR_SHAPE,
            "%s: rows=%d V=%d Vpad=%d (multiple of 8, >= V) d=%d (multiple of 64) ldh=%ld", who, rows, V, Vpad, d, (long)ldh);
  V2S_CHECK((long)rows * ldh < (1L << 30) && (long)Vpad * d < (1L << 30), V2S_ERR_SHAPE, "%s: operand beyond the 32-bit lane offsets of the LDS-DMA loop", who);
  p = GemmP{};
  p.M = rows; p.N = Vpad; p.K = d;
  p.A = (const bf16_t*)h; p.B = (const bf16_t*)E; p.lda = ldh; p.ldb = d;
  p.alpha = alpha;
  p.tilesM = (rows + BM - 1) / BM; p.tilesN = (Vpad + BN - 1) / BN; p.splitk = 1; p.kper = d;
  p.order = 8;            // grouped walk, 8 tile rows deep: the blocks an XCD runs together share a few E column slabs in its L2
  return V2S_OK;
}
}  // namespace

extern "C" int64_t v2s_lmhead_ce_workspace_floats(int32_t rows, int32_t Vpad) {
  return (int64_t)rows * ((Vpad + BN - 1) / BN) * 2 * 4;
}

extern "C" int v2s_lmhead_ce_fwd(const void* h, int64_t ldh, const void* E, int32_t rows, int32_t V, int32_t Vpad, int32_t d, float alpha,
                                 const int64_t* labels, float eps, float* part, float* row_out, float* loss_sum, float* count, void* stream) {
  GemmP p;
  if (int e = lmhead_fill(p, "v2s_lmhead_ce_fwd", h, ldh, E, rows, V, Vpad, d, alpha)) return e;
  V2S_CHECK(labels && part && row_out && loss_sum && count && ((uintptr_t)part & 15) == 0, V2S_ERR_ARG, "v2s_lmhead_ce_fwd: labels / part (16-byte aligned) / row_out / loss_sum / count");
  HeadX x{};
  x.V = V; x.tilesN = p.tilesN; x.labels = (const long*)labels; x.part = (float4*)part; x.eps = eps;
  hipStream_t s = (hipStream_t)stream;
  g_last_gemm = "lmhead_ce_kernel<0>";
  hipLaunchKernelGGL((lmhead_ce_kernel<0>), dim3((unsigned)(p.tilesM * p.tilesN)), dim3(NTHREADS), 0, s, p, x);
  V2S_LAUNCH_CHECK();
  hipLaunchKernelGGL(lmhead_finish_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const float4*)part, (const long*)labels, rows, p.tilesN, V, eps, row_out);
  V2S_LAUNCH_CHECK();
  hipLaunchKernelGGL(lmhead_reduce_kernel, dim3(1), dim3(1024), 0, s, (const float*)row_out, (const long*)labels, rows, loss_sum, count);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_lmhead_ce_bwd(const void* h, int64_t ldh, const void* E, int32_t rows, int32_t V, int32_t Vpad, int32_t d, float alpha,
                                 const int64_t* labels, const float* row_out, float eps, const float* gscale, void* dlogits, int64_t ldd, void* stream) {
  GemmP p;
  if (int e = lmhead_fill(p, "v2s_lmhead_ce_bwd", h, ldh, E, rows, V, Vpad, d, alpha)) return e;
  V2S_CHECK(labels && row_out && gscale && dlogits && (ldd % 8) == 0 && ldd >= Vpad, V2S_ERR_ARG, "v2s_lmhead_ce_bwd: labels / row_out / gscale / dlogits, ldd (%ld) a multiple of 8 >= Vpad", (long)ldd);
  HeadX x{};
  x.V = V; x.tilesN = p.tilesN; x.labels = (const long*)labels; x.row = row_out; x.eps = eps; x.gscale = gscale; x.dl = (bf16_t*)dlogits; x.ldd = ldd;
  g_last_gemm = "lmhead_ce_kernel<1>";
  hipLaunchKernelGGL((lmhead_ce_kernel<1>), dim3((unsigned)(p.tilesM * p.tilesN)), dim3(NTHREADS), 0, (hipStream_t)stream, p, x);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_colsum(const void* X, int64_t ldx, int32_t M, int32_t N, float* out, int32_t accumulate,
                          void* stream) {
  V2S_CHECK(M > 0 && N > 0 && (N % 8) == 0 && (ldx % 8) == 0, V2S_ERR_SHAPE, "v2s_colsum: bad shape M=%d N=%d", M, N);
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) {
    if (hipMemsetAsync(out, 0, sizeof(float) * (size_t)N, s) != hipSuccess) { v2s_set_error("v2s_colsum: memset failed"); return V2S_ERR_LAUNCH; }
  }
  const int rows_per_block = 1024;
  const dim3 grid((N + 63) / 64, (M + rows_per_block - 1) / rows_per_block), block(256);
  hipLaunchKernelGGL(colsum_kernel, grid, block, 0, s, (const bf16_t*)X, (long)ldx, M, N, out, rows_per_block);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}

extern "C" int v2s_scale_cols(const void* W, const float* w, void* out, int32_t rows, int32_t cols, void* stream) {
  V2S_CHECK(W && w && out && rows > 0 && cols > 0 && (cols % 8) == 0, V2S_ERR_SHAPE, "v2s_scale_cols: bad shape %d x %d", rows, cols);
  const long total8 = (long)rows * (cols / 8);
  long blocks = (total8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scale_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)W, w, (bf16_t*)out, total8,
                     cols / 8);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}
