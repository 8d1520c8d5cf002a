// gemm_a4_kernel: 256 x 256 x 32 block tile, FOUR waves (one per SIMD), 128 x 128 wave tiles = 256 fp32 accumulators in AGPRs,
// v_mfma_f32_32x32x16_bf16, and a K loop that is ONE inline-asm statement scheduled by gen_gemm_a4.py (included below as
// v2s_gemm_a4.inc; the generator's header describes the loop, the LDS images and the hazard protocol; tools/a4_emu.py executes the
// generated text on a functional model with adversarial completion semantics, tests/test_gemm_a4_emu.py).
// Included by v2s_gemm.hip inside its anonymous namespace (GemmP, tile_coords, epilogue_chunk, lds_addr).
//
// Why this shape: bytes per flop.  The global -> LDS path moves <= 32 B per clock and CU; a 128 x 128 block tile needs
// twice that per MFMA clock and stops at ~1.0 PF whatever its pipeline looks like, 256 x 256 is the first tile that does not.  One wave
// per SIMD with 128 x 128 per wave reads a third fewer fragment bytes per MFMA than 128 x 64 (LDS reads are the second power sink after
// the matrix pipe), and 32 x 32 x 16 is the MFMA shape a single wave issues at the full rate (tools/ubench/mfma_rate.hip).
// Replaces nn.Linear forward / dgrad GEMMs: model/modeling_t5.py:304-311,528-536,581; model/vit.py:41,53,17,20.
//
// Epilogue: the asm statement leaves the accumulators in a[0:255] (literal registers, all 256 named as clobbers); four dump statements
// (A4_DUMP_0..3: 16 ds_write_b128 straight from AGPRs) move one 32-row block row of every wave into an fp32 staging block in the (then
// idle) ring, and the library's ordinary 8-wide row-chunk epilogue (epilogue_chunk: bias / activation / mask / dropout / residual / fp32
// or bf16 store / split-K slice) runs on it.  The compiler never sees the accumulators; build.sh fails the build if it ever emits a
// v_accvgpr_* or scratch access of its own in this kernel (it could only be a spill into our registers).
#ifndef A4_INC
#define A4_INC "v2s_gemm_a4.inc"      // (tools/build_a4_ablations.sh substitutes experimental schedules)
#endif
#include A4_INC

constexpr int A4_LDS = 4 * 32768;                 // ring of 4 stages; the epilogue's 64 x 260 fp32 staging block (66 560 B) lives in it
constexpr int A4_PB = 260;                        // staging pitch in floats (1040 B: eight consecutive rows of a ds_write_b128 pass cover all banks)

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 1) void gemm_a4_kernel(const GemmP p) {
  static_assert(!TA || TB, "(transA, !transB) is not generated");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int nwg = p.tilesM * p.tilesN * p.splitk;
  int tm, tn, slice;
  tile_coords(p, xcd_remap(blockIdx.x, nwg), tm, tn, slice);
  const int m0 = tm * 256, n0 = tn * 256;
  const int kbeg = slice * p.kper, kend = min(p.K, kbeg + p.kper);
  const uint32_t lds = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const char* pa = reinterpret_cast<const char*>(p.A) + (TA ? (long)kbeg * p.lda * 2 : (long)kbeg * 2);
  const char* pb = reinterpret_cast<const char*>(p.B) + (TB ? (long)kbeg * p.ldb * 2 : (long)kbeg * 2);
  const uint32_t pa0 = (uint32_t)(uintptr_t)pa, pa1 = (uint32_t)((uintptr_t)pa >> 32);
  const uint32_t pb0 = (uint32_t)(uintptr_t)pb, pb1 = (uint32_t)((uintptr_t)pb >> 32);
  const uint32_t lda = (uint32_t)(p.lda * 2), ldb = (uint32_t)(p.ldb * 2);
  const uint32_t mmax = TA ? (uint32_t)(((p.M + 7) & ~7) - 8) : (uint32_t)(p.M - 1);
  const uint32_t nmax = TB ? (uint32_t)(((p.N + 7) & ~7) - 8) : (uint32_t)(p.N - 1);
  const uint32_t niter = (uint32_t)((kend - kbeg) / 128);
  if constexpr (TA) {
    asm volatile(A4_MAIN_TN
                 :
                 : [tid] "v"(tid), [pa0] "s"(pa0), [pa1] "s"(pa1), [pb0] "s"(pb0), [pb1] "s"(pb1), [lda] "s"(lda), [ldb] "s"(ldb),
                   [m0] "s"(m0), [n0] "s"(n0), [mmax] "s"(mmax), [nmax] "s"(nmax), [niter] "s"(niter), [lds] "s"(lds)
                 : A4_CLOBBERS);
  } else if constexpr (TB) {
    asm volatile(A4_MAIN_NN
                 :
                 : [tid] "v"(tid), [pa0] "s"(pa0), [pa1] "s"(pa1), [pb0] "s"(pb0), [pb1] "s"(pb1), [lda] "s"(lda), [ldb] "s"(ldb),
                   [m0] "s"(m0), [n0] "s"(n0), [mmax] "s"(mmax), [nmax] "s"(nmax), [niter] "s"(niter), [lds] "s"(lds)
                 : A4_CLOBBERS);
  } else {
#ifdef V2S_A4_ABLATIONS      // profiling build only (tools/build_a4_ablations.sh): gemm_dbg = 11..14 selects an ablated main loop (results invalid)
#define A4_ABL_ARGS : [tid] "v"(tid), [pa0] "s"(pa0), [pa1] "s"(pa1), [pb0] "s"(pb0), [pb1] "s"(pb1), [lda] "s"(lda), [ldb] "s"(ldb), \
                   [m0] "s"(m0), [n0] "s"(n0), [mmax] "s"(mmax), [nmax] "s"(nmax), [niter] "s"(niter), [lds] "s"(lds) : A4_CLOBBERS
    if (p.dbg == 11) { asm volatile(A4_MAIN_NT_NODMA : A4_ABL_ARGS); return; }
    if (p.dbg == 12) { asm volatile(A4_MAIN_NT_NOREAD : A4_ABL_ARGS); return; }
    if (p.dbg == 13) { asm volatile(A4_MAIN_NT_NONE : A4_ABL_ARGS); return; }
    if (p.dbg == 14) { asm volatile(A4_MAIN_NT_NOBAR : A4_ABL_ARGS); return; }
    if (p.dbg == 15) { asm volatile(A4_MAIN_NT_NOWAIT : A4_ABL_ARGS); return; }
#endif
    asm volatile(A4_MAIN_NT
                 :
                 : [tid] "v"(tid), [pa0] "s"(pa0), [pa1] "s"(pa1), [pb0] "s"(pb0), [pb1] "s"(pb1), [lda] "s"(lda), [ldb] "s"(ldb),
                   [m0] "s"(m0), [n0] "s"(n0), [mmax] "s"(mmax), [nmax] "s"(nmax), [niter] "s"(niter), [lds] "s"(lds)
                 : A4_CLOBBERS);
  }
  // every DMA of the block has landed, every wave is past its last fragment read (the statement ends with vmcnt(0) lgkmcnt(0) + s_barrier)
  if (p.dbg == 2) return;                         // ablation: main loop only (results invalid)

  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  // lane (h, m) of block (bi, bj) holds row m, columns 32 bj + 16 h + r: staging row wm * 32 + m of pass bi, column wn * 128 + 32 bj + 16 h + r
  const uint32_t sa = lds + (uint32_t)((wm * 32 + (lane & 31)) * (A4_PB * 4) + (wn * 128 + 16 * (lane >> 5)) * 4);
  const float* cs = reinterpret_cast<const float*>(smem);
  // one global operand (the ReLU / GELU mask source z, or the residual): the chunks of pass ps + 1 are requested before the write-out of
  // pass ps (those of pass 0 before the first dump), see gemm_epilogue
  const bf16_t* gsrc = p.dact != V2S_ACT_NONE ? p.z : p.residual;
  const long gld = p.dact != V2S_ACT_NONE ? p.ldz : p.ldr;
  const bool ahead = gsrc != nullptr && p.dbg == 0 && p.splitk == 1;
  uint4 gop[8], gnext[8];
  auto fetch = [&](int ps, uint4 (&g)[8]) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int c = tid + it * 256;
      const int lr = c >> 5, cc = (c & 31) * 8;
      const int gm = m0 + (lr >> 5) * 128 + ps * 32 + (lr & 31), gn = n0 + cc;
      g[it] = (gm < p.M && gn < p.N) ? *reinterpret_cast<const uint4*>(gsrc + (long)gm * gld + gn) : make_uint4(0, 0, 0, 0);
    }
  };
  if (ahead) fetch(0, gop);
  // the four passes are a ROLLED loop (one copy of the chunk code: the unrolled form was 32 copies of epilogue_chunk, ~250 KiB of
  // instructions every block would fetch once); only the dump statement, which names literal AGPRs, is selected per pass
#pragma unroll 1
  for (int ps = 0; ps < 4; ++ps) {
    if (ps == 0) asm volatile(A4_DUMP_0 : : [sa] "v"(sa) : "memory");
    else if (ps == 1) asm volatile(A4_DUMP_1 : : [sa] "v"(sa) : "memory");
    else if (ps == 2) asm volatile(A4_DUMP_2 : : [sa] "v"(sa) : "memory");
    else asm volatile(A4_DUMP_3 : : [sa] "v"(sa) : "memory");
    __syncthreads();
    if (ahead) {
      if (ps + 1 < 4) fetch(ps + 1, gnext);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int c = tid + it * 256;
        const int lr = c >> 5, cc = (c & 31) * 8;
        const int gm = m0 + (lr >> 5) * 128 + ps * 32 + (lr & 31), gn = n0 + cc;
        if (gm < p.M && gn < p.N) {
          float v[8];
          const float4 x0 = *reinterpret_cast<const float4*>(cs + lr * A4_PB + cc);
          const float4 x1 = *reinterpret_cast<const float4*>(cs + lr * A4_PB + cc + 4);
          v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
          epilogue_chunk<true>(p, v, gm, gn, slice, gop[it]);
        }
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) gop[it] = gnext[it];
    } else {
#pragma unroll 1
      for (int it = 0; it < 8; ++it) {
        const int c = tid + it * 256;
        const int lr = c >> 5, cc = (c & 31) * 8;
        const int gm = m0 + (lr >> 5) * 128 + ps * 32 + (lr & 31), gn = n0 + cc;
        if (gm >= p.M || gn >= p.N) continue;
        float v[8];
        const float4 x0 = *reinterpret_cast<const float4*>(cs + lr * A4_PB + cc);
        const float4 x1 = *reinterpret_cast<const float4*>(cs + lr * A4_PB + cc + 4);
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
        if (p.dbg == 1) {
          asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
          continue;
        }
        epilogue_chunk(p, v, gm, gn, slice);
      }
    }
    if (ps + 1 < 4) __syncthreads();
  }
}

// =====================================================================================================================================
// gemm_a4p_kernel: the PERSISTENT form with a deferred write-out (generator: class GenP).  One block per CU walks its tiles (ids
// blockIdx.x + k * gridDim.x through the XCD-aware remap); the bf16 output of a tile is rounded out of the AGPRs at the first step of the
// next tile and leaves through a wave-private LDS transposition as full 256-byte row segments while that tile's MFMAs run; the DMA
// stream never stops at a tile edge.  Plain bf16 epilogue, M >= 256, N >= 512 (a ragged last tile row / column is shifted up to end at the
// edge and overlaps its neighbour), K a multiple of 128 and >= 384.  The whole kernel
// body is the asm statement.
constexpr int A4P_LDS = 4 * 32768 + 4 * 8192;      // ring + one 8 KiB staging block per wave = all 160 KiB

template <bool TB, int EPI>      // EPI 0: plain; 1: ReLU-mask epilogue (dact = RELU, z, optional 1 / (1 - p) scale; TB only); 2: ReLU; 3: ReLU + dropout
                                 // (the library's counter-based mask v2s_keep8 regenerated in the MFMA gaps, scale 1 / (1 - p); ldc == N, K >= 640) -- !TB only
__global__ __launch_bounds__(256, 1) void gemm_a4p_kernel(const GemmP p) {
  static_assert(EPI == 0 || (EPI == 1) == TB, "the mask epilogue is generated for the dgrad layout only, the ReLU ones for the forward layout");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const uint32_t lds = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const uint32_t pa0 = (uint32_t)(uintptr_t)p.A, pa1 = (uint32_t)((uintptr_t)p.A >> 32);
  const uint32_t pb0 = (uint32_t)(uintptr_t)p.B, pb1 = (uint32_t)((uintptr_t)p.B >> 32);
  const uint32_t pc0 = (uint32_t)(uintptr_t)p.C, pc1 = (uint32_t)((uintptr_t)p.C >> 32);
  const uint32_t pz0 = (uint32_t)(uintptr_t)p.z, pz1 = (uint32_t)((uintptr_t)p.z >> 32);
  const uint32_t lda = (uint32_t)(p.lda * 2), ldb = (uint32_t)(p.ldb * 2), ldc = (uint32_t)(p.ldc * 2);
  const uint32_t cbytes = p.dbg == 1 ? 0u : (uint32_t)((long)p.M * p.ldc * 2);      // gemm_dbg = 1: every store out of range (ablation)
  const uint32_t zbytes = (uint32_t)((long)p.M * p.ldc * 2);                         // (ldz == ldc: dispatcher)
  const uint32_t scale = __float_as_uint(p.inv_keep);
  const uint32_t niter = (uint32_t)(p.K / 128);
  const uint32_t ntiles = (uint32_t)(p.tilesM * p.tilesN), tilesn = (uint32_t)p.tilesN;
  const uint32_t bid = blockIdx.x, grid = gridDim.x;
  const uint32_t q = ntiles >> 3, r = ntiles & 7;
  // grouped tile walk (generator: tile_setup): GM tile rows per group, column by column inside a group; GM = 1 is the row-major walk
  const uint32_t tilesm = (uint32_t)p.tilesM;
  uint32_t GM = (uint32_t)(p.order > 0 ? p.order : 1);
  if (GM > tilesm) GM = tilesm;
  if (GM > 255) GM = 255;
  const uint32_t gm_last = tilesm % GM, glast = gm_last ? tilesm / GM : 0xffffu;
  const auto magic_of = [](uint32_t d) { return d >= 2 ? (uint32_t)(((1ull << 32) + d - 1) / d) : 0u; };
  const uint32_t gsz = GM * tilesn, magicg = magic_of(gsz), magicm = magic_of(GM), magicl = magic_of(gm_last);
  const uint32_t walk = GM | (gm_last << 8) | (glast << 16);
  const uint32_t nmy = (ntiles - bid + grid - 1) / grid;
  const uint32_t mlast = (uint32_t)(p.M - 256), nlast = (uint32_t)(p.N - 256);
  // dropout: element (gm, gn) belongs to chunk ((gm + row0) * N + gn) / 8 of the launch's mask stream (v2s_keep8)
  const uint32_t hseed = v2s_salted(p.seed, p.salt) * 0x9E3779B1u + (uint32_t)p.row0 * (uint32_t)(p.N / 8);
  const uint32_t hpp = (((p.p16 ^ 0x8000u) - 1u) & 0xffffu) * 0x10001u;
#define A4P_ARGS                                                                                                                          \
  : [tid] "v"(tid), [pa0] "s"(pa0), [pa1] "s"(pa1), [pb0] "s"(pb0), [pb1] "s"(pb1), [lda] "s"(lda), [ldb] "s"(ldb), [pc0] "s"(pc0),       \
    [pc1] "s"(pc1), [ldc] "s"(ldc), [cbytes] "s"(cbytes), [niter] "s"(niter), [lds] "s"(lds), [bid] "s"(bid), [grid] "s"(grid), [q] "s"(q), \
    [r] "s"(r), [magicg] "s"(magicg), [gsz] "s"(gsz), [magicm] "s"(magicm), [magicl] "s"(magicl), [walk] "s"(walk), [nmy] "s"(nmy), [mlast] "s"(mlast), [nlast] "s"(nlast), [pz0] "s"(pz0),          \
    [pz1] "s"(pz1), [zbytes] "s"(zbytes), [scale] "s"(scale), [hseed] "s"(hseed), [hpp] "s"(hpp)                                           \
  : A4P_CLOBBERS
  if constexpr (EPI == 1) {
    asm volatile(A4P_MAIN_NN_DACT : A4P_ARGS);
  } else if constexpr (EPI == 2) {
    asm volatile(A4P_MAIN_NT_RELU : A4P_ARGS);
  } else if constexpr (EPI == 3) {
    asm volatile(A4P_MAIN_NT_RELUDROP : A4P_ARGS);
  } else if constexpr (TB) {
    asm volatile(A4P_MAIN_NN : A4P_ARGS);
  } else {
    asm volatile(A4P_MAIN_NT : A4P_ARGS);
  }
#undef A4P_ARGS
}
