/*
 * libvid2seq_hip.so -- C ABI of the MI355X-native (gfx950) Vid2Seq hot path.
 *
 * The reference (antoyang/VidChapters) has no FFI for this path: its boundary is the Python module
 * surface Vid2Seq.forward()/generate() (model/vid2seq.py:58-167).  This header is the C-ABI that a
 * maintainer would bind underneath that surface (INTEGRATION.md shows the ctypes stubs).  Each entry
 * point names the reference arithmetic it replaces (file:line relative to the reference root).
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory unless noted "host".
 *   - the caller owns every buffer (inputs, outputs, saved-for-backward, workspace); the library
 *     allocates nothing and never synchronises the device.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - activations are bf16 (raw uint16 bits), accumulators/statistics/gradients-of-parameters fp32.
 *   - return value: 0 = ok, negative = V2S_ERR_*; message via v2s_last_error() (thread local).
 */
#ifndef VID2SEQ_HIP_H
#define VID2SEQ_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define V2S_OK 0
#define V2S_ERR_SHAPE (-1)
#define V2S_ERR_DTYPE (-2)
#define V2S_ERR_ALIGN (-3)
#define V2S_ERR_LAUNCH (-4)
#define V2S_ERR_ARG (-5)

#define V2S_BF16 0
#define V2S_F32 1

/* 3 (round 4): v2s_attn_bwd's `delta` became a [B][H][Nq][4] workspace it WRITES, v2s_topp_sample_step gained top_k, dact=RELU with
 * dropout expects z = the post-dropout activation (round 3 changes that a version-2 caller would corrupt memory with);
 * + v2s_rowsumsq_range / v2s_timetoken_renorm_sq, options gemm_ps / gemm_ps_nst / gemm_w128
 * 4 (round 4): + v2s_decode_qfold / v2s_decode_memattn_plan / v2s_decode_memattn / v2s_decode_ctxfold, v2s_beam_advance (additions only)
 * 5 (round 5): + v2s_sum_n, v2s_argmax_step_tail, options gemm_a4 / gemm_a4_grid / gemm_a4_relu / gemm_a4_walk (additions only)
 * 6 (round 6): + v2s_clock_probe, v2s_lmhead_ce_fwd / _bwd / _workspace_floats; the round-4 experiment options gemm_ps / gemm_ps_nst / gemm_w128 are gone
 *   (v2s_set_option returns V2S_ERR_ARG for them) */
#define V2S_ABI_VERSION 6

int v2s_version(void);
const char* v2s_last_error(void);
/* runtime switches (tuning / profiling aids; defaults are what the product path uses):
 *   "tr_read"       1: ds_read_b64_tr_b16 operand transposes (default), 0: scalar LDS gathers
 *   "gemm_dma"      2: LDS-DMA 128x128 main loop for every variant with K % 64 == 0 (default), 1: transposed-operand variants only,
 *                   0: register-staged loop everywhere
 *   "gemm_big"      1: tile-size heuristics (default), 0: 128x128 only, 2: 256x128 8-wave only, 3: force the 4-wave 256x128x32 kernel
 *   "gemm_skinny"   1: dedicated weight-streaming kernels for cached decoding (M <= 64; M <= 512 when N < 8192; default),
 *                   0: general tiles (A/B of the block shape, tools/decode_ab.py: 2 = four row fragments per block,
 *                   4 = four waves, <waves><mt><nt> / 1<waves><mt><nt> = explicit shape for M > 64 / M <= 64, 20000 + n = one row fragment per
 *                   block only up to n blocks; none of them faster on the decode loop: profiles/r06_decode_chain_study.txt)
 *   "gemm_order"    GM > 0: grouped tile walk, GM tile rows deep, K slices tile-major (default 4: the blocks an XCD runs together share
 *                   operand slabs in its L2; +20..40 % on the split-K weight gradients), 0: row-major with adjacent K slices
 *   "gemm_split"    1: split-K slice count from the rounds x length cost model (default), 0: fixed block-count target
 *   "gemm_p8"       8-phase ping-pong kernel (256-row tiles, counted vmcnt LDS-DMA ring): 1: where it measured faster (default),
 *                   0: never, 2: 256x256 tiles wherever legal, 3: 256x128 tiles wherever legal
 *   "gemm_a4"       4-wave 256x256 kernels with a generated, hand-scheduled asm K loop (128x128 wave tiles in AGPRs, v_mfma 32x32x16, counted
 *                   waits; round 5): 1: where they measured faster (default: the persistent deferred-write-out form on plain bf16 GEMMs with
 *                   >= 256 tiles), 0: never, 2: wherever legal, 3: the one-tile form wherever legal, 4: like 1 plus long-contraction weight gradients, 5: like 1 plus the ReLU-mask dgrad
 *                   epilogue (4 and 5: faster alone, slower inside the train step)
 *   "gemm_a4_grid"  blocks of the persistent form: 0 (default) one per CU, n > 0 at most n, -1 the fewest that need the same number of rounds (A/B knobs: the
 *                   step is work-bound, none of them moves it)
 *   "gemm_a4_walk"  tile walk of the persistent form: 0 (default) row-major below 16 tile columns, groups of 4 tile rows from there; n > 0: groups of n tile rows
 *   "gemm_a4_relu"  1 (default): the persistent form also takes forward GEMMs with a ReLU or ReLU + dropout epilogue (the FFN's wi: the
 *                   dropout mask of the library's counter-based generator is recomputed inside the kernel's MFMA gaps), 0: plain epilogues only
 *   "gemm_dbg"      profiling ablations of the tiled kernels (results invalid when non-zero): 1 = no global store, 2 = no epilogue, 3 = no hand-off
 *   "fp32_io"       DEBUG: 1 = v2s_*norm_fwd/bwd, v2s_ce_bwd and v2s_attn_fwd/bwd take and return FP32 activations (attention: fp32-arithmetic
 *                   reference kernels, dense layout only); parity work against fp32 references (<= 1e-4), never set by the product path
 *   "attn_bwd_part" 0: v2s_attn_bwd launches dQ and dK/dV kernels (default), 1: dQ only, 2: dK/dV only (per-kernel timing)
 *   "attn_order"    block -> (sequence, head, query / key block) of the three attention kernels: 0 (default, round 6): the (sequence, head) groups are dealt to
 *                   the 8 XCDs round-robin (all XCDs work through the same sequences at the same time: the in-order workgroup dispatch no longer waits
 *                   for the XCD with the longest sequences), the dK/dV kernel in two passes; 1: every XCD a contiguous range of groups (rounds 1-5);
 *                   2: round-robin without the two passes; 400 + n / 500 + n: sweep knobs.  Results do not depend on it */
int v2s_set_option(const char* name, int value);
int v2s_get_option(const char* name);

/* sizeof() of an argument struct ("v2s_gemm_args", "v2s_attn_args", "v2s_adam_args", "v2s_decode_attn_args") as this library was
 * compiled, -1 for an unknown name.  A binding compares it with the size of its own struct definition before the first call: a
 * definition that is one field short would make the library read past the caller's buffer. */
int64_t v2s_sizeof(const char* struct_name);

/* Device-resident dropout salt: while set, every launch that draws a dropout mask (GEMM epilogues, attention, embedding, v2s_dropout)
 * uses seed ^ *dev_word instead of its by-value seed.  The pointer is read when a launch is ENQUEUED and travels as a kernel argument,
 * so a training step captured into a hipGraph draws new masks on every replay once the caller changes the word between replays
 * (vidchapters_amd.train.Trainer.step_graph).  NULL restores the by-value seeds.  The setting is PER HOST THREAD (it applies to the
 * launches the calling thread enqueues): no process-global mutable state besides the read-mostly tuning options above. */
int v2s_set_seed_salt(const uint32_t* dev_word);

/* ------------------------------------------------------------------------------------------------
 * GEMM:  C[M,N] (+)= epilogue( alpha * sum_k A(m,k) * B(n,k) )
 * replaces every nn.Linear on the path: vit.py:41,53,17,20; modeling_t5.py:304-311,528-536,581,1714
 * (forward), and their autograd dgrad/wgrad.
 *   transA = 0: A stored [M][K] (row stride lda, K contiguous);  1: stored [K][M] (row stride lda)
 *   transB = 0: B stored [N][K] (row stride ldb, K contiguous);  1: stored [K][N] (row stride ldb)
 *     forward  y = x W^T        : transA=0, transB=0 (W is [out,in] like nn.Linear.weight)
 *     dgrad    dx = dy W        : transA=0, transB=1
 *     wgrad    dW = dy^T x      : transA=1, transB=1
 *   epilogue order: v = alpha*acc; v += bias[n]; (pre-activation optionally stored to `pre`);
 *     v = act(v); v *= dact(z[m,n]) ; dropout(v) ; v += residual[m,n]; C = (accumulate ? C + v : v)
 *   M, N arbitrary; K, N, lda, ldb, ldc multiples of 8 elements; pointers 16-byte aligned.
 * ---------------------------------------------------------------------------------------------- */
#define V2S_ACT_NONE 0
#define V2S_ACT_RELU 1
#define V2S_ACT_GELU 2 /* exact erf GELU (torch nn.GELU default) */

typedef struct v2s_gemm_args {
  int32_t M, N, K;
  int32_t transA, transB;
  const void* A; /* bf16 */
  const void* B; /* bf16 */
  int64_t lda, ldb;
  void* C;
  int64_t ldc;
  int32_t c_dtype;    /* V2S_BF16 or V2S_F32 */
  int32_t accumulate; /* C += result (f32 output only) */
  float alpha;
  const float* bias;    /* [N] fp32 or NULL */
  int32_t act;          /* V2S_ACT_* applied in the forward direction */
  void* pre;            /* bf16 [M][ldc] or NULL: pre-activation copy (needed by GELU backward) */
  int32_t dact;         /* V2S_ACT_*: multiply by act'(z) (RELU: z>0 ? 1:0 with z = forward output;
                           GELU: gelu'(z) with z = saved pre-activation).  RELU together with dropout_p > 0: z must be the forward
                           GEMM's own output dropout(relu(.)) (same p): z > 0 then already says "active and kept", so only the
                           1/(1-p) scale is applied and dropout_seed is not consulted */
  const void* z;        /* bf16 [M][ldz] */
  int64_t ldz;
  const void* residual; /* bf16 [M][ldr] or NULL */
  int64_t ldr;
  float dropout_p;      /* 0 = off */
  uint32_t dropout_seed;
  void* workspace;      /* optional fp32 scratch: enables split-K for few-tile/long-K (weight-gradient) shapes */
  int64_t workspace_bytes;
  float rms_eps;        /* > 0: fused T5 RMSNorm prologue for cached decoding (M <= 64, or M <= 512 with N < 8192): row m of the result is multiplied by
                           rsqrt(mean_k(A[m][k]^2) + rms_eps) before alpha/bias/...; the norm's weight vector must have been folded
                           into B's columns (v2s_scale_cols).  Replaces T5LayerNorm + Linear, modeling_t5.py:263-277 + :528-536 */
  int32_t decode;       /* 1: the call is part of a cached decode step (a chain of small dependent launches): the weight-streaming
                           kernels then also take 64 < M <= 512 rows (beam search).  They split K across the waves of a block, so
                           their fp32 summation order differs from the tiled kernels; training keeps 0 and with it results that do
                           not depend on how many rows a call has (M <= 64 always takes them) */
} v2s_gemm_args;

int v2s_gemm(const v2s_gemm_args* args, void* stream);
/* GROUPED weight gradients: `count` (<= 16) problems of ONE shape in one launch -- C[i][M][N] (+)= alpha * A[i]^T B[i] with A[i] = [K][lda]
 * (dY), B[i] = [K][ldb] (X), fp32 C -- e.g. the same projection's weight gradient of every decoder layer, whose operands live in
 * separate allocations.  `args` carries the shape, leading dimensions, alpha and accumulate (transA = transB = 1, c_dtype = V2S_F32, no
 * epilogue, its A / B / C pointers are ignored); K a multiple of 64.  Whole-K tiles, no split-K, no reduction: deterministic.
 * (modeling_t5.py:304-311,528-536 autograd weight gradients of the decoder / ViT layers.) */
int v2s_gemm_grouped(const v2s_gemm_args* args, int32_t count, const void* const* A, const void* const* B, void* const* C, void* stream);
/* symbol of the kernel variant the calling thread's last v2s_gemm dispatched (for profiling: matches rocprofv3 kernel names) */
const char* v2s_last_gemm_kernel(void);

/* out[r][c] = W[r][c] * w[c]  (bf16 [rows][cols] x fp32 [cols] -> bf16): folds a norm weight into the next projection */
int v2s_scale_cols(const void* W, const float* w, void* out, int32_t rows, int32_t cols, void* stream);

/* column sums of a bf16 matrix (bias gradients of the ViT linears: autograd of vit.py:41,53,17,20)
 * out[n] (+)= sum_m X[m][n];  out fp32 */
int v2s_colsum(const void* X, int64_t ldx, int32_t M, int32_t N, float* out, int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Normalisation
 *   rmsnorm: modeling_t5.py:263-277 (T5LayerNorm): y = w * x * rsqrt(mean(x^2)+eps)
 *   layernorm: torch nn.LayerNorm used at vit.py:64,69,99 (eps 1e-5, affine)
 *   x,y bf16 [rows][cols]; w,b fp32 [cols]; rstd/mean fp32 [rows] (saved for backward)
 *   backward: dx bf16; dw/db fp32 [cols], accumulated (+=, hardware float atomics) into the caller's gradient buffers.
 *   `dx_add` (bf16, may be NULL) is added to dx: the residual-stream gradient that bypasses the norm.
 * ---------------------------------------------------------------------------------------------- */
int v2s_rmsnorm_fwd(const void* x, const float* w, void* y, float* rstd, int32_t rows, int32_t cols,
                    float eps, void* stream);
int v2s_rmsnorm_bwd(const void* x, const float* w, const float* rstd, const void* dy, void* dx,
                    const void* dx_add, float* dw, int32_t rows, int32_t cols, void* stream);
int v2s_layernorm_fwd(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                      int32_t rows, int32_t cols, float eps, void* stream);
int v2s_layernorm_bwd(const void* x, const float* w, const float* mean, const float* rstd, const void* dy,
                      void* dx, const void* dx_add, float* dw, float* db, int32_t rows, int32_t cols, void* stream);
/* The same backward with a second output dx_drop = dropout(dx; dropout_p, dropout_seed) (bf16 [rows][cols], mask and values identical to
 * v2s_dropout on the stored dx): the gradient operand of the sublayer the backward pass visits next, whose forward dropped its
 * output before the residual add (modeling_t5.py:618,654,353; vit.py:54,21) -- saves re-reading the residual-stream gradient */
int v2s_rmsnorm_bwd_drop(const void* x, const float* w, const float* rstd, const void* dy, void* dx, const void* dx_add,
                         float* dw, int32_t rows, int32_t cols, void* dx_drop, float dropout_p, uint32_t dropout_seed, void* stream);
int v2s_layernorm_bwd_drop(const void* x, const float* w, const float* mean, const float* rstd, const void* dy, void* dx,
                           const void* dx_add, float* dw, float* db, int32_t rows, int32_t cols, void* dx_drop, float dropout_p,
                           uint32_t dropout_seed, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused attention (flash-style, scores never materialised)
 *   replaces vit.py:47-51 and modeling_t5.py:539-580 (+ compute_bias :445-460, masks :996,1005,559)
 *   S[b,h,q,k] = scale * <Q[b,q,h,:], K[b,k,h,:]> + bias[h][k-q+(Nq-1)]   (bias optional)
 *   masked (key_mask[b,k]==0, or causal && k>q+causal_off) scores are REPLACED by finfo(float32).min,
 *   which reproduces the reference's additive (1-m)*finfo.min exactly in fp32 (fully masked rows
 *   become uniform over all keys, like the reference).
 *   P = softmax_k(S); dropout(P); O[b,q,h,:] = P V.   head_dim must be 64.
 *   Q/K/V/O are bf16 with element strides (batch stride, row stride); head h sits at column h*64.
 *   `ml` fp32 [B][H][Nq][2] = (row max, row sum) saved for backward.
 *   backward needs `o` (the forward output) and a caller-owned fp32 workspace `delta` [B][H][Nq][4] that v2s_attn_bwd fills
 *   itself (per row: -(m + log2 l), exponent of a masked element, -sum_d dO*O, dropout row seed -- bit 31 of that word: the
 *   row's dO is all zero (+-0), such rows add nothing to any gradient and are skipped exactly: written by its dQ kernel, read
 *   by its dK/dV kernel) and produces dQ,dK,dV
 *   (same strides as q/k/v via dq_*, dk_*, dv_*) and, if bias != NULL, dbias_diag fp32
 *   [H][Nq+Nk-1] accumulated (+=) per relative position (bucket-reduced by v2s_bias_bucket_bwd).
 * ---------------------------------------------------------------------------------------------- */
typedef struct v2s_attn_args {
  int32_t B, H, Nq, Nk;
  const void *q, *k, *v;
  int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs; /* batch / row strides in elements */
  void* o;
  int64_t o_bs, o_rs;
  float* ml;              /* [B][H][Nq][2] */
  float scale;            /* 1.0 for T5 (modeling_t5.py:539-541), head_dim^-0.5 for ViT (vit.py:30) */
  const float* bias_diag; /* fp32 [H][Nq+Nk-1] or NULL; index (k - q) + (Nq-1) */
  const uint8_t* key_mask; /* [B][Nk] 1=keep, or NULL */
  int32_t causal;         /* 1: key k visible iff k <= q + causal_off */
  int32_t causal_off;     /* Nk - Nq for cached decoding, else 0 */
  float dropout_p;
  uint32_t dropout_seed;
  /* backward only */
  const void* d_o;
  int64_t do_bs, do_rs;
  float* delta;           /* workspace [B][H][Nq][4], 16-byte aligned (see above) */
  void *dq, *dk, *dv;
  int64_t dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs;
  float* dbias_diag;      /* fp32 [H][Nq+Nk-1], += ; or NULL */
  /* optional (backward): every relative position d = k-q <= bias_far_lo shares ONE bias bucket, likewise every
   * d >= bias_far_hi (T5: +-max_distance).  The gradient mass of those regions is then summed without per-diagonal
   * resolution and deposited on the diagonals bias_far_lo / bias_far_hi themselves, i.e. dbias_diag is exact after
   * bucket reduction (v2s_bias_bucket_bwd) but not per diagonal.  Disabled when bias_far_lo >= bias_far_hi (0,0). */
  int32_t bias_far_lo, bias_far_hi;
  /* optional packed ("varlen") SELF-attention: B+1 int32 row offsets on the device.  Sequence b then occupies rows
   * [seq_off[b], seq_off[b+1]) of q/k/v/o (and d_o/dq/dk/dv): the *_bs batch strides are ignored, Nq = Nk = the nominal
   * (maximum) length that sizes the grid, the bias diagonal table, ml and delta ([B][H][Nq] indexing as before) -- no sequence may be longer;
   * key_mask must be NULL.
   * Rows of pad tokens simply do not exist -- the reference computes them and masks them as keys (modeling_t5.py:996), so the
   * rows that remain are identical */
  const int32_t* seq_off;
  /* 1: seq_off packs the QUERY side only (q, o, d_o, dq; ml / delta keep their [B][H][Nq] indexing): K, V, dK, dV stay dense
   * [B][Nk] with their batch strides and key_mask is allowed -- cross-attention of a padding-free decoder over the padded memory */
  int32_t seq_q_only;
  /* optional row offsets of the KEY side (B+1 int32 on the device): K, V, dK, dV of sequence b occupy rows [kv_seq_off[b],
   * kv_seq_off[b+1]) (batch strides ignored, Nk = the nominal maximum, key_mask must be NULL) -- cross-attention over a padding-free
   * [video ; text] memory; the query side is dense or packed by seq_off independently */
  const int32_t* kv_seq_off;
} v2s_attn_args;

int v2s_attn_fwd(const v2s_attn_args* a, void* stream);
int v2s_attn_delta(const v2s_attn_args* a, float* delta, void* stream); /* stand-alone delta[B][H][Nq] = rowsum(dO * O); v2s_attn_bwd no longer needs it */
int v2s_attn_bwd(const v2s_attn_args* a, void* stream);

/* relative-position bias (modeling_t5.py:397-460): host passes the bucket LUT lut[i] = bucket(i-(Nq-1)),
 * i in [0, Nq+Nk-1) (int32, computed on the host exactly as the reference does in fp32).
 *   fwd: bias_diag[h][i] = table[lut[i]][h]          (table fp32 [num_buckets][H])
 *   bwd: dtable[lut[i]][h] += dbias_diag[h][i]       */
int v2s_bias_diag_fwd(const float* table, const int32_t* lut, float* bias_diag, int32_t H, int32_t n,
                      int32_t num_buckets, void* stream);
int v2s_bias_bucket_bwd(const float* dbias_diag, const int32_t* lut, float* dtable, int32_t H, int32_t n,
                        int32_t num_buckets, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Embedding (modeling_t5.py:972, vid2seq.py:71): out[i,:] = table[ids[i],:]  (bf16 table -> bf16)
 * optional dropout; backward scatter-adds bf16 dy rows into the fp32 gradient of the tied table.
 * ---------------------------------------------------------------------------------------------- */
int v2s_embed_fwd(const int64_t* ids, const void* table, void* out, int64_t n, int32_t d, int32_t vocab,
                  float dropout_p, uint32_t dropout_seed, void* stream);
int v2s_embed_bwd(const int64_t* ids, const void* dy, float* dtable, int64_t n, int32_t d, int32_t vocab,
                  float dropout_p, uint32_t dropout_seed, void* stream);

/* elementwise helpers on bf16 [n] : y = x + add[(i mod add_n)] ; dropout ; grad-of-dropout */
int v2s_add_bcast(const void* x, const void* add, void* y, int64_t n, int64_t add_n, void* stream);
int v2s_dropout(const void* x, void* y, int64_t n, float p, uint32_t seed, void* stream);
int v2s_add(const void* a, const void* b, void* y, int64_t n, void* stream);
/* y[n] = sum_{p < nparts} parts[p * stride + .] (bf16 terms summed in fp32, one rounding; n, stride multiples of 8).  Round 5: the twelve
 * decoder layers' cross-attention memory gradients are written by plain GEMMs and summed once (modeling_t5.py:528-536 backward) instead of
 * a chain of residual epilogues. */
int v2s_sum_n(const void* parts, int64_t stride, int32_t nparts, void* y, int64_t n, void* stream);
/* Measurement aid, no counterpart in the reference: one wave per XCD writes out[xcd][4] = (shader-clock cycle counter s_memtime, constant
 * 100 MHz counter s_memrealtime, xcd id, 1) as uint64 (out: 8 x 4 words, zeroed by the caller).  Two probes around a region on the same
 * stream give the average shader clock it ran at -- (cycles1 - cycles0) / (ticks1 - ticks0) x 100 MHz -- which is what separates a slow
 * box or a power-throttled step from a slow kernel (bench.py: roofline.effective_sclk_mhz). */
int v2s_clock_probe(uint64_t* out, void* stream);
/* out_f32[i mod add_n] += sum over broadcast copies of dy (pos_embed gradient) */
int v2s_bcast_grad(const void* dy, float* out, int64_t n, int64_t add_n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Label-smoothed cross entropy over fp32 logits (modeling_t5.py:1721 == F.cross_entropy(ignore_index
 * =-100, label_smoothing=eps)), forward + gradient in one pass:
 *   per row i with label y != -100: l_i = (1-eps)*(lse - z_y) + eps*(lse - mean_c z_c)
 *   row_lse: fp32 [rows][2] = (logsumexp, l_i) saved for backward;
 *   loss_sum += sum l_i ; count += #rows kept   (fp32 device scalars, deterministic single-block
 *   reduction; caller zeroes them);  dlogits (bf16, same shape) = (softmax - (1-eps)*onehot - eps/V) * gscale[0]
 *   for kept rows and 0 for ignored rows, where gscale is a device scalar (= upstream grad / count).
 *   Two entry points so that count is known before gradients are scaled.
 * ---------------------------------------------------------------------------------------------- */
int v2s_ce_fwd(const float* logits, int64_t ld, const int64_t* labels, int32_t rows, int32_t V, float eps,
               float* row_lse, float* loss_sum, float* count, void* stream);
int v2s_ce_bwd(const float* logits, int64_t ld, const int64_t* labels, const float* row_lse, int32_t rows,
               int32_t V, float eps, const float* gscale, void* dlogits, int64_t ldd, void* stream);

/* Tied LM head + label-smoothed cross entropy WITHOUT logits in memory (round 6; modeling_t5.py:1709-1721: lm_logits = (h * d_model^-0.5) E^T, then
 * F.cross_entropy(ignore_index = -100, label_smoothing = eps)).  h: bf16 [rows][ldh] (decoder output), E: bf16 [Vpad][d] (the tied embedding; rows
 * V..Vpad-1 exist and are zero), labels: int64 [rows] (-100 = ignored), alpha = d_model^-0.5 for a tied head.
 *   v2s_lmhead_ce_fwd: logits are computed 128 x 128 tiles at a time and reduced on the spot to per-(row, 64-column half tile) statistics in `part`
 *     (v2s_lmhead_ce_workspace_floats(rows, Vpad) floats, 16-byte aligned); row_out[rows][2] = (log-sum-exp, smoothed loss) per row -- the same
 *     two numbers v2s_ce_fwd leaves --, *loss_sum += sum of the row losses, *count += number of non-ignored rows (deterministic reduction).
 *   v2s_lmhead_ce_bwd: recomputes the tiles and writes d(logits)[rows][ldd] = (softmax - (1 - eps) onehot - eps / V) * *gscale as bf16 (columns
 *     V..ldd-1 zero: the d(hidden) GEMM contracts over the padded vocabulary); ignored rows are zero.  The two consumers are ordinary v2s_gemm
 *     calls (d(hidden) = d(logits) E alpha, d(E) += d(logits)^T h alpha). */
int64_t v2s_lmhead_ce_workspace_floats(int32_t rows, int32_t Vpad);
int v2s_lmhead_ce_fwd(const void* h, int64_t ldh, const void* E, int32_t rows, int32_t V, int32_t Vpad, int32_t d, float alpha,
                      const int64_t* labels, float eps, float* part, float* row_out, float* loss_sum, float* count, void* stream);
int v2s_lmhead_ce_bwd(const void* h, int64_t ldh, const void* E, int32_t rows, int32_t V, int32_t Vpad, int32_t d, float alpha,
                      const int64_t* labels, const float* row_out, float eps, const float* gscale, void* dlogits, int64_t ldd, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimiser over the flat parameter arena (dvc.py:112-126: clip_grad_norm_, torch.optim.Adam step,
 * time-token renorm) + bf16 shadow refresh.
 * ---------------------------------------------------------------------------------------------- */
/* out_sum[0] += sum(g^2), deterministic two-stage reduction; partial_ws: >= 1024 floats */
int v2s_sqnorm(const float* g, int64_t n, float* partial_ws, float* out_sum, void* stream);
typedef struct v2s_adam_args {
  float* p; float* m; float* v; const float* g; void* p_bf16; /* bf16 shadow or NULL */
  int64_t n;
  float lr, beta1, beta2, eps, weight_decay;
  int32_t step;            /* 1-based */
  const float* gnorm_sq;   /* device scalar: sum of squared grads (for clipping) or NULL */
  float max_norm;          /* <=0: no clipping */
  float grad_scale;        /* extra multiplier applied to g before everything (e.g. 1/world) */
  const float* hyper_dev;  /* NULL, or device {lr / (1 - beta1^step), 1 / sqrt(1 - beta2^step)} read by the kernel INSTEAD of lr / step: a captured
                            * hipGraph replays the launch with the values of the current step */
} v2s_adam_args;
int v2s_adam_step(const v2s_adam_args* a, void* stream);
int v2s_cast_bf16(const float* src, void* dst, int64_t n, void* stream);
/* dvc.py:118-126: rows [V-num_bins, V) of emb (fp32 [V][d]) are divided by
 * mean(row-norm of those rows)/mean(row-norm of rows [0,V-num_bins)); shadow refreshed. ws: V+2 floats */
int v2s_timetoken_renorm(float* emb, void* emb_bf16, int32_t V, int32_t d, int32_t num_bins, float* ws,
                         void* stream);
/* The same renorm for a SHARDED data-parallel optimizer (train.GradSync shard=True), where a rank's fp32 values of the frozen rows
 * are current only inside the stripes it owns: v2s_rowsumsq_range adds, per row, the squares of the elements whose flat index
 * row*d + c lies in [f0, f1) to sumsq[row] (the caller all-reduces the per-rank partial sums of the frozen rows);
 * v2s_timetoken_renorm_sq then takes the frozen rows' norms from text_sumsq[0 .. V-num_bins) and reads only the time-token rows
 * (whole on every rank) from emb.  Same arithmetic as v2s_timetoken_renorm otherwise (dvc.py:118-126). */
int v2s_rowsumsq_range(const float* emb, int32_t V, int32_t d, int64_t f0, int64_t f1, float* sumsq, void* stream);
int v2s_timetoken_renorm_sq(float* emb, void* emb_bf16, int32_t V, int32_t d, int32_t num_bins, const float* text_sumsq, float* ws,
                            void* stream);

/* ------------------------------------------------------------------------------------------------
 * Greedy decoding helpers (HF 4.28 greedy_search, call site vid2seq.py:150-162)
 *   decode attention: one query row per (b,h) against a KV cache [B][Nk_max][H*64]-strided
 *   argmax over fp32 logits rows with the finished-row rule (finished rows emit pad)
 * ---------------------------------------------------------------------------------------------- */
typedef struct v2s_decode_attn_args {
  int32_t B, H, Nk;
  const void* q; int64_t q_bs;            /* bf16 [B][H*64] */
  const void* k; const void* v; int64_t kv_bs, kv_rs;
  void* o; int64_t o_bs;
  const float* bias_row;                  /* fp32, element (h,k) at bias_row[h*bias_ld + k], or NULL */
  int64_t bias_ld;
  const uint8_t* key_mask;                /* [B][mask_ld] or NULL */
  int64_t mask_ld;
  float scale;
  /* optional device-resident step counter (hipGraph-replayable decode loops): when pos_dev != NULL the number of keys is
   * *pos_dev + 1 (Nk is only an upper bound) and bias_row is advanced by (bias_maxlen - 1 - *pos_dev) elements, i.e. the
   * row of a [H][2*bias_maxlen-1] relative-position diagonal table that belongs to query position *pos_dev */
  const int32_t* pos_dev;
  int32_t bias_maxlen;
  int32_t kv_group;                        /* >1: KV batch index = b / kv_group (beams sharing the cross K/V); 0/1 = b */
  /* optional fused cache append (self-attention, needs pos_dev): the step's fresh K / V rows, bf16 [B][H*64] with batch stride new_bs.
   * Key *pos_dev is read from here instead of the cache, and the (b, h) block writes its pieces into cache row *pos_dev
   * (replaces a separate v2s_kv_append launch per layer and step; modeling_t5.py:555-556 grows the cache with torch.cat) */
  const void* new_k; const void* new_v; int64_t new_bs;
  /* optional beam bookkeeping WITHOUT moving the cache (needs new_k): row_map[b*row_map_ld + k] = the cache row that physically holds
   * key k of query row b.  The (b, h) blocks append the step's K/V into cache row b and set row_map[b][*pos_dev] = b; a beam reorder
   * is then row_map[b][:] = row_map[src[b]][:] on a [rows][maxlen] int32 table instead of a copy of every cached K/V row
   * (v2s_kv_gather: 12 layers x rows x len x 3 KB per step; HF reorders the cache itself, modeling_t5.py:1771-1793).  B <= 65535. */
  int32_t* row_map; int64_t row_map_ld;
} v2s_decode_attn_args;
int v2s_decode_attn(const v2s_decode_attn_args* a, void* stream);
/* Decode-step cross-attention against the encoder memory itself instead of per-layer K / V caches (d = 768, head width 64).
 * The reference projects the memory to K and V in every decoder layer and keeps both (modeling_t5.py:484-525, past_key_value[2:4]);
 * both come from the same rows, so with the folded query qp_h = Wk_h^T q_h and accn_h = sum_k p_k mem_k:
 *   score_h[k] = qp_h . mem_k,  ctx_h = Wv_h accn_h   -- one pass over [S, d] per layer (half the bytes, one tensor for all layers).
 * v2s_decode_qfold:   qp[rows][H][d] = per head ((rstd * x) Wq_h^T) Wk_h; x bf16 [rows][ldx]; wq bf16 [H*64][d] (RMSNorm weight
 *                     folded into its columns when rms_eps > 0, else rstd = 1); wkT bf16 [d][H*64] = the K projection transposed.
 * v2s_decode_memattn_plan (host only, no GPU work): cuts every entry's ceil(klen / 32) key tiles into ceil(tiles / tiles_per_piece) pieces
 *                     of equal length (one block per piece).  The cut of an entry depends on its own length only, so a sequence decodes
 *                     bit-identically in any batch; 9 tiles per piece = 4 pieces for 1100 keys: 64 such entries fill 256 CUs once.  klen_host[e] >= 1 =
 *                     the valid keys of entry e (a prefix of its memory rows).  Writes blk[4 * nblk] = (entry, first tile | end tile
 *                     << 16, slot, klen) and slot_off[entries + 1] (the slots of entry e are slot_off[e] .. slot_off[e + 1], at most
 *                     64); both go to the device.
 * v2s_decode_memattn: one block per blk entry: the R = beams * H consecutive rows of qp of its entry (R <= 48) against its key tiles of
 *                     mem + entry * mem_es (bf16 [.][d]); writes the piece's normalised sums to part[slot][R16][d] (bf16) and (running
 *                     max in the log2 domain, weight sum) to ml[slot][R16][2]; R16 = R rounded up to 16.
 * v2s_decode_ctxfold: merges the pieces and applies the V projection: ctx[m][h*64 .. +64] = Wv_h accn[m][h] for the rows = entries * G
 *                     query rows (row m = entry m / G, beam m % G); wv bf16 [H*64][d]; ctx bf16 with row stride ld_ctx. */
int v2s_decode_qfold(const void* x, int64_t ldx, int32_t rows, const void* wq, const void* wkT, float rms_eps, void* qp, int32_t H,
                     int32_t d, void* stream);
int v2s_decode_memattn_plan(const int32_t* klen_host, int32_t entries, int32_t tiles_per_piece, int32_t max_blocks, int32_t* blk,
                            int32_t* slot_off, int32_t* nblk);
int v2s_decode_memattn(const void* qp, const void* mem, int64_t mem_es, const int32_t* blk, int32_t nblk, int32_t R, float scale,
                       void* part, float* ml, int32_t d, void* stream);
int v2s_decode_ctxfold(const void* part, const float* ml, const int32_t* slot_off, int32_t rows, int32_t G, int32_t H, const void* wv,
                       void* ctx, int64_t ld_ctx, int32_t d, void* stream);
int v2s_argmax_step(const float* logits, int64_t ld, int32_t rows, int32_t V, int64_t* next_tok,
                    int32_t* unfinished, int32_t eos_id, int32_t pad_id, void* stream);
/* same, and additionally stores the token at seq_out[row*seq_ld + *pos_dev + 1] (device-resident step counter) */
int v2s_argmax_step_seq(const float* logits, int64_t ld, int32_t rows, int32_t V, int64_t* next_tok,
                        int32_t* unfinished, int32_t eos_id, int32_t pad_id, int64_t* seq_out, int64_t seq_ld,
                        const int32_t* pos_dev, void* stream);
/* the whole tail of a greedy step in one launch (round 5): v2s_argmax_step_seq, then h_out[row, :] = table[token] (bf16 [vocab][d]: the
 * embedding lookup that opens the NEXT step, modeling_t5.py:968-975 through greedy_search's loop) and, by the last block to finish,
 * *pos_dev += 1 (what v2s_counter_add does).  ticket: one zero-initialised int32 the call owns between launches. */
int v2s_argmax_step_tail(const float* logits, int64_t ld, int32_t rows, int32_t V, int64_t* next_tok, int32_t* unfinished,
                         int32_t eos_id, int32_t pad_id, int64_t* seq_out, int64_t seq_ld, int32_t* pos_dev, const void* table,
                         void* h_out, int32_t d, int32_t vocab, int32_t* ticket, void* stream);
/* append new K/V rows ([B][H*64], strided) into the cache at position pos (or *pos_dev when pos_dev != NULL) */
int v2s_kv_append(const void* src, int64_t src_bs, void* cache, int64_t cache_bs, int64_t cache_rs,
                  int32_t B, int32_t width, int32_t pos, const int32_t* pos_dev, void* stream);
/* beam search (HF 4.28 beam_search, call site vid2seq.py:150-162): per row the K best of log_softmax(logits) + beam_scores[row],
 * sorted descending (K in {2,4,8,16,32}: specialised kernels; 33..512: K rounds over an LDS-resident row, for num_beams > 16); ban_token >= 0 is excluded from the candidates (not from the softmax) while
 * *pos_dev + 1 < min_length (HF's MinLengthLogitsProcessor on EOS, applied to log-probs; pos_dev = device step counter);
 * row_lse != NULL: use these row log-sum-exps instead of recomputing them (after v2s_repetition_penalty); and the beam reorder of the self-attention cache (modeling_t5.py:1771-1793):
 * dst[b, 0:len, :] = src[idx[b], 0:len, :] for [B][*][width] bf16 caches with batch stride bs and row stride rs */
int v2s_topk_logprob(const float* logits, int64_t ld, int32_t rows, int32_t V, int32_t K, const float* beam_scores,
                     float* out_val, int32_t* out_idx, int32_t ban_token, const int32_t* pos_dev, int32_t min_length, const float* row_lse,
                     void* stream);
/* HF 4.28 RepetitionPenaltyLogitsProcessor (vid2seq.py:159, dvc.py:182), in place, one row per block: tokens in hist[row][0..n), n =
 * *pos_dev + 1 (or n_static), are penalised once each.  row_lse == NULL: scores are raw logits (greedy_search).  row_lse != NULL:
 * beam_search semantics -- the penalty acts on log-probabilities; the row log-sum-exp is stored there and v2s_topk_logprob must be
 * given the same row_lse */
int v2s_repetition_penalty(float* scores, int64_t ld, int32_t rows, int32_t V, const int64_t* hist, int64_t hist_ld,
                           const int32_t* pos_dev, int32_t n_static, float penalty, float* row_lse, void* stream);
/* Beam bookkeeping on the device (no host round trip per step): what transformers==4.28.0 BeamSearchScorer.process / BeamHypotheses.add do
 * between two decoder steps for num_beams > 1, do_sample = False, early_stopping = False (call site vid2seq.py:150-162).  One block per batch
 * entry merges the entry's nb x K per-beam candidates (v2s_topk_logprob output) into its 2 nb best, turns EOS candidates of rank < nb into
 * finished hypotheses (the nb best per entry by sum_logprobs / len_pow[len] in double; len_pow [max_length + 1] device doubles =
 * n^length_penalty as the host computes it, so that scores and their ties are bit-identical to the Python scorer), the first nb others into the next beams
 * (next_tok / beam_scores / src_rows), marks the entry done when its heap is full and cannot be beaten (done[e], *ndone counts them), and
 * applies the beam permutation in place to the entry's rows of hist ([rows][max_length] int64 decoder ids so far; column *pos_dev + 1 gets
 * the new tokens) and of row_map (v2s_decode_attn; may be NULL).  *pos_dev = step index t (sequences hold t + 1 tokens).  State (caller
 * allocates, zero-initialised except heap_worst = 1e9): hyp_tok [B][nb][max_length], hyp_len / hyp_score / hyp_order [B][nb], heap_n /
 * heap_worst / heap_added / done [B], ndone [1].  num_beams <= 16, 2 <= K <= 32. */
int v2s_beam_advance(const float* cand_val, const int32_t* cand_tok, int32_t K, int32_t B, int32_t nb, int32_t eos_id, int32_t pad_id,
                     const double* len_pow, const int32_t* pos_dev, int64_t* hist, int64_t hist_ld, int32_t max_length,
                     int32_t* row_map, int64_t row_map_ld, int64_t* next_tok, float* beam_scores, int32_t* src_rows,
                     int32_t* hyp_tok, int32_t* hyp_len, double* hyp_score, int32_t* hyp_order, int32_t* heap_n,
                     double* heap_worst, int32_t* heap_added, int32_t* done, int32_t* ndone, void* stream);
int v2s_kv_gather(const void* src, void* dst, const int32_t* idx, int64_t bs, int64_t rs, int32_t B, int32_t len,
                  int32_t width, void* stream);
/* HF 4.28 MinLengthLogitsProcessor for greedy_search (generate(min_length=...), vid2seq.py:155): scores[r][token] = -inf for every row
 * while *pos_dev + 1 < min_length (decoder sequence = start token + *pos_dev decoded tokens) */
int v2s_ban_token(float* scores, int64_t ld, int32_t rows, int32_t V, int32_t token, const int32_t* pos_dev, int32_t min_length,
                  void* stream);
/* T5 span corruption of a 0-padded id batch on the device (util/t5.py:3-32 as used by dataset/dvc_dataset.py:127-145).
 * ids [B][ld_ids] int64, lens [B] valid lengths (<= max_len), noise [B][ld_noise] uint8 (1 = noise token; the reference draws it
 * with numpy's RNG on the host).  den_in / den_out rows receive the corrupted input / target sequence incl. the trailing EOS and
 * are 0-filled up to ld_in / ld_out -- the caller sizes them from the mask (row length = kept + spans + 1); out_lens [B][2] gets
 * the two lengths.  Sentinel k of a row = num_text_tokens - k.  Rows with lens <= 1 give [0] / [eos]. */
int v2s_span_corrupt(const int64_t* ids, int64_t ld_ids, const int32_t* lens, const uint8_t* noise, int64_t ld_noise, int32_t B,
                     int32_t max_len, int64_t num_text_tokens, int64_t eos, int64_t* den_in, int64_t ld_in, int64_t* den_out,
                     int64_t ld_out, int32_t* out_lens, void* stream);
/* one nucleus-sampling step (HF 4.28 sample(): temperature + top-k + top-p warpers + multinomial; call site vid2seq.py:150-162 with
 * do_sample=use_nucleus_sampling): per row, softmax(logits / temperature), keep the top_k most probable tokens (0 = no top-k filter;
 * HF's generation default top_k = 50 is in force whenever do_sample is set -- the reference never overrides it), of those the most
 * probable ones whose preceding mass is < top_p, draw from the renormalised kept set with a counter-based uniform number (seed, row,
 * *pos_dev) -- torch's RNG stream is not reproducible, so parity is distributional.  Finished rows emit pad; the token is also written
 * to seq_out[row][*pos_dev + 1] when seq_out != NULL.  EOS has probability 0 while *pos_dev + 1 < min_length
 * (MinLengthLogitsProcessor).  probs_out (optional, [rows][V]) receives the filtered, renormalised distribution (tests). */
int v2s_topp_sample_step(const float* logits, int64_t ld, int32_t rows, int32_t V, float top_p, float temperature, uint32_t seed,
                         int64_t* next_tok, int32_t* unfinished, int32_t eos_id, int32_t pad_id, int64_t* seq_out, int64_t seq_ld,
                         const int32_t* pos_dev, float* probs_out, int32_t min_length, int32_t top_k, void* stream);
/* candidates of one beam-sample step (HF 4.28 beam_sample(): do_sample with num_beams > 1 at the call site vid2seq.py:150-162).
 * Per beam row: score = (log_softmax(logits) [row_lse: a processor rewrote the logits against that normaliser; EOS banned while
 * *pos_dev + 1 < min_length] + beam_scores[row]) / temperature, kept set = TopK(max(top_k, 2)) then TopP(top_p, keep >= 2) of the
 * row; HF draws 2 * num_beams tokens without replacement from the softmax over all kept scores of a batch entry = the largest
 * keys score + Gumbel noise (counter-based hash of seed, *pos_dev, row, token).  The kernel writes the K kept candidates of each row
 * with the largest keys, sorted by key: out_val (scores), out_tok, out_key, each [rows][K]; unused slots hold -inf / 0 / -inf.
 * K in [1, 64], top_k in [0, 64] (0 = 64: a row's kept list has 64 slots). */
int v2s_beam_sample_cand(const float* logits, int64_t ld, int32_t rows, int32_t V, int32_t K, const float* beam_scores, float top_p,
                         float temperature, int32_t top_k, uint32_t seed, float* out_val, int32_t* out_tok, float* out_key,
                         int32_t ban_token, const int32_t* pos_dev, int32_t min_length, const float* row_lse, void* stream);
/* *ctr += delta (one thread; closes a captured decode step) */
int v2s_counter_add(int32_t* ctr, int32_t delta, void* stream);

#ifdef __cplusplus
}
#endif
#endif
