// Input-side data formats on the device: T5 span corruption of a 0-padded id batch.
//
// Replaces, for a whole batch at once, util/t5.py:create_sentinel_ids (:3-16) + filter_input_ids (:19-32) as used by
// dataset/dvc_dataset.py:127-145.  Integer work, one pass over the ids: bit-exact with the reference by construction.
// Given the noise mask m of a row (the reference draws it with numpy's RNG on the host; so does data.py):
//   variant IN  (mask = m):   a masked token that starts a span becomes sentinel (num_text_tokens - k), k = 1,2,.. in order;
//                             the other masked tokens are dropped; unmasked tokens are kept
//   variant OUT (mask = ~m):  the same with the complement mask
// then EOS is appended and the row is 0-padded.  Rows of length <= 1 give IN = [0], OUT = [eos] (dvc_dataset.py:141-144).
#include "v2s_common.h"

namespace {

struct Cnt4 { int s_in, k_in, s_out, k_out; };

__global__ __launch_bounds__(256) void span_corrupt_kernel(const long* __restrict__ ids, long ld_ids, const int* __restrict__ lens,
                                                           const uint8_t* __restrict__ noise, long ld_noise, long ntext, long eos,
                                                           long* __restrict__ den_in, long ld_in, long* __restrict__ den_out, long ld_out,
                                                           int* __restrict__ out_lens) {
  __shared__ Cnt4 part[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = lens[b];
  const long* row = ids + (long)b * ld_ids;
  const uint8_t* m = noise + (long)b * ld_noise;
  long* oi = den_in + (long)b * ld_in;
  long* oo = den_out + (long)b * ld_out;
  if (n <= 1) {
    for (long i = tid; i < ld_in; i += 256) oi[i] = 0;
    for (long i = tid; i < ld_out; i += 256) oo[i] = (i == 0) ? eos : 0;
    if (tid == 0) { out_lens[2 * b] = 1; out_lens[2 * b + 1] = 1; }
    return;
  }
  const int per = (n + 255) / 256;
  const int beg = min(n, tid * per), end = min(n, beg + per);
  Cnt4 c = {0, 0, 0, 0};
  for (int i = beg; i < end; ++i) {
    const int mi = m[i] != 0, mp = (i > 0) ? (m[i - 1] != 0) : 0;
    const int st_in = mi & !mp, st_out = (!mi) & (i == 0 ? 1 : mp);
    c.s_in += st_in; c.k_in += (!mi) | st_in;
    c.s_out += st_out; c.k_out += mi | st_out;
  }
  part[tid] = c;
  __syncthreads();
  // exclusive block scan (256 partials; serial in one wave's worth of threads is plenty for <= 4 K tokens)
  if (tid == 0) {
    Cnt4 run = {0, 0, 0, 0};
    for (int t = 0; t < 256; ++t) {
      const Cnt4 v = part[t];
      part[t] = run;
      run.s_in += v.s_in; run.k_in += v.k_in; run.s_out += v.s_out; run.k_out += v.k_out;
    }
    out_lens[2 * b] = run.k_in + 1;
    out_lens[2 * b + 1] = run.k_out + 1;
    if (run.k_in < ld_in) oi[run.k_in] = eos;          // an undersized output row is truncated, never overrun
    if (run.k_out < ld_out) oo[run.k_out] = eos;
  }
  __syncthreads();
  Cnt4 o = part[tid];
  for (int i = beg; i < end; ++i) {
    const int mi = m[i] != 0, mp = (i > 0) ? (m[i - 1] != 0) : 0;
    const int st_in = mi & !mp, st_out = (!mi) & (i == 0 ? 1 : mp);
    const long v = row[i];
    if (st_in) { ++o.s_in; if (o.k_in < ld_in) oi[o.k_in] = ntext - o.s_in; ++o.k_in; } else if (!mi) { if (o.k_in < ld_in) oi[o.k_in] = v; ++o.k_in; }
    if (st_out) { ++o.s_out; if (o.k_out < ld_out) oo[o.k_out] = ntext - o.s_out; ++o.k_out; } else if (mi) { if (o.k_out < ld_out) oo[o.k_out] = v; ++o.k_out; }
  }
  // totals (thread 255's running offsets after its chunk == row totals because chunks are contiguous and ordered)
  __shared__ int tot[2];
  if (tid == 255) { tot[0] = o.k_in; tot[1] = o.k_out; }
  __syncthreads();
  for (long i = tot[0] + 1 + tid; i < ld_in; i += 256) oi[i] = 0;
  for (long i = tot[1] + 1 + tid; i < ld_out; i += 256) oo[i] = 0;
}

}  // namespace

extern "C" int v2s_span_corrupt(const int64_t* ids, int64_t ld_ids, const int32_t* lens, const uint8_t* noise, int64_t ld_noise, int32_t B,
                                int32_t max_len, int64_t num_text_tokens, int64_t eos, int64_t* den_in, int64_t ld_in, int64_t* den_out,
                                int64_t ld_out, int32_t* out_lens, void* stream) {
  V2S_CHECK(ids && lens && noise && den_in && den_out && out_lens && B > 0 && max_len > 0, V2S_ERR_ARG, "v2s_span_corrupt: bad args");
  V2S_CHECK(ld_ids >= max_len && ld_noise >= max_len, V2S_ERR_SHAPE, "v2s_span_corrupt: ld_ids/ld_noise must cover max_len=%d", max_len);
  V2S_CHECK(ld_in >= 1 && ld_out >= 1, V2S_ERR_SHAPE, "v2s_span_corrupt: empty outputs");
  V2S_CHECK(max_len <= 256 * 64, V2S_ERR_SHAPE, "v2s_span_corrupt: rows longer than 16384 tokens are not supported");
  hipLaunchKernelGGL(span_corrupt_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const long*)ids, (long)ld_ids, lens, noise,
                     (long)ld_noise, (long)num_text_tokens, (long)eos, (long*)den_in, (long)ld_in, (long*)den_out, (long)ld_out, out_lens);
  V2S_LAUNCH_CHECK();
  return V2S_OK;
}
