// Library plumbing: version, thread-local error text, runtime options, struct-size handshake.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include "v2s_common.h"

static thread_local char g_err[512] = "";

namespace {
struct Opt { const char* name; std::atomic<int> v; };
// index order = the V2S_OPT_* enumerators below
Opt g_opts[] = {
    {"tr_read", {1}}, {"gemm_dma", {2}}, {"gemm_big", {1}}, {"gemm_split", {1}}, {"gemm_order", {4}}, {"gemm_skinny", {1}},
    {"attn_bwd_part", {0}}, {"gemm_p8", {1}}, {"ce_fused", {1}}, {"gemm_dbg", {0}}, {"fp32_io", {0}}, {"gemm_a4", {1}}, {"gemm_a4_grid", {0}}, {"gemm_a4_relu", {1}}, {"gemm_a4_walk", {0}}, {"attn_order", {0}},
};
enum { O_TR_READ, O_GEMM_DMA, O_GEMM_BIG, O_GEMM_SPLIT, O_GEMM_ORDER, O_GEMM_SKINNY, O_ATTN_BWD_PART, O_GEMM_P8, O_CE_FUSED, O_GEMM_DBG, O_FP32_IO, O_GEMM_A4, O_GEMM_A4_GRID, O_GEMM_A4_RELU, O_GEMM_A4_WALK, O_ATTN_ORDER };
inline int opt(int i) { return g_opts[i].v.load(std::memory_order_relaxed); }
Opt* find_opt(const char* name) {
  if (!name) return nullptr;
  for (auto& o : g_opts)
    if (strcmp(name, o.name) == 0) return &o;
  return nullptr;
}
}  // namespace

void v2s_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int v2s_opt_tr_read() { return opt(O_TR_READ); }
int v2s_opt_gemm_dma() { return opt(O_GEMM_DMA); }
int v2s_opt_gemm_big() { return opt(O_GEMM_BIG); }
int v2s_opt_gemm_split() { return opt(O_GEMM_SPLIT); }
int v2s_opt_gemm_order() { return opt(O_GEMM_ORDER); }
int v2s_opt_gemm_skinny() { return opt(O_GEMM_SKINNY); }
int v2s_opt_attn_bwd_part() { return opt(O_ATTN_BWD_PART); }
int v2s_opt_gemm_p8() { return opt(O_GEMM_P8); }
int v2s_opt_ce_fused() { return opt(O_CE_FUSED); }
int v2s_opt_gemm_dbg() { return opt(O_GEMM_DBG); }
int v2s_opt_fp32_io() { return opt(O_FP32_IO); }
int v2s_opt_gemm_a4() { return opt(O_GEMM_A4); }
int v2s_opt_gemm_a4_grid() { return opt(O_GEMM_A4_GRID); }
int v2s_opt_gemm_a4_relu() { return opt(O_GEMM_A4_RELU); }
int v2s_opt_gemm_a4_walk() { return opt(O_GEMM_A4_WALK); }
int v2s_opt_attn_order() { return opt(O_ATTN_ORDER); }

// Dropout seed salt of the launches ENQUEUED BY THE CALLING THREAD (a device word XOR-ed into every by-value seed, so that a captured
// hipGraph draws new masks per replay).  Thread-local, not process-global: a capture running on one host thread
// cannot leak its salt into the eager launches another thread enqueues for another engine; on one thread the owner sets it around its
// capture and clears it (Trainer.step_graph).  The tuning options above stay process-wide by design: read-mostly A/B knobs.
static thread_local const uint32_t* g_seed_salt = nullptr;
const uint32_t* v2s_seed_salt() { return g_seed_salt; }
extern "C" int v2s_set_seed_salt(const uint32_t* dev_word) { g_seed_salt = dev_word; return V2S_OK; }

extern "C" int v2s_version(void) { return V2S_ABI_VERSION; }
extern "C" const char* v2s_last_error(void) { return g_err; }

extern "C" int v2s_set_option(const char* name, int value) {
  Opt* o = find_opt(name);
  if (!o) {
    v2s_set_error("v2s_set_option: unknown option '%s'", name ? name : "(null)");
    return V2S_ERR_ARG;
  }
  o->v.store(o == &g_opts[O_TR_READ] ? (value ? 1 : 0) : value);
  return V2S_OK;
}
extern "C" int v2s_get_option(const char* name) {
  Opt* o = find_opt(name);
  if (!o) {
    v2s_set_error("v2s_get_option: unknown option '%s'", name ? name : "(null)");
    return V2S_ERR_ARG;
  }
  return o->v.load();
}

// sizeof() of the argument structs as THIS library was compiled: a binding checks its own struct definitions against these
// before the first call (a struct that is short by one field makes the library read past the caller's buffer).
extern "C" int64_t v2s_sizeof(const char* name) {
  if (name) {
    if (strcmp(name, "v2s_gemm_args") == 0) return (int64_t)sizeof(v2s_gemm_args);
    if (strcmp(name, "v2s_attn_args") == 0) return (int64_t)sizeof(v2s_attn_args);
    if (strcmp(name, "v2s_adam_args") == 0) return (int64_t)sizeof(v2s_adam_args);
    if (strcmp(name, "v2s_decode_attn_args") == 0) return (int64_t)sizeof(v2s_decode_attn_args);
  }
  v2s_set_error("v2s_sizeof: unknown struct '%s'", name ? name : "(null)");
  return -1;
}
