// Library plumbing: version, thread-local error text, runtime options.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include "v2s_common.h"

static thread_local char g_err[512] = "";
static std::atomic<int> g_tr_read{1};
static std::atomic<int> g_gemm_dma{2};
static std::atomic<int> g_gemm_big{1};
static std::atomic<int> g_gemm_split{1};
static std::atomic<int> g_gemm_order{4};
static std::atomic<int> g_gemm_skinny{1};
static std::atomic<int> g_attn_bwd_part{0};

void v2s_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int v2s_opt_tr_read() { return g_tr_read.load(std::memory_order_relaxed); }
int v2s_opt_gemm_dma() { return g_gemm_dma.load(std::memory_order_relaxed); }
int v2s_opt_gemm_big() { return g_gemm_big.load(std::memory_order_relaxed); }
int v2s_opt_gemm_split() { return g_gemm_split.load(std::memory_order_relaxed); }
int v2s_opt_gemm_order() { return g_gemm_order.load(std::memory_order_relaxed); }
int v2s_opt_gemm_skinny() { return g_gemm_skinny.load(std::memory_order_relaxed); }
int v2s_opt_attn_bwd_part() { return g_attn_bwd_part.load(std::memory_order_relaxed); }

extern "C" int v2s_version(void) { return V2S_ABI_VERSION; }
extern "C" const char* v2s_last_error(void) { return g_err; }

extern "C" int v2s_set_option(const char* name, int value) {
  if (name && strcmp(name, "tr_read") == 0) { g_tr_read.store(value ? 1 : 0); return V2S_OK; }
  if (name && strcmp(name, "gemm_dma") == 0) { g_gemm_dma.store(value); return V2S_OK; }
  if (name && strcmp(name, "gemm_big") == 0) { g_gemm_big.store(value); return V2S_OK; }
  if (name && strcmp(name, "gemm_split") == 0) { g_gemm_split.store(value); return V2S_OK; }
  if (name && strcmp(name, "gemm_order") == 0) { g_gemm_order.store(value); return V2S_OK; }
  if (name && strcmp(name, "gemm_skinny") == 0) { g_gemm_skinny.store(value); return V2S_OK; }
  if (name && strcmp(name, "attn_bwd_part") == 0) { g_attn_bwd_part.store(value); return V2S_OK; }
  v2s_set_error("v2s_set_option: unknown option '%s'", name ? name : "(null)");
  return V2S_ERR_ARG;
}
extern "C" int v2s_get_option(const char* name) {
  if (name && strcmp(name, "tr_read") == 0) return g_tr_read.load();
  if (name && strcmp(name, "gemm_dma") == 0) return g_gemm_dma.load();
  if (name && strcmp(name, "gemm_big") == 0) return g_gemm_big.load();
  if (name && strcmp(name, "gemm_split") == 0) return g_gemm_split.load();
  if (name && strcmp(name, "gemm_order") == 0) return g_gemm_order.load();
  if (name && strcmp(name, "gemm_skinny") == 0) return g_gemm_skinny.load();
  if (name && strcmp(name, "attn_bwd_part") == 0) return g_attn_bwd_part.load();
  v2s_set_error("v2s_get_option: unknown option '%s'", name ? name : "(null)");
  return V2S_ERR_ARG;
}
