"""ctypes binding of libvid2seq_hip.so (the C-ABI declared in include/vid2seq_hip.h).

PyTorch is used only for device memory and streams: every wrapper below takes torch tensors, passes raw
device pointers + sizes + the current HIP stream to the library, and raises ``RuntimeError`` with
``v2s_last_error()`` on a non-zero return.  There is NO fallback: if the shared library is missing the
import of :func:`lib` fails loudly (product path never routes through the CPU oracle).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("V2S_LIB") or os.path.join(_HERE, "libvid2seq_hip.so")      # V2S_LIB: developer override (profiling builds)

V2S_BF16, V2S_F32 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2

_vp = C.c_void_p
_i32, _i64, _f32, _u32 = C.c_int32, C.c_int64, C.c_float, C.c_uint32


class GemmArgs(C.Structure):
    _fields_ = [("M", _i32), ("N", _i32), ("K", _i32), ("transA", _i32), ("transB", _i32),
                ("A", _vp), ("B", _vp), ("lda", _i64), ("ldb", _i64), ("C", _vp), ("ldc", _i64),
                ("c_dtype", _i32), ("accumulate", _i32), ("alpha", _f32), ("bias", _vp), ("act", _i32),
                ("pre", _vp), ("dact", _i32), ("z", _vp), ("ldz", _i64), ("residual", _vp), ("ldr", _i64),
                ("dropout_p", _f32), ("dropout_seed", _u32), ("workspace", _vp), ("workspace_bytes", _i64),
                ("rms_eps", _f32), ("decode", _i32)]


class AttnArgs(C.Structure):
    _fields_ = [("B", _i32), ("H", _i32), ("Nq", _i32), ("Nk", _i32),
                ("q", _vp), ("k", _vp), ("v", _vp),
                ("q_bs", _i64), ("q_rs", _i64), ("k_bs", _i64), ("k_rs", _i64), ("v_bs", _i64), ("v_rs", _i64),
                ("o", _vp), ("o_bs", _i64), ("o_rs", _i64), ("ml", _vp), ("scale", _f32),
                ("bias_diag", _vp), ("key_mask", _vp), ("causal", _i32), ("causal_off", _i32),
                ("dropout_p", _f32), ("dropout_seed", _u32),
                ("d_o", _vp), ("do_bs", _i64), ("do_rs", _i64), ("delta", _vp),
                ("dq", _vp), ("dk", _vp), ("dv", _vp),
                ("dq_bs", _i64), ("dq_rs", _i64), ("dk_bs", _i64), ("dk_rs", _i64), ("dv_bs", _i64), ("dv_rs", _i64),
                ("dbias_diag", _vp), ("bias_far_lo", _i32), ("bias_far_hi", _i32), ("seq_off", _vp), ("seq_q_only", _i32), ("kv_seq_off", _vp)]


class AdamArgs(C.Structure):
    _fields_ = [("p", _vp), ("m", _vp), ("v", _vp), ("g", _vp), ("p_bf16", _vp), ("n", _i64),
                ("lr", _f32), ("beta1", _f32), ("beta2", _f32), ("eps", _f32), ("weight_decay", _f32),
                ("step", _i32), ("gnorm_sq", _vp), ("max_norm", _f32), ("grad_scale", _f32), ("hyper_dev", _vp)]


class DecodeAttnArgs(C.Structure):
    _fields_ = [("B", _i32), ("H", _i32), ("Nk", _i32), ("q", _vp), ("q_bs", _i64), ("k", _vp), ("v", _vp),
                ("kv_bs", _i64), ("kv_rs", _i64), ("o", _vp), ("o_bs", _i64), ("bias_row", _vp), ("bias_ld", _i64),
                ("key_mask", _vp), ("mask_ld", _i64), ("scale", _f32), ("pos_dev", _vp), ("bias_maxlen", _i32),
                ("kv_group", _i32), ("new_k", _vp), ("new_v", _vp), ("new_bs", _i64), ("row_map", _vp), ("row_map_ld", _i64)]


ABI_VERSION = 6   # V2S_ABI_VERSION this binding was written against (include/vid2seq_hip.h)

#: every symbol include/vid2seq_hip.h declares (checked by tests/test_oracle_cpu.py::test_c_abi_exports_every_declared_symbol)
SYMBOLS = {
    "v2s_version": (C.c_int, []),
    "v2s_last_error": (C.c_char_p, []),
    "v2s_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "v2s_get_option": (C.c_int, [C.c_char_p]),
    "v2s_sizeof": (_i64, [C.c_char_p]),
    "v2s_set_seed_salt": (C.c_int, [_vp]),
    "v2s_gemm": (C.c_int, [C.POINTER(GemmArgs), _vp]),
    "v2s_colsum": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i32, _vp]),
    "v2s_rmsnorm_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp]),
    "v2s_rmsnorm_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "v2s_layernorm_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp]),
    "v2s_layernorm_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "v2s_rmsnorm_bwd_drop": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _f32, _u32, _vp]),
    "v2s_layernorm_bwd_drop": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _f32, _u32, _vp]),
    "v2s_attn_fwd": (C.c_int, [C.POINTER(AttnArgs), _vp]),
    "v2s_attn_delta": (C.c_int, [C.POINTER(AttnArgs), _vp, _vp]),
    "v2s_attn_bwd": (C.c_int, [C.POINTER(AttnArgs), _vp]),
    "v2s_bias_diag_fwd": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "v2s_bias_bucket_bwd": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "v2s_embed_fwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _f32, _u32, _vp]),
    "v2s_embed_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _f32, _u32, _vp]),
    "v2s_add_bcast": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "v2s_dropout": (C.c_int, [_vp, _vp, _i64, _f32, _u32, _vp]),
    "v2s_add": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "v2s_sum_n": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _vp]),
    "v2s_clock_probe": (C.c_int, [_vp, _vp]),
    "v2s_bcast_grad": (C.c_int, [_vp, _vp, _i64, _i64, _vp]),
    "v2s_ce_fwd": (C.c_int, [_vp, _i64, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp]),
    "v2s_ce_bwd": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _i64, _vp]),
    "v2s_lmhead_ce_workspace_floats": (_i64, [_i32, _i32]),
    "v2s_lmhead_ce_fwd": (C.c_int, [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _f32, _vp, _vp, _vp, _vp, _vp]),
    "v2s_lmhead_ce_bwd": (C.c_int, [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _f32, _vp, _vp, _i64, _vp]),
    "v2s_sqnorm": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "v2s_adam_step": (C.c_int, [C.POINTER(AdamArgs), _vp]),
    "v2s_cast_bf16": (C.c_int, [_vp, _vp, _i64, _vp]),
    "v2s_timetoken_renorm": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "v2s_rowsumsq_range": (C.c_int, [_vp, _i32, _i32, _i64, _i64, _vp, _vp]),
    "v2s_timetoken_renorm_sq": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "v2s_decode_attn": (C.c_int, [C.POINTER(DecodeAttnArgs), _vp]),
    "v2s_decode_qfold": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _f32, _vp, _i32, _i32, _vp]),
    "v2s_decode_memattn_plan": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "v2s_decode_memattn": (C.c_int, [_vp, _vp, _i64, _vp, _i32, _i32, _f32, _vp, _vp, _i32, _vp]),
    "v2s_decode_ctxfold": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _i64, _i32, _vp]),
    "v2s_argmax_step": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _i32, _i32, _vp]),
    "v2s_argmax_step_seq": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp]),
    "v2s_argmax_step_tail": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "v2s_kv_append": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp]),
    "v2s_counter_add": (C.c_int, [_vp, _i32, _vp]),
    "v2s_last_gemm_kernel": (C.c_char_p, []),
    "v2s_gemm_grouped": (C.c_int, [C.POINTER(GemmArgs), _i32, _vp, _vp, _vp, _vp]),
    "v2s_scale_cols": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "v2s_span_corrupt": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp]),
    "v2s_topk_logprob": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp]),
    "v2s_topp_sample_step": (C.c_int, [_vp, _i64, _i32, _i32, _f32, _f32, _u32, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp, _i32, _i32, _vp]),
    "v2s_beam_sample_cand": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _f32, _f32, _i32, _u32, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp]),
    "v2s_repetition_penalty": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _vp, _i32, _f32, _vp, _vp]),
    "v2s_kv_gather": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp]),
    "v2s_beam_advance": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _vp, _vp, _vp,
                                   _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "v2s_ban_token": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _i32, _vp]),
}

_LIB = None


def lib() -> C.CDLL:
    """Load the HIP library (once).  Raises if it has not been built -- there is no fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with vidchapters_amd/csrc/build.sh (or __graft_entry__.build()). "
                "The Vid2Seq hot path has no non-HIP fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        # ABI handshake: the version this binding was written against, then the argument-struct sizes
        got = int(l.v2s_version())
        if got != ABI_VERSION:
            raise RuntimeError(f"ABI mismatch: {LIB_PATH} reports V2S_ABI_VERSION {got}, this binding needs {ABI_VERSION} "
                               "(rebuild the library: vidchapters_amd/csrc/build.sh)")
        for cname, ctype in (("v2s_gemm_args", GemmArgs), ("v2s_attn_args", AttnArgs), ("v2s_adam_args", AdamArgs),
                             ("v2s_decode_attn_args", DecodeAttnArgs)):
            want = int(l.v2s_sizeof(cname.encode()))
            if want != C.sizeof(ctype):
                raise RuntimeError(f"ABI mismatch: {cname} is {want} bytes in {LIB_PATH} but {C.sizeof(ctype)} in the ctypes binding "
                                   "(rebuild the library or update vidchapters_amd/lib.py)")
        # developer override of the runtime options (include/vid2seq_hip.h): V2S_OPTIONS="gemm_p8=2,gemm_big=0"
        for kv in filter(None, os.environ.get("V2S_OPTIONS", "").split(",")):
            k, v = kv.split("=")
            if l.v2s_set_option(k.strip().encode(), int(v)) != 0:
                raise RuntimeError(f"V2S_OPTIONS: {l.v2s_last_error().decode()}")
        _LIB = l
    return _LIB


launch_count = 0      # successful library calls so far (one kernel launch each on the decode path): Engine.last_decode_launches, bench.py


def _check(rc: int, what: str) -> None:
    global launch_count
    launch_count += 1
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {lib().v2s_last_error().decode()}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def set_option(name: str, value: int) -> None:
    _check(lib().v2s_set_option(name.encode(), int(value)), "v2s_set_option")


def get_option(name: str) -> int:
    return int(lib().v2s_get_option(name.encode()))


_FP32_IO = False


class fp32_io:
    """Debug context (library option "fp32_io"): inside it the norm / cross-entropy-backward / attention entry points
    take and return FP32 activations (attention: fp32-arithmetic reference kernels).  Parity tests only -- the product path never
    enters it; GEMMs are not affected (their fp32-output form exists already)."""

    def __enter__(self):
        global _FP32_IO
        set_option("fp32_io", 1); _FP32_IO = True
        return self

    def __exit__(self, *exc):
        global _FP32_IO
        set_option("fp32_io", 0); _FP32_IO = False
        return False


def _need_act(t: torch.Tensor, what: str) -> None:
    """An activation operand: bf16, or fp32 inside the fp32_io debug context."""
    _need(t, torch.float32 if _FP32_IO else torch.bfloat16, what)


def _need(t: torch.Tensor, dtype, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor must live on the GPU (HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise RuntimeError(f"{what}: expected dtype {dtype}, got {t.dtype}")


# --------------------------------------------------------------------------------------------- live kernel timing
class KernelTimer:
    """HIP-event timing of individual launches on the stream they are issued on (bench.py roofline leg).
    Usage: ``with KernelTimer() as kt: step()`` then ``kt.summary()`` -> {tag: (launches, total_ms, total_work)}."""
    active = None

    def __init__(self, detail: bool = False, by_symbol: bool = False):
        self.records = []
        self.detail = detail
        self.by_symbol = by_symbol        # tag GEMMs with the dispatched kernel symbol instead of the role

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        return e

    def end(self, tag: str, work: float, e0) -> None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(torch.cuda.current_stream())
        self.records.append((tag, work, e0, e1))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for tag, work, e0, e1 in self.records:
            n, ms, w = out.get(tag, (0, 0.0, 0.0))
            out[tag] = (n + 1, ms + e0.elapsed_time(e1), w + work)
        return out


# --------------------------------------------------------------------------------------------- GEMM
def gemm(A: torch.Tensor, B: torch.Tensor, C_out: torch.Tensor, M: int, N: int, K: int, *, transA=False, transB=False,
         lda=None, ldb=None, ldc=None, accumulate=False, alpha=1.0, bias=None, act=ACT_NONE, pre=None, dact=ACT_NONE,
         z=None, ldz=None, residual=None, ldr=None, dropout_p=0.0, dropout_seed=0, workspace=None, rms_eps=0.0, decode=False) -> None:
    """C[M,N] (+)= epilogue(alpha * A(m,k) B(n,k)); see include/vid2seq_hip.h for layouts."""
    _need(A, torch.bfloat16, "gemm A"); _need(B, torch.bfloat16, "gemm B")
    a = GemmArgs()
    a.M, a.N, a.K = M, N, K
    a.transA, a.transB = int(transA), int(transB)
    a.A, a.B = A.data_ptr(), B.data_ptr()
    a.lda = lda if lda is not None else (M if transA else K)
    a.ldb = ldb if ldb is not None else (N if transB else K)
    a.C = C_out.data_ptr()
    a.ldc = ldc if ldc is not None else N
    a.c_dtype = V2S_F32 if C_out.dtype == torch.float32 else V2S_BF16
    a.accumulate = int(accumulate)
    a.alpha = alpha
    a.bias = ptr(bias)
    a.act = act
    a.pre = ptr(pre)
    a.dact = dact
    a.z = ptr(z)
    a.ldz = ldz if ldz is not None else N
    a.residual = ptr(residual)
    a.ldr = ldr if ldr is not None else N
    a.dropout_p = dropout_p
    a.dropout_seed = dropout_seed & 0xFFFFFFFF
    a.rms_eps = rms_eps
    a.decode = 1 if decode else 0
    if workspace is not None:
        a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    kt = KernelTimer.active
    if kt is not None:
        e0 = kt.begin()
    _check(lib().v2s_gemm(C.byref(a), stream_ptr()), "v2s_gemm")
    if kt is not None:
        kind = "gemm_wgrad" if transA else ("gemm_dgrad" if transB else "gemm_nt")
        if kt.by_symbol:
            kind = lib().v2s_last_gemm_kernel().decode() + (" + splitk_reduce_kernel" if workspace is not None else "")
        if kt.detail:
            kind += f":{M}x{N}x{K}" + (":drop" if dropout_p > 0 else "") + (":res" if residual is not None else "") + \
                    (":act" if act or dact else "") + (":bias" if bias is not None else "") + (":f32" if a.c_dtype == V2S_F32 else "")
        kt.end(kind, 2.0 * M * N * K, e0)


def gemm_grouped(As, Bs, Cs, M: int, N: int, K: int, *, lda=None, ldb=None, ldc=None, accumulate=False, alpha=1.0) -> None:
    """Cs[i][M, N] (+)= alpha * As[i]^T @ Bs[i] (As[i]: bf16 [K, lda], Bs[i]: bf16 [K, ldb], Cs[i]: fp32) for up to 16 problems of one shape
    in ONE launch (v2s_gemm_grouped: the same weight gradient of many layers)."""
    n = len(As)
    if not (n == len(Bs) == len(Cs) and 1 <= n <= 16):
        raise ValueError(f"gemm_grouped: 1..16 problems with one A, B and C each (got {n}, {len(Bs)}, {len(Cs)})")
    for t in As + Bs:
        _need(t, torch.bfloat16, "gemm_grouped operand")
    for t in Cs:
        _need(t, torch.float32, "gemm_grouped C")
    a = GemmArgs()
    a.M, a.N, a.K = M, N, K
    a.transA, a.transB = 1, 1
    a.lda = lda if lda is not None else M
    a.ldb = ldb if ldb is not None else N
    a.ldc = ldc if ldc is not None else N
    a.c_dtype = V2S_F32
    a.accumulate = int(accumulate)
    a.alpha = alpha
    arr = C.c_void_p * n
    pa, pb, pc = arr(*[t.data_ptr() for t in As]), arr(*[t.data_ptr() for t in Bs]), arr(*[t.data_ptr() for t in Cs])
    kt = KernelTimer.active
    if kt is not None:
        e0 = kt.begin()
    _check(lib().v2s_gemm_grouped(C.byref(a), n, pa, pb, pc, stream_ptr()), "v2s_gemm_grouped")
    if kt is not None:
        kind = "gemm_dma_grouped_kernel" if kt.by_symbol else "gemm_wgrad"
        if kt.detail:
            kind += f":{n}x{M}x{N}x{K}"
        kt.end(kind, 2.0 * n * M * N * K, e0)


def colsum(X: torch.Tensor, M: int, N: int, out: torch.Tensor, accumulate=True, ldx=None) -> None:
    _need(X, torch.bfloat16, "colsum X"); _need(out, torch.float32, "colsum out")
    _check(lib().v2s_colsum(X.data_ptr(), ldx if ldx is not None else N, M, N, out.data_ptr(), int(accumulate), stream_ptr()),
           "v2s_colsum")


# --------------------------------------------------------------------------------------------- norms
def rmsnorm_fwd(x, w, y, rstd, rows, cols, eps):
    _need_act(x, "rmsnorm x"); _need(w, torch.float32, "rmsnorm w")
    _check(lib().v2s_rmsnorm_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), rstd.data_ptr(), rows, cols, eps, stream_ptr()),
           "v2s_rmsnorm_fwd")


def rmsnorm_bwd(x, w, rstd, dy, dx, dx_add, dw, rows, cols, dx_drop=None, dropout_p=0.0, dropout_seed=0):
    """``dx_drop`` (bf16, optional): second output dropout(dx; dropout_p, dropout_seed), same mask / values as ``dropout(dx, ...)``."""
    if dx_drop is not None:
        _check(lib().v2s_rmsnorm_bwd_drop(x.data_ptr(), w.data_ptr(), rstd.data_ptr(), dy.data_ptr(), dx.data_ptr(), ptr(dx_add),
                                          dw.data_ptr(), rows, cols, dx_drop.data_ptr(), dropout_p, dropout_seed & 0xFFFFFFFF, stream_ptr()),
               "v2s_rmsnorm_bwd_drop")
        return
    _check(lib().v2s_rmsnorm_bwd(x.data_ptr(), w.data_ptr(), rstd.data_ptr(), dy.data_ptr(), dx.data_ptr(), ptr(dx_add),
                                 dw.data_ptr(), rows, cols, stream_ptr()), "v2s_rmsnorm_bwd")


def layernorm_fwd(x, w, b, y, mean, rstd, rows, cols, eps):
    _need_act(x, "layernorm x")
    _check(lib().v2s_layernorm_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                   rows, cols, eps, stream_ptr()), "v2s_layernorm_fwd")


def layernorm_bwd(x, w, mean, rstd, dy, dx, dx_add, dw, db, rows, cols, dx_drop=None, dropout_p=0.0, dropout_seed=0):
    if dx_drop is not None:
        _check(lib().v2s_layernorm_bwd_drop(x.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dy.data_ptr(), dx.data_ptr(),
                                            ptr(dx_add), dw.data_ptr(), db.data_ptr(), rows, cols, dx_drop.data_ptr(), dropout_p,
                                            dropout_seed & 0xFFFFFFFF, stream_ptr()), "v2s_layernorm_bwd_drop")
        return
    _check(lib().v2s_layernorm_bwd(x.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dy.data_ptr(), dx.data_ptr(),
                                   ptr(dx_add), dw.data_ptr(), db.data_ptr(), rows, cols, stream_ptr()),
           "v2s_layernorm_bwd")


# --------------------------------------------------------------------------------------------- attention
def attn_args(B, H, Nq, Nk, q, k, v, o, q_st, k_st, v_st, o_st, *, ml=None, scale=1.0, bias_diag=None, key_mask=None,
              causal=False, causal_off=0, dropout_p=0.0, dropout_seed=0, seq_off=None, seq_q_only=False, kv_seq_off=None) -> AttnArgs:
    """q_st etc. are (batch_stride, row_stride) in elements; q/k/v/o are tensors whose data_ptr() already
    points at column 0 of head 0."""
    a = AttnArgs()
    a.B, a.H, a.Nq, a.Nk = B, H, Nq, Nk
    a.q, a.k, a.v, a.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr()
    a.q_bs, a.q_rs = q_st; a.k_bs, a.k_rs = k_st; a.v_bs, a.v_rs = v_st; a.o_bs, a.o_rs = o_st
    a.ml = ptr(ml); a.scale = scale; a.bias_diag = ptr(bias_diag); a.key_mask = ptr(key_mask)
    a.causal, a.causal_off = int(causal), causal_off
    a.dropout_p, a.dropout_seed = dropout_p, dropout_seed & 0xFFFFFFFF
    if seq_off is not None:
        _need(seq_off, torch.int32, "attn seq_off")
    a.seq_off = ptr(seq_off)
    a.seq_q_only = 1 if (seq_q_only and seq_off is not None) else 0
    if kv_seq_off is not None:
        _need(kv_seq_off, torch.int32, "attn kv_seq_off")
    a.kv_seq_off = ptr(kv_seq_off)
    # the struct only carries raw pointers: keep every tensor alive for as long as the struct (backward reuses it)
    a._refs = (q, k, v, o, ml, bias_diag, key_mask, seq_off, kv_seq_off)
    return a


def attn_fwd(a: AttnArgs) -> None:
    kt = KernelTimer.active
    if kt is not None:
        e0 = kt.begin()
    _check(lib().v2s_attn_fwd(C.byref(a), stream_ptr()), "v2s_attn_fwd")
    if kt is not None:
        tag = "attn_fwd"
        if kt.by_symbol:
            tag = "attn_fwd_kernel<" + ", ".join("true" if f else "false" for f in (get_option("tr_read") != 0, bool(a.bias_diag), bool(a.causal), a.dropout_p > 0)) + ">"
        kt.end(tag, 4.0 * a.B * a.H * a.Nq * a.Nk * 64, e0)


def attn_bwd(a: AttnArgs, d_o, do_st, delta, dq, dk, dv, dq_st, dk_st, dv_st, dbias_diag=None, far=(0, 0)) -> None:
    a.d_o = d_o.data_ptr(); a.do_bs, a.do_rs = do_st
    _need(delta, torch.float32, "attn_bwd row-statistics workspace (delta)")
    if delta.numel() < a.B * a.H * a.Nq * 4:
        raise ValueError(f"attn_bwd: `delta` is the fp32 [B, H, Nq, 4] row-statistics workspace of v2s_attn_bwd ({a.B * a.H * a.Nq * 4} floats), got {delta.numel()}")
    a.delta = delta.data_ptr()
    a.dq, a.dk, a.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    a.dq_bs, a.dq_rs = dq_st; a.dk_bs, a.dk_rs = dk_st; a.dv_bs, a.dv_rs = dv_st
    a.dbias_diag = ptr(dbias_diag)
    a.bias_far_lo, a.bias_far_hi = far
    kt = KernelTimer.active
    if kt is not None and kt.by_symbol:          # time the two kernels of the call separately, tagged with their symbols
        flags = ", ".join("true" if f else "false" for f in (get_option("tr_read") != 0, bool(a.bias_diag), bool(a.causal), a.dropout_p > 0))
        work = 8.0 * a.B * a.H * a.Nq * a.Nk * 64
        for part, name, share in ((1, "attn_bwd_dq_kernel", 0.5), (2, "attn_bwd_dkv_kernel", 0.5)):
            set_option("attn_bwd_part", part)
            e0 = kt.begin()
            _check(lib().v2s_attn_bwd(C.byref(a), stream_ptr()), "v2s_attn_bwd")
            kt.end(f"{name}<{flags}>", work * share, e0)     # algorithmic: dP+dQ (dq kernel), dV+dK (dkv kernel); recompute excluded
        set_option("attn_bwd_part", 0)
        return
    if kt is not None:
        e0 = kt.begin()
    _check(lib().v2s_attn_bwd(C.byref(a), stream_ptr()), "v2s_attn_bwd")
    if kt is not None:
        kt.end("attn_bwd", 8.0 * a.B * a.H * a.Nq * a.Nk * 64, e0)   # algorithmic: dV, dP, dQ, dK (recompute excluded)


def bias_diag_fwd(table, lut, out, H, n, num_buckets):
    _check(lib().v2s_bias_diag_fwd(table.data_ptr(), lut.data_ptr(), out.data_ptr(), H, n, num_buckets, stream_ptr()),
           "v2s_bias_diag_fwd")


def bias_bucket_bwd(dd, lut, dtable, H, n, num_buckets):
    _check(lib().v2s_bias_bucket_bwd(dd.data_ptr(), lut.data_ptr(), dtable.data_ptr(), H, n, num_buckets, stream_ptr()),
           "v2s_bias_bucket_bwd")


# --------------------------------------------------------------------------------------------- misc
def embed_fwd(ids, table, out, n, d, vocab, p=0.0, seed=0):
    _need(ids, torch.int64, "embed ids"); _need(table, torch.bfloat16, "embed table")
    _check(lib().v2s_embed_fwd(ids.data_ptr(), table.data_ptr(), out.data_ptr(), n, d, vocab, p, seed & 0xFFFFFFFF, stream_ptr()),
           "v2s_embed_fwd")


def embed_bwd(ids, dy, dtable, n, d, vocab, p=0.0, seed=0):
    _need(dtable, torch.float32, "embed dtable")
    _check(lib().v2s_embed_bwd(ids.data_ptr(), dy.data_ptr(), dtable.data_ptr(), n, d, vocab, p, seed & 0xFFFFFFFF, stream_ptr()),
           "v2s_embed_bwd")


def add_bcast(x, add, y, n, add_n):
    _check(lib().v2s_add_bcast(x.data_ptr(), add.data_ptr(), y.data_ptr(), n, add_n, stream_ptr()), "v2s_add_bcast")


def add(a, b, y, n):
    _check(lib().v2s_add(a.data_ptr(), b.data_ptr(), y.data_ptr(), n, stream_ptr()), "v2s_add")


def sum_n(parts, stride, nparts, y, n):
    """y[n] = sum_p parts[p * stride + :n] (bf16 in / out, fp32 sum)"""
    _check(lib().v2s_sum_n(parts.data_ptr(), stride, nparts, y.data_ptr(), n, stream_ptr()), "v2s_sum_n")


def lmhead_ce_workspace_floats(rows, vpad):
    return int(lib().v2s_lmhead_ce_workspace_floats(rows, vpad))


def lmhead_ce_fwd(h, ldh, E, rows, V, vpad, d, alpha, labels, eps, part, row_out, loss_sum, count):
    """tied LM head + label-smoothed CE forward without logits in memory: row_out[rows, 2] = (log-sum-exp, loss), loss_sum / count accumulate"""
    _need(h, torch.bfloat16, "lmhead h"); _need(E, torch.bfloat16, "lmhead E"); _need(part, torch.float32, "lmhead workspace")
    kt = KernelTimer.active
    if kt is not None:
        e0 = kt.begin()
    _check(lib().v2s_lmhead_ce_fwd(h.data_ptr(), ldh, E.data_ptr(), rows, V, vpad, d, alpha, labels.data_ptr(), eps, part.data_ptr(), row_out.data_ptr(),
                                   loss_sum.data_ptr(), count.data_ptr(), stream_ptr()), "v2s_lmhead_ce_fwd")
    if kt is not None:
        kt.end("lmhead_ce_kernel<0>" if kt.by_symbol else "lmhead_ce_fwd", 2.0 * rows * vpad * d, e0)


def lmhead_ce_bwd(h, ldh, E, rows, V, vpad, d, alpha, labels, row_out, eps, gscale, dlogits, ldd):
    """d(logits) (bf16 [rows, ldd]) of the same head, tiles recomputed"""
    _need(h, torch.bfloat16, "lmhead h"); _need(E, torch.bfloat16, "lmhead E"); _need(dlogits, torch.bfloat16, "lmhead d(logits)")
    kt = KernelTimer.active
    if kt is not None:
        e0 = kt.begin()
    _check(lib().v2s_lmhead_ce_bwd(h.data_ptr(), ldh, E.data_ptr(), rows, V, vpad, d, alpha, labels.data_ptr(), row_out.data_ptr(), eps, gscale.data_ptr(),
                                   dlogits.data_ptr(), ldd, stream_ptr()), "v2s_lmhead_ce_bwd")
    if kt is not None:
        kt.end("lmhead_ce_kernel<1>" if kt.by_symbol else "lmhead_ce_bwd", 2.0 * rows * vpad * d, e0)


def clock_probe(out):
    """out: int64 [8, 4] (zeroed) <- per XCD (shader-clock cycles, 100 MHz ticks, xcd id, 1) at the point of the current stream"""
    _check(lib().v2s_clock_probe(out.data_ptr(), stream_ptr()), "v2s_clock_probe")


def effective_sclk_mhz(p0, p1):
    """average shader clock (MHz) between two clock_probe() results (host tensors / arrays [8, 4]); None if no XCD reported both times"""
    vals = []
    for a, b in zip(p0.tolist(), p1.tolist()):
        if a[3] == 1 and b[3] == 1 and b[1] > a[1]:
            vals.append((b[0] - a[0]) / (b[1] - a[1]) * 100.0)
    return sum(vals) / len(vals) if vals else None


def dropout(x, y, n, p, seed):
    _check(lib().v2s_dropout(x.data_ptr(), y.data_ptr(), n, p, seed & 0xFFFFFFFF, stream_ptr()), "v2s_dropout")


def bcast_grad(dy, out, n, add_n):
    _check(lib().v2s_bcast_grad(dy.data_ptr(), out.data_ptr(), n, add_n, stream_ptr()), "v2s_bcast_grad")


def ce_fwd(logits, ld, labels, rows, V, eps, row_lse, loss_sum, count):
    _need(logits, torch.float32, "ce logits"); _need(labels, torch.int64, "ce labels")
    _check(lib().v2s_ce_fwd(logits.data_ptr(), ld, labels.data_ptr(), rows, V, eps, row_lse.data_ptr(), loss_sum.data_ptr(),
                            count.data_ptr(), stream_ptr()), "v2s_ce_fwd")


def ce_bwd(logits, ld, labels, row_lse, rows, V, eps, gscale, dlogits, ldd):
    _check(lib().v2s_ce_bwd(logits.data_ptr(), ld, labels.data_ptr(), row_lse.data_ptr(), rows, V, eps, gscale.data_ptr(),
                            dlogits.data_ptr(), ldd, stream_ptr()), "v2s_ce_bwd")


def sqnorm(g, n, ws, out):
    _check(lib().v2s_sqnorm(g.data_ptr(), n, ws.data_ptr(), out.data_ptr(), stream_ptr()), "v2s_sqnorm")


def set_seed_salt(word: Optional[torch.Tensor]) -> None:
    """Device int32/uint32 word XOR-ed into every dropout seed of the launches enqueued from now on (None: off)."""
    _check(lib().v2s_set_seed_salt(None if word is None else word.data_ptr()), "v2s_set_seed_salt")


def adam_step(p, m, v, g, p_bf16, n, lr, beta1, beta2, eps, wd, step, gnorm_sq=None, max_norm=0.0, grad_scale=1.0, hyper_dev=None):
    a = AdamArgs()
    a.p, a.m, a.v, a.g, a.p_bf16, a.n = p.data_ptr(), m.data_ptr(), v.data_ptr(), g.data_ptr(), ptr(p_bf16), n
    a.lr, a.beta1, a.beta2, a.eps, a.weight_decay, a.step = lr, beta1, beta2, eps, wd, step
    a.gnorm_sq, a.max_norm, a.grad_scale = ptr(gnorm_sq), max_norm, grad_scale
    a.hyper_dev = ptr(hyper_dev)
    _check(lib().v2s_adam_step(C.byref(a), stream_ptr()), "v2s_adam_step")


def cast_bf16(src, dst, n):
    _need(src, torch.float32, "cast src")
    _check(lib().v2s_cast_bf16(src.data_ptr(), dst.data_ptr(), n, stream_ptr()), "v2s_cast_bf16")


def timetoken_renorm(emb, emb_bf16, V, d, num_bins, ws):
    _check(lib().v2s_timetoken_renorm(emb.data_ptr(), ptr(emb_bf16), V, d, num_bins, ws.data_ptr(), stream_ptr()),
           "v2s_timetoken_renorm")


def rowsumsq_range(emb, V, d, f0, f1, sumsq):
    """sumsq[row] += sum of squares of the elements of ``emb`` ([V, d] fp32) whose flat index lies in [f0, f1)."""
    _check(lib().v2s_rowsumsq_range(emb.data_ptr(), V, d, f0, f1, sumsq.data_ptr(), stream_ptr()), "v2s_rowsumsq_range")


def timetoken_renorm_sq(emb, emb_bf16, V, d, num_bins, text_sumsq, ws):
    _check(lib().v2s_timetoken_renorm_sq(emb.data_ptr(), ptr(emb_bf16), V, d, num_bins, text_sumsq.data_ptr(), ws.data_ptr(), stream_ptr()),
           "v2s_timetoken_renorm_sq")


def decode_attn(B, H, Nk, q, q_bs, k, v, kv_bs, kv_rs, o, o_bs, bias_row=None, bias_ld=0, key_mask=None, mask_ld=0, scale=1.0,
                pos_dev=None, bias_maxlen=0, kv_group=0, new_k=None, new_v=None, new_bs=0, row_map=None, row_map_ld=0):
    a = DecodeAttnArgs()
    a.B, a.H, a.Nk = B, H, Nk
    a.q, a.q_bs, a.k, a.v, a.kv_bs, a.kv_rs = q.data_ptr(), q_bs, k.data_ptr(), v.data_ptr(), kv_bs, kv_rs
    a.o, a.o_bs, a.bias_row, a.bias_ld, a.key_mask, a.mask_ld, a.scale = o.data_ptr(), o_bs, ptr(bias_row), bias_ld, ptr(key_mask), mask_ld, scale
    a.pos_dev, a.bias_maxlen, a.kv_group = ptr(pos_dev), bias_maxlen, kv_group
    a.new_k, a.new_v, a.new_bs = ptr(new_k), ptr(new_v), new_bs
    a.row_map, a.row_map_ld = ptr(row_map), row_map_ld
    _check(lib().v2s_decode_attn(C.byref(a), stream_ptr()), "v2s_decode_attn")


def decode_qfold(x, rows, wq, wkT, rms_eps, qp, H, d, ldx=None):
    """Folded cross-attention queries of a decode step: qp[rows, H, d] = per head ((rstd * x) Wq_h^T) Wk_h (v2s_decode_qfold)."""
    for t, n in ((x, "x"), (wq, "wq"), (wkT, "wkT"), (qp, "qp")):
        _need(t, torch.bfloat16, "decode_qfold " + n)
    _check(lib().v2s_decode_qfold(x.data_ptr(), ldx if ldx is not None else d, rows, wq.data_ptr(), wkT.data_ptr(), rms_eps, qp.data_ptr(),
                                  H, d, stream_ptr()), "v2s_decode_qfold")


class MemAttnPlan:
    """Host plan of the decode step's memory cross-attention (v2s_decode_memattn_plan): which block takes which key tiles of which
    entry, and where the pieces of an entry land.  Built once per generate() call from the entries' valid memory lengths; owns the
    device copies of the tables and the partial-sum buffers.  An entry's cut depends on its own length only (bit-identical results in
    any batch); 9 tiles = 288 keys per piece: the reference's longest memory (100 frames + 1000 tokens) is 4 pieces, 64 entries = 256 blocks."""

    def __init__(self, klen, R, device, tiles_per_piece=9):
        import numpy as np
        klen = np.ascontiguousarray(np.asarray(klen, dtype=np.int32))
        E = int(klen.shape[0])
        cap = int(((np.maximum(klen, 1) + 31) // 32 + tiles_per_piece - 1).sum() // tiles_per_piece) + E
        blk = np.zeros((cap, 4), dtype=np.int32); off = np.zeros(E + 1, dtype=np.int32); nb = C.c_int32(0)
        _check(lib().v2s_decode_memattn_plan(klen.ctypes.data, E, tiles_per_piece, cap, blk.ctypes.data, off.ctypes.data, C.byref(nb)),
               "v2s_decode_memattn_plan")
        self.entries, self.R, self.nblk = E, R, int(nb.value)
        self.blk_host, self.slot_off_host = blk[:self.nblk].copy(), off
        self.blk = torch.from_numpy(self.blk_host).to(device)
        self.slot_off = torch.from_numpy(off).to(device)
        qr = (R + 15) // 16 * 16
        self.part = torch.empty(self.nblk * qr * 768, dtype=torch.bfloat16, device=device)
        self.ml = torch.empty(self.nblk * qr * 2, dtype=torch.float32, device=device)


def decode_memattn(qp, mem, mem_es, plan: MemAttnPlan, d, scale=1.0):
    _need(qp, torch.bfloat16, "decode_memattn qp"); _need(mem, torch.bfloat16, "decode_memattn mem")
    if qp.numel() != plan.entries * plan.R * d or mem.numel() < (plan.entries - 1) * mem_es + d:
        raise ValueError(f"decode_memattn: qp holds {qp.numel()} elements for {plan.entries} entries x {plan.R} rows x {d}, memory {mem.numel()} "
                         f"elements for entry stride {mem_es}")
    _check(lib().v2s_decode_memattn(qp.data_ptr(), mem.data_ptr(), mem_es, plan.blk.data_ptr(), plan.nblk, plan.R, scale,
                                    plan.part.data_ptr(), plan.ml.data_ptr(), d, stream_ptr()), "v2s_decode_memattn")


def decode_ctxfold(plan: MemAttnPlan, rows, G, H, wv, ctx, d, ld_ctx=None):
    _need(wv, torch.bfloat16, "decode_ctxfold wv"); _need(ctx, torch.bfloat16, "decode_ctxfold ctx")
    if rows != plan.entries * G or G * H != plan.R:
        raise ValueError(f"decode_ctxfold: {rows} rows / {G} beams x {H} heads do not match the plan ({plan.entries} entries, {plan.R} query rows each)")
    _check(lib().v2s_decode_ctxfold(plan.part.data_ptr(), plan.ml.data_ptr(), plan.slot_off.data_ptr(), rows, G, H, wv.data_ptr(),
                                    ctx.data_ptr(), ld_ctx if ld_ctx is not None else H * 64, d, stream_ptr()), "v2s_decode_ctxfold")


def argmax_step(logits, ld, rows, V, next_tok, unfinished, eos_id, pad_id):
    _check(lib().v2s_argmax_step(logits.data_ptr(), ld, rows, V, next_tok.data_ptr(), unfinished.data_ptr(), eos_id, pad_id,
                                 stream_ptr()), "v2s_argmax_step")


def kv_append(src, src_bs, cache, cache_bs, cache_rs, B, width, pos, pos_dev=None):
    _check(lib().v2s_kv_append(src.data_ptr(), src_bs, cache.data_ptr(), cache_bs, cache_rs, B, width, pos, ptr(pos_dev), stream_ptr()),
           "v2s_kv_append")


def argmax_step_seq(logits, ld, rows, V, next_tok, unfinished, eos_id, pad_id, seq_out, seq_ld, pos_dev):
    _check(lib().v2s_argmax_step_seq(logits.data_ptr(), ld, rows, V, next_tok.data_ptr(), unfinished.data_ptr(), eos_id, pad_id,
                                     seq_out.data_ptr(), seq_ld, pos_dev.data_ptr(), stream_ptr()), "v2s_argmax_step_seq")


def argmax_step_tail(logits, ld, rows, V, next_tok, unfinished, eos_id, pad_id, seq_out, seq_ld, pos_dev, table, h_out, d, vocab, ticket):
    """argmax_step_seq + the next step's embedding lookup + the step counter's increment as one launch (greedy decode tail)"""
    _need(table, torch.bfloat16, "embed table"); _need(ticket, torch.int32, "ticket")
    _check(lib().v2s_argmax_step_tail(logits.data_ptr(), ld, rows, V, next_tok.data_ptr(), unfinished.data_ptr(), eos_id, pad_id,
                                      seq_out.data_ptr(), seq_ld, pos_dev.data_ptr(), table.data_ptr(), h_out.data_ptr(), d, vocab,
                                      ticket.data_ptr(), stream_ptr()), "v2s_argmax_step_tail")


def counter_add(ctr, delta):
    _check(lib().v2s_counter_add(ctr.data_ptr(), delta, stream_ptr()), "v2s_counter_add")


def topk_logprob(logits, ld, rows, V, K, beam_scores, out_val, out_idx, ban_token=-1, pos_dev=None, min_length=0, row_lse=None):
    _check(lib().v2s_topk_logprob(logits.data_ptr(), ld, rows, V, K, ptr(beam_scores), out_val.data_ptr(), out_idx.data_ptr(),
                                  ban_token, ptr(pos_dev), min_length, ptr(row_lse), stream_ptr()), "v2s_topk_logprob")


class BeamState:
    """Device-resident hypothesis bookkeeping of a beam search (v2s_beam_advance): the state BeamSearchScorer keeps on the host."""

    def __init__(self, B, nb, max_length, device, length_penalty=1.0):
        i32 = dict(dtype=torch.int32, device=device)
        self.B, self.nb, self.max_length = B, nb, max_length
        # n ** length_penalty as Python computes it (BeamHypotheses.add: sum_logprobs / (hyp.shape[-1] ** length_penalty)): scores bit-identical
        self.len_pow = torch.tensor([float(n) ** float(length_penalty) for n in range(max_length + 1)], dtype=torch.float64).to(device)
        self.hyp_tok = torch.zeros(B, nb, max_length, **i32)
        self.hyp_len = torch.zeros(B, nb, **i32); self.hyp_order = torch.zeros(B, nb, **i32)
        self.hyp_score = torch.zeros(B, nb, dtype=torch.float64, device=device)
        self.heap_n = torch.zeros(B, **i32); self.heap_added = torch.zeros(B, **i32)
        self.heap_worst = torch.full((B,), 1e9, dtype=torch.float64, device=device)
        self.done = torch.zeros(B, **i32); self.ndone = torch.zeros(1, **i32)


def beam_advance(cand_val, cand_tok, K, st: BeamState, eos_id, pad_id, pos_dev, hist, row_map, next_tok, beam_scores, src_rows):
    _need(cand_val, torch.float32, "beam_advance cand_val"); _need(cand_tok, torch.int32, "beam_advance cand_tok")
    _need(hist, torch.int64, "beam_advance hist"); _need(next_tok, torch.int64, "beam_advance next_tok")
    _need(beam_scores, torch.float32, "beam_advance beam_scores"); _need(src_rows, torch.int32, "beam_advance src_rows")
    _check(lib().v2s_beam_advance(cand_val.data_ptr(), cand_tok.data_ptr(), K, st.B, st.nb, eos_id, pad_id, st.len_pow.data_ptr(), pos_dev.data_ptr(),
                                  hist.data_ptr(), hist.stride(0), st.max_length, ptr(row_map), row_map.stride(0) if row_map is not None else 0,
                                  next_tok.data_ptr(), beam_scores.data_ptr(), src_rows.data_ptr(), st.hyp_tok.data_ptr(), st.hyp_len.data_ptr(),
                                  st.hyp_score.data_ptr(), st.hyp_order.data_ptr(), st.heap_n.data_ptr(), st.heap_worst.data_ptr(),
                                  st.heap_added.data_ptr(), st.done.data_ptr(), st.ndone.data_ptr(), stream_ptr()), "v2s_beam_advance")


def repetition_penalty(scores, ld, rows, V, hist, penalty, pos_dev=None, n_static=0, row_lse=None):
    _need(hist, torch.int64, "repetition_penalty hist")
    _check(lib().v2s_repetition_penalty(scores.data_ptr(), ld, rows, V, hist.data_ptr(), hist.stride(0), ptr(pos_dev), n_static, penalty,
                                        ptr(row_lse), stream_ptr()), "v2s_repetition_penalty")


def ban_token(scores, ld, rows, V, token, pos_dev, min_length):
    _check(lib().v2s_ban_token(scores.data_ptr(), ld, rows, V, token, pos_dev.data_ptr(), min_length, stream_ptr()), "v2s_ban_token")


def kv_gather(src, dst, idx, bs, rs, B, length, width):
    _need(idx, torch.int32, "kv_gather idx")
    _check(lib().v2s_kv_gather(src.data_ptr(), dst.data_ptr(), idx.data_ptr(), bs, rs, B, length, width, stream_ptr()), "v2s_kv_gather")


def span_corrupt(ids, lens, noise, max_len, num_text_tokens, eos, den_in, den_out, out_lens):
    _need(ids, torch.int64, "span_corrupt ids"); _need(lens, torch.int32, "span_corrupt lens"); _need(noise, torch.uint8, "span_corrupt noise")
    _need(den_in, torch.int64, "span_corrupt den_in"); _need(den_out, torch.int64, "span_corrupt den_out")
    _check(lib().v2s_span_corrupt(ids.data_ptr(), ids.stride(0), lens.data_ptr(), noise.data_ptr(), noise.stride(0), ids.shape[0], max_len,
                                  num_text_tokens, eos, den_in.data_ptr(), den_in.stride(0), den_out.data_ptr(), den_out.stride(0),
                                  out_lens.data_ptr(), stream_ptr()), "v2s_span_corrupt")


def scale_cols(W, w, out, rows, cols):
    _need(W, torch.bfloat16, "scale_cols W"); _need(w, torch.float32, "scale_cols w")
    _check(lib().v2s_scale_cols(W.data_ptr(), w.data_ptr(), out.data_ptr(), rows, cols, stream_ptr()), "v2s_scale_cols")


def topp_sample_step(logits, ld, rows, V, top_p, temperature, seed, next_tok, unfinished, eos_id, pad_id, seq_out=None, seq_ld=0, pos_dev=None,
                     probs_out=None, min_length=0, top_k=0):
    _check(lib().v2s_topp_sample_step(logits.data_ptr(), ld, rows, V, top_p, temperature, seed & 0xFFFFFFFF, next_tok.data_ptr(),
                                      unfinished.data_ptr(), eos_id, pad_id, ptr(seq_out), seq_ld, ptr(pos_dev), ptr(probs_out), min_length,
                                      top_k, stream_ptr()),
           "v2s_topp_sample_step")


def beam_sample_cand(logits, ld, rows, V, K, beam_scores, top_p, temperature, top_k, seed, out_val, out_tok, out_key, ban_token=-1,
                     pos_dev=None, min_length=0, row_lse=None):
    _need(out_val, torch.float32, "beam_sample_cand out_val"); _need(out_tok, torch.int32, "beam_sample_cand out_tok")
    _need(out_key, torch.float32, "beam_sample_cand out_key")
    _check(lib().v2s_beam_sample_cand(logits.data_ptr(), ld, rows, V, K, ptr(beam_scores), top_p, temperature, top_k, seed & 0xFFFFFFFF,
                                      out_val.data_ptr(), out_tok.data_ptr(), out_key.data_ptr(), ban_token, ptr(pos_dev), min_length,
                                      ptr(row_lse), stream_ptr()),
           "v2s_beam_sample_cand")
