"""Per-kernel parity tests: every C-ABI entry point vs an fp32 torch statement of the same op, on the GPU.

Inputs are bf16-representable so the only differences are accumulation order and bf16 rounding of outputs.
Tolerances are written next to each check.  (The end-to-end parity against the CPU oracle / reference
goldens is in test_model_gpu.py.)
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from vidchapters_amd import lib as L  # noqa: E402

DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def relerr(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def cos(a, b):
    a, b = a.float().flatten(), b.float().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


@pytest.fixture(params=[1, 0], ids=["tr_read", "scalar_lds"])
def tr_mode(request):
    L.set_option("tr_read", request.param)
    yield request.param
    L.set_option("tr_read", 1)


# ----------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (200, 136, 72), (77, 200, 328), (3200, 768, 768), (1024, 32200, 768)])
def test_gemm_nt(M, N, K):
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
    Cb = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    L.gemm(A, B, Cb, M, N, K)
    ref = A.float() @ B.float().T
    e = relerr(Cb, ref)
    print(f"gemm_nt {M}x{N}x{K}: relerr {e:.2e}")
    assert e < 1e-2          # bf16 output rounding: 2^-8 relative on the largest element
    Cf = torch.empty(M, N, dtype=torch.float32, device=DEV)
    L.gemm(A, B, Cf, M, N, K, alpha=0.5)
    e = relerr(Cf, 0.5 * ref)
    print(f"  fp32 out: relerr {e:.2e}")
    assert e < 2e-5          # fp32 accumulate of exact bf16 products; only summation order differs


def test_gemm_ragged_vocab():
    """LM-head shapes with a vocabulary that is not a multiple of 8 (612 tiny, 32100 for num_bins=0)."""
    M, V, d = 72, 612, 64
    ldv = (V + 7) // 8 * 8
    H, E = rnd(M, d, seed=1), rnd(V, d, seed=2)
    logits = torch.full((M, ldv), float("nan"), dtype=torch.float32, device=DEV)
    L.gemm(H, E, logits, M, V, d, ldc=ldv)
    assert relerr(logits[:, :V], H.float() @ E.float().T) < 2e-5
    dl = torch.zeros(M, ldv, dtype=torch.bfloat16, device=DEV); dl[:, :V] = rnd(M, V, seed=3)
    dH = torch.empty(M, d, dtype=torch.float32, device=DEV)
    L.gemm(dl, E, dH, M, d, V, transB=True, lda=ldv)                       # dgrad, ragged K
    assert relerr(dH, dl[:, :V].float() @ E.float()) < 2e-5
    dE = torch.zeros(V, d, dtype=torch.float32, device=DEV)
    L.gemm(dl, H, dE, V, d, M, transA=True, transB=True, lda=ldv, accumulate=True)   # wgrad, ragged M
    assert relerr(dE, dl[:, :V].float().T @ H.float()) < 2e-5


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 72), (264, 768, 3072), (1000, 64, 200)])
def test_gemm_dgrad_wgrad(M, N, K, tr_mode):
    # dgrad: dx[M, N] = dy[M, K] @ W[K, N]   (W stored [K][N] -> transB)
    dy, W = rnd(M, K, seed=3), rnd(K, N, seed=4)
    dx = torch.empty(M, N, dtype=torch.float32, device=DEV)
    L.gemm(dy, W, dx, M, N, K, transB=True)
    e = relerr(dx, dy.float() @ W.float())
    print(f"dgrad {M}x{N}x{K} tr={tr_mode}: relerr {e:.2e}")
    assert e < 2e-5
    # wgrad: dW[M', N'] = dY[Kc, M']^T @ X[Kc, N']  (both stored [K][*] -> transA, transB), accumulate into fp32
    Mp, Np, Kc = (N // 8) * 8, (K // 8) * 8, M
    dY, X = rnd(Kc, Mp, seed=5), rnd(Kc, Np, seed=6)
    dW = torch.ones(Mp, Np, dtype=torch.float32, device=DEV)
    L.gemm(dY, X, dW, Mp, Np, Kc, transA=True, transB=True, accumulate=True)
    ref = 1.0 + dY.float().T @ X.float()
    e = relerr(dW, ref)
    print(f"wgrad {Mp}x{Np}x{Kc} tr={tr_mode}: relerr {e:.2e}")
    assert e < 2e-5


@pytest.mark.parametrize("M,N,K", [(8192, 2048, 256), (8200, 2056, 320), (16384, 512, 128), (16400, 520, 192)])
def test_gemm_big_tiles(M, N, K):
    """Shapes that dispatch to the 256x256 / 256x128 LDS-DMA kernel (K % 64 == 0, >= 240 tiles), incl. ragged edges."""
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
    bias = rnd(N, seed=9, dtype=torch.float32)
    C = torch.empty(M, N, dtype=torch.float32, device=DEV)
    L.gemm(A, B, C, M, N, K, bias=bias)
    assert relerr(C, A.float() @ B.float().T + bias) < 2e-5
    W = rnd(K, N, seed=4)                                        # dgrad: dy[M,K] @ W[K,N]
    L.gemm(A, W, C, M, N, K, transB=True)
    assert relerr(C, A.float() @ W.float()) < 2e-5
    L.set_option("gemm_big", 0)
    C0 = torch.empty_like(C)
    L.gemm(A, W, C0, M, N, K, transB=True)
    L.set_option("gemm_big", 1)
    assert relerr(C, C0) < 2e-6                                  # same products, different tile shape / summation grouping


def test_gemm_big_wgrad_splitk():
    Mp, Np, Kc = 1024, 512, 4096
    dY, X = rnd(Kc, Mp, seed=5, scale=0.1), rnd(Kc, Np, seed=6, scale=0.1)
    ws = torch.empty(32 * Mp * Np, dtype=torch.float32, device=DEV)
    dW = torch.ones(Mp, Np, dtype=torch.float32, device=DEV)
    L.gemm(dY, X, dW, Mp, Np, Kc, transA=True, transB=True, accumulate=True, alpha=0.5, workspace=ws)
    assert relerr(dW, 1.0 + 0.5 * (dY.float().T @ X.float())) < 2e-5


@pytest.mark.parametrize("M,N,K,ta,tb", [(512, 256, 96, False, False), (300, 200, 64, False, False), (256, 128, 32, False, False),
                                          (512, 384, 160, False, True), (1000, 520, 224, True, True), (2304, 768, 2048, True, True)])
def test_gemm_w4_kernel(M, N, K, ta, tb):
    """The 4-wave 256x128x32 three-stage kernel (dispatched by default only for the large weight gradients) forced on every
    variant, incl. ragged edges, a one-stage contraction, and an epilogue chain."""
    A = rnd(*((K, M) if ta else (M, K)), seed=31, scale=0.25)
    B = rnd(*((K, N) if tb else (N, K)), seed=32, scale=0.25)
    ref = (A.float().T if ta else A.float()) @ (B.float() if tb else B.float().T)
    L.set_option("gemm_big", 3)
    try:
        C = torch.zeros(M, N, dtype=torch.float32, device=DEV)
        L.gemm(A, B, C, M, N, K, transA=ta, transB=tb)
        assert relerr(C, ref) < 2e-5
        if not ta:
            bias = rnd(N, seed=33, dtype=torch.float32); res = rnd(M, N, seed=34)
            out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            L.gemm(A, B, out, M, N, K, transB=tb, bias=bias, act=L.ACT_RELU, residual=res)
            assert relerr(out, torch.relu(ref + bias) + res.float()) < 1e-2
        else:
            ws = torch.empty(16 * M * N, dtype=torch.float32, device=DEV)
            dW = torch.ones(M, N, dtype=torch.float32, device=DEV)
            L.gemm(A, B, dW, M, N, K, transA=True, transB=True, accumulate=True, alpha=0.5, workspace=ws)
            assert relerr(dW, 1.0 + 0.5 * ref) < 2e-5
    finally:
        L.set_option("gemm_big", 1)


@pytest.mark.parametrize("mode", [2, 3], ids=["p8_256", "p8_128"])
@pytest.mark.parametrize("M,N,K,ta,tb", [(512, 512, 128, False, False), (300, 200, 160, False, False), (1000, 520, 224, False, True),
                                          (2304, 768, 2048, True, True), (520, 1032, 192, True, True), (4096, 2304, 768, False, False),
                                          (3000, 776, 3072, False, True), (256, 128, 256, False, False), (777, 264, 352, False, False)])
def test_gemm_p8_kernel(M, N, K, ta, tb, mode):
    """The 8-phase ping-pong kernel (counted-vmcnt LDS-DMA ring, staggered halves, transposed accumulators) forced on every operand
    layout: 4..96 stages (the prologue, the steady state and every tail length of the counted waits), ragged edges in M and N, an
    epilogue chain, fp32 accumulate and split-K.  Repeated launches must be bit-identical (a race in the ring would not be)."""
    A = rnd(*((K, M) if ta else (M, K)), seed=41, scale=0.25)
    B = rnd(*((K, N) if tb else (N, K)), seed=42, scale=0.25)
    ref = (A.float().T if ta else A.float()) @ (B.float() if tb else B.float().T)
    L.set_option("gemm_p8", mode)
    try:
        C = torch.zeros(M, N, dtype=torch.float32, device=DEV)
        L.gemm(A, B, C, M, N, K, transA=ta, transB=tb)
        assert "gemm_p8_kernel" in L.lib().v2s_last_gemm_kernel().decode()
        e = relerr(C, ref)
        print(f"p8 mode {mode} {M}x{N}x{K} ta={ta} tb={tb}: relerr {e:.2e}")
        assert e < 2e-5
        for _ in range(5):
            C2 = torch.empty_like(C)
            L.gemm(A, B, C2, M, N, K, transA=ta, transB=tb)
            assert torch.equal(C, C2)
        if not ta:
            bias = rnd(N, seed=43, dtype=torch.float32); res = rnd(M, N, seed=44)
            out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            L.gemm(A, B, out, M, N, K, transB=tb, bias=bias, act=L.ACT_RELU, residual=res)
            assert relerr(out, torch.relu(ref + bias) + res.float()) < 1e-2
        else:
            ws = torch.empty(16 * M * N, dtype=torch.float32, device=DEV)
            dW = torch.ones(M, N, dtype=torch.float32, device=DEV)
            L.gemm(A, B, dW, M, N, K, transA=True, transB=True, accumulate=True, alpha=0.5, workspace=ws)
            assert relerr(dW, 1.0 + 0.5 * ref) < 2e-5
    finally:
        L.set_option("gemm_p8", 1)


@pytest.mark.parametrize("M,N,K,tb,epi", [(4096, 2304, 768, False, "plain"), (4096, 3072, 768, False, "relu_drop"), (3000, 776, 3072, True, "plain"),
                                           (70000, 512, 288, False, "drop"), (1024, 256, 320, True, "relu"), (66000, 768, 768, True, "plain"),
                                           (2048, 512, 4096, False, "relu_drop")])
def test_gemm_p8_deferred_epilogue_kernel(M, N, K, tb, epi):
    """The persistent 8-phase kernel whose tile output leaves the chip during the next tile's main loop (bf16 held registers -> LDS
    staging slabs -> 16-byte stores with the dropout mask applied on the way, stores counted in the same vmcnt stream as the DMAs).
    Must be BIT-identical to the synchronous kernels (same products, same single rounding): 1 .. 3 tiles per block, ragged M and N,
    both operand layouts, ReLU / dropout epilogues, repeated launches."""
    A = rnd(M, K, seed=51, scale=0.25)
    B = rnd(*((K, N) if tb else (N, K)), seed=52, scale=0.25)
    kw = dict(transB=tb)
    if "relu" in epi:
        kw.update(act=L.ACT_RELU)
    if "drop" in epi:
        kw.update(dropout_p=0.1, dropout_seed=77)
    outs = {}
    try:
        for mode in (0, 4):
            L.set_option("gemm_p8", mode)
            C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            L.gemm(A, B, C, M, N, K, **kw)
            kern = L.lib().v2s_last_gemm_kernel().decode()
            assert ("gemm_p8d_kernel" in kern) == (mode == 4), kern
            outs[mode] = C
            if mode == 4:
                for _ in range(4):
                    C2 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
                    L.gemm(A, B, C2, M, N, K, **kw)
                    assert torch.equal(C2.view(torch.int16), C.view(torch.int16))
    finally:
        L.set_option("gemm_p8", 1)
    assert torch.isfinite(outs[4].float()).all()
    assert torch.equal(outs[0].view(torch.int16), outs[4].view(torch.int16))
    if epi == "plain":
        ref = A.float() @ (B.float() if tb else B.float().T)
        assert relerr(outs[4], ref) < 1e-2


@pytest.mark.parametrize("split", [0, 1])
def test_gemm_splitk_policies_agree(split):
    Mp, Np, Kc = 768, 768, 8192
    dY, X = rnd(Kc, Mp, seed=5, scale=0.1), rnd(Kc, Np, seed=6, scale=0.1)
    ws = torch.empty(32 * Mp * Np, dtype=torch.float32, device=DEV)
    dW = torch.zeros(Mp, Np, dtype=torch.float32, device=DEV)
    L.set_option("gemm_split", split)
    try:
        L.gemm(dY, X, dW, Mp, Np, Kc, transA=True, transB=True, accumulate=True, workspace=ws)
    finally:
        L.set_option("gemm_split", 1)
    assert relerr(dW, dY.float().T @ X.float()) < 2e-5


@pytest.mark.parametrize("M,N,K", [(64, 768, 768), (64, 2304, 768), (37, 768, 3072), (1, 32200, 768), (64, 616, 128), (17, 768, 768),
                                   (64, 32128, 768), (5, 9000, 1024), (33, 8192, 512), (64, 32128, 640),
                                   (256, 768, 768), (200, 3072, 768), (512, 768, 3072)])
def test_gemm_skinny_decode_kernel(M, N, K):
    """M <= 64 weight-streaming kernels used by the cached decoder (incl. ragged M / N and the epilogues the decode step uses): the
    K-split kernel for the layer projections, the LDS-resident-activation kernel for the LM head (N >= 8192, K in {512, 768, 1024})."""
    A, B = rnd(M, K, seed=41, scale=0.5), rnd(N, K, seed=42, scale=0.1)
    ref = A.float() @ B.float().T
    ld = (N + 7) // 8 * 8
    C = torch.zeros(M, ld, dtype=torch.float32, device=DEV)
    L.gemm(A, B, C, M, N, K, ldc=ld, alpha=0.25, decode=True)
    if M > 64:        # without the decode flag a call with more than 64 rows stays on the K-sequential tiled kernels
        Ct = torch.zeros(M, ld, dtype=torch.float32, device=DEV)
        L.gemm(A, B, Ct, M, N, K, ldc=ld, alpha=0.25)
        assert b"skinny" not in L.lib().v2s_last_gemm_kernel()
        L.gemm(A, B, C, M, N, K, ldc=ld, alpha=0.25, decode=True)
    wide = N >= 8192 and K in (512, 768, 1024)
    assert L.lib().v2s_last_gemm_kernel() == (b"gemm_skinny_wide_kernel" if wide else b"gemm_skinny_kernel")
    assert relerr(C[:, :N], 0.25 * ref) < 2e-5
    if wide:      # the LM head runs with the final RMSNorm fused (rms_eps): rows scaled by rsqrt(mean(x^2) + eps)
        C2 = torch.zeros(M, ld, dtype=torch.float32, device=DEV)
        L.gemm(A, B, C2, M, N, K, ldc=ld, alpha=0.25, rms_eps=1e-6, decode=True)
        af = A.float()
        assert relerr(C2[:, :N], 0.25 * ref * torch.rsqrt((af * af).mean(-1, keepdim=True) + 1e-6)) < 2e-5
        L.set_option("gemm_skinny", 2)      # the K-split kernel on the same problem
        try:
            C3 = torch.zeros(M, ld, dtype=torch.float32, device=DEV)
            L.gemm(A, B, C3, M, N, K, ldc=ld, alpha=0.25, rms_eps=1e-6, decode=True)
            assert L.lib().v2s_last_gemm_kernel() == b"gemm_skinny_kernel"
        finally:
            L.set_option("gemm_skinny", 1)
        assert relerr(C2[:, :N], C3[:, :N]) < 2e-5
    if N % 8 == 0:
        res = rnd(M, N, seed=43)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        L.gemm(A, B, out, M, N, K, act=L.ACT_RELU, residual=res, decode=True)
        assert relerr(out, torch.relu(ref) + res.float()) < 1e-2
        L.set_option("gemm_skinny", 0)
        try:
            out2 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            L.gemm(A, B, out2, M, N, K, act=L.ACT_RELU, residual=res, decode=True)
        finally:
            L.set_option("gemm_skinny", 1)
        assert relerr(out, out2) < 1e-2


def test_gemm_fused_rmsnorm_prologue():
    """v2s_gemm rms_eps (decode path): RMSNorm (modeling_t5.py:263-277) + Linear as one GEMM over weights with the norm weight folded
    into their columns (v2s_scale_cols)."""
    M, N, K = 48, 512, 768
    x = rnd(M, K, seed=51, scale=2.0); W = rnd(N, K, seed=52, scale=0.1)
    w = rnd(K, seed=53, dtype=torch.float32).abs() + 0.5
    Wf = torch.empty_like(W)
    L.scale_cols(W, w, Wf, N, K)
    assert relerr(Wf, W.float() * w) < 1e-2
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    L.gemm(x, Wf, out, M, N, K, rms_eps=1e-6, act=L.ACT_RELU)
    xf = x.float()
    ref = torch.relu((xf * torch.rsqrt((xf * xf).mean(-1, keepdim=True) + 1e-6) * w) @ W.float().T)
    assert relerr(out, ref) < 1.5e-2
    with pytest.raises(RuntimeError, match="rms_eps"):
        big = rnd(640, K, seed=54)
        L.gemm(big, Wf, torch.empty(640, N, dtype=torch.bfloat16, device=DEV), 640, N, K, rms_eps=1e-6)


def test_gemm_splitk_workspace():
    """Weight-gradient shape (few output tiles, long contraction): split-K through a caller workspace."""
    Mp, Np, Kc = 768, 256, 8000
    dY, X = rnd(Kc, Mp, seed=5, scale=0.1), rnd(Kc, Np, seed=6, scale=0.1)
    ws = torch.empty(8 * Mp * Np, dtype=torch.float32, device=DEV)
    dW = torch.ones(Mp, Np, dtype=torch.float32, device=DEV)
    L.gemm(dY, X, dW, Mp, Np, Kc, transA=True, transB=True, accumulate=True, alpha=0.5, workspace=ws)
    ref = 1.0 + 0.5 * (dY.float().T @ X.float())
    assert relerr(dW, ref) < 2e-5


def test_gemm_epilogues():
    M, N, K = 264, 256, 128
    A, B = rnd(M, K, seed=7), rnd(N, K, seed=8, scale=0.1)
    bias = rnd(N, seed=9, dtype=torch.float32)
    res = rnd(M, N, seed=10)
    acc = A.float() @ B.float().T
    # bias + gelu (+ saved pre-activation) + residual
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    L.gemm(A, B, out, M, N, K, bias=bias, act=L.ACT_GELU, pre=pre, residual=res)
    ref_pre = acc + bias
    ref = torch.nn.functional.gelu(ref_pre) + res.float()
    assert relerr(pre, ref_pre) < 1e-2 and relerr(out, ref) < 1e-2
    # relu
    L.gemm(A, B, out, M, N, K, act=L.ACT_RELU)
    assert relerr(out, torch.relu(acc)) < 1e-2
    # dact relu / gelu against saved z
    z = rnd(M, N, seed=11)
    o32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    L.gemm(A, B, o32, M, N, K, dact=L.ACT_RELU, z=z)
    assert relerr(o32, acc * (z.float() > 0)) < 2e-5
    L.gemm(A, B, o32, M, N, K, dact=L.ACT_GELU, z=z)
    zf = z.float().requires_grad_(True)
    torch.nn.functional.gelu(zf).sum().backward()
    assert relerr(o32, acc * zf.grad) < 1e-4
    # dropout: keep-rate and scaling; same seed -> same mask; mask reproduced by v2s_dropout at equal indices
    p = 0.1
    L.gemm(A, B, o32, M, N, K, dropout_p=p, dropout_seed=1234)
    kept = o32 != 0
    rate = 1.0 - kept.float().mean().item()
    assert abs(rate - p) < 0.01, rate
    inv = 1.0 / (1.0 - round(p * 65536) / 65536)
    assert relerr(o32[kept], (acc * inv)[kept]) < 2e-5
    ones = torch.ones(M * N, dtype=torch.bfloat16, device=DEV)
    dm = torch.empty_like(ones)
    L.dropout(ones, dm, M * N, p, 1234)
    assert torch.equal(dm.view(M, N) != 0, kept | (acc == 0))


def test_colsum():
    X = rnd(1000, 136, seed=12)
    out = torch.zeros(136, dtype=torch.float32, device=DEV)
    L.colsum(X, 1000, 136, out, accumulate=False)
    assert relerr(out, X.float().sum(0)) < 1e-5


# ----------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("rows,cols", [(37, 64), (1000, 768), (515, 1024), (64, 48)])
def test_rmsnorm(rows, cols):
    x, w, dy, addg = rnd(rows, cols, seed=1, scale=3), 1 + 0.1 * rnd(cols, seed=2, dtype=torch.float32), rnd(rows, cols, seed=3), rnd(rows, cols, seed=4)
    y = torch.empty_like(x); rstd = torch.empty(rows, dtype=torch.float32, device=DEV)
    L.rmsnorm_fwd(x, w, y, rstd, rows, cols, 1e-6)
    xf = x.float().requires_grad_(True); wf = w.clone().requires_grad_(True)
    ref = wf * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6))
    assert relerr(y, ref) < 1e-2
    ref.backward(dy.float())
    dx = torch.empty_like(x); dw = torch.zeros(cols, dtype=torch.float32, device=DEV)
    L.rmsnorm_bwd(x, w, rstd, dy, dx, addg, dw, rows, cols)
    assert relerr(dx, xf.grad + addg.float()) < 1e-2
    assert relerr(dw, wf.grad) < 1e-4


@pytest.mark.parametrize("rows,cols", [(37, 64), (300, 768), (64, 48)])
def test_layernorm(rows, cols):
    x, dy = rnd(rows, cols, seed=1, scale=2), rnd(rows, cols, seed=3)
    w, b = 1 + 0.1 * rnd(cols, seed=2, dtype=torch.float32), 0.1 * rnd(cols, seed=5, dtype=torch.float32)
    y = torch.empty_like(x); mean = torch.empty(rows, dtype=torch.float32, device=DEV); rstd = torch.empty_like(mean)
    L.layernorm_fwd(x, w, b, y, mean, rstd, rows, cols, 1e-5)
    xf = x.float().requires_grad_(True); wf = w.clone().requires_grad_(True); bf = b.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xf, (cols,), wf, bf, 1e-5)
    assert relerr(y, ref) < 1e-2
    ref.backward(dy.float())
    dx = torch.empty_like(x); dw = torch.zeros(cols, dtype=torch.float32, device=DEV); db = torch.zeros_like(dw)
    L.layernorm_bwd(x, w, mean, rstd, dy, dx, None, dw, db, rows, cols)
    assert relerr(dx, xf.grad) < 1e-2
    assert relerr(dw, wf.grad) < 1e-4 and relerr(db, bf.grad) < 1e-4


# ----------------------------------------------------------------------------------------------- attention
def attn_ref(q, k, v, scale, bias, mask, causal, causal_off):
    """fp32 reference: q [B,Nq,H,64] etc.  bias [H,Nq,Nk] or None, mask [B,Nk] bool or None (additive finfo.min)."""
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) * scale
    fmin = torch.finfo(torch.float32).min
    Nq, Nk = q.shape[1], k.shape[1]
    if bias is not None:
        s = s + bias[None]
    keep = torch.ones(q.shape[0], 1, Nq, Nk, dtype=torch.bool, device=q.device)
    if mask is not None:
        keep = keep & mask[:, None, None, :]
    if causal:
        qi = torch.arange(Nq, device=q.device)[:, None]; ki = torch.arange(Nk, device=q.device)[None, :]
        keep = keep & (ki <= qi + causal_off)[None, None]
    s = s + (~keep).float() * fmin
    p = torch.softmax(s, -1)
    return torch.einsum("bhqk,bkhd->bqhd", p, v)


def bias_from_diag(diag, Nq, Nk):
    qi = torch.arange(Nq, device=diag.device)[:, None]; ki = torch.arange(Nk, device=diag.device)[None, :]
    return diag[:, (ki - qi) + Nq - 1]


@pytest.mark.parametrize("B,H,Nq,Nk,scale,use_bias,use_mask,causal", [
    (2, 2, 100, 100, 0.125, False, False, False),     # ViT shape
    (2, 3, 200, 200, 1.0, True, True, False),         # encoder self-attention (ragged tiles)
    (2, 2, 72, 72, 1.0, True, True, True),            # decoder self-attention
    (2, 2, 40, 300, 1.0, False, True, False),         # cross-attention
    (1, 1, 7, 9, 1.0, True, True, False),             # tiny ragged
    (3, 4, 256, 1100, 1.0, False, True, False),       # cfg-2 cross-attention shape
    (2, 12, 1000, 1000, 1.0, True, True, False),      # cfg-2 encoder shape
    (1, 2, 2000, 2000, 1.0, True, True, False),       # cfg-5 encoder length: the forward / dQ kernels take the two-copy bias window
    (1, 1, 2700, 2700, 1.0, True, False, False),      # long enough for the dK/dV kernel's two-copy window as well
])
def test_attention_fwd_bwd(B, H, Nq, Nk, scale, use_bias, use_mask, causal, tr_mode):
    W = H * 64
    qkv_q = rnd(B, Nq, 3 * W, seed=1, scale=0.5)           # fused [q|k|v] rows like the QKV GEMM writes them
    qkv_k = qkv_q if Nq == Nk else rnd(B, Nk, 3 * W, seed=2, scale=0.5)
    q = qkv_q[..., :W]; k = qkv_k[..., W:2 * W]; v = qkv_k[..., 2 * W:]
    mask = None
    if use_mask:
        lens = torch.tensor([max(1, Nk - 3 - 17 * i) for i in range(B)], device=DEV)
        mask = torch.arange(Nk, device=DEV)[None, :] < lens[:, None]
        if B > 1 and not causal:
            mask[1, :] = mask[1, :] & (torch.arange(Nk, device=DEV) != 0)      # a hole at key 0
    diag = rnd(H, Nq + Nk - 1, seed=3, dtype=torch.float32) if use_bias else None
    o = torch.empty(B, Nq, W, dtype=torch.bfloat16, device=DEV)
    ml = torch.empty(B, H, Nq, 2, dtype=torch.float32, device=DEV)
    mk = mask.to(torch.uint8).contiguous() if mask is not None else None
    a = L.attn_args(B, H, Nq, Nk, q, k, v, o, (Nq * 3 * W, 3 * W), (Nk * 3 * W, 3 * W), (Nk * 3 * W, 3 * W), (Nq * W, W),
                    ml=ml, scale=scale, bias_diag=diag, key_mask=mk, causal=causal)
    L.attn_fwd(a)
    qf = q.float().reshape(B, Nq, H, 64).requires_grad_(True)
    kf = k.float().reshape(B, Nk, H, 64).requires_grad_(True)
    vf = v.float().reshape(B, Nk, H, 64).requires_grad_(True)
    df = diag.clone().requires_grad_(True) if use_bias else None
    ref = attn_ref(qf, kf, vf, scale, bias_from_diag(df, Nq, Nk) if use_bias else None, mask, causal, 0)
    e = relerr(o.view(B, Nq, H, 64), ref)
    print(f"attn fwd B{B} H{H} {Nq}x{Nk} tr={tr_mode}: relerr {e:.2e}")
    assert e < 2e-2          # P and O are rounded to bf16 (2^-8) once each
    d_o = rnd(B, Nq, W, seed=5)
    ref.backward(d_o.float().view(B, Nq, H, 64))
    dqkv_q = torch.zeros(B, Nq, 3 * W, dtype=torch.bfloat16, device=DEV)
    dqkv_k = dqkv_q if Nq == Nk else torch.zeros(B, Nk, 3 * W, dtype=torch.bfloat16, device=DEV)
    delta = torch.empty(B, H, Nq, 4, dtype=torch.float32, device=DEV)
    ddiag = torch.zeros(H, Nq + Nk - 1, dtype=torch.float32, device=DEV) if use_bias else None
    L.attn_bwd(a, d_o, (Nq * W, W), delta, dqkv_q[..., :W], dqkv_k[..., W:2 * W], dqkv_k[..., 2 * W:],
               (Nq * 3 * W, 3 * W), (Nk * 3 * W, 3 * W), (Nk * 3 * W, 3 * W), dbias_diag=ddiag)
    dq, dk, dv = dqkv_q[..., :W], dqkv_k[..., W:2 * W], dqkv_k[..., 2 * W:]
    for name, got, want in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        c = cos(got, want.reshape(B, -1, W)); e = relerr(got, want.reshape(B, -1, W))
        print(f"  {name}: cos {c:.5f} relerr {e:.2e}")
        assert c > 0.999 and e < 4e-2
    if use_bias:
        c = cos(ddiag, df.grad); e = relerr(ddiag, df.grad)
        print(f"  dbias_diag: cos {c:.5f} relerr {e:.2e}")
        assert c > 0.999 and e < 3e-2


@pytest.mark.parametrize("Nq,Nk,use_bias", [(256, 1100, False), (1000, 1000, True)])
def test_attention_backward_full_chip_is_repeatable_and_right(Nq, Nk, use_bias):
    """cfg-2's full grids (32 sequences x 12 heads; ragged lengths whose tail key blocks are all masked, so that the dK / dV kernel's
    early exit runs in thousands of blocks next to thousands that go on): five backward passes are bit-identical, and the first
    sequences equal the fp32 statement.  The small cases above launch too few blocks to see a race between the waves of a block
    (round 6: a block-wide vote read from LDS words that a faster wave was already overwriting with its Q tile)."""
    B, H, W = 32, 12, 768
    qkv_q = rnd(B, Nq, 3 * W, seed=11, scale=0.5)
    qkv_k = qkv_q if Nq == Nk else rnd(B, Nk, 3 * W, seed=12, scale=0.5)
    q = qkv_q[..., :W]; k = qkv_k[..., W:2 * W]; v = qkv_k[..., 2 * W:]
    g = torch.Generator(device="cpu").manual_seed(5)
    lens = torch.randint(int(0.3 * Nk), Nk + 1, (B,), generator=g).to(DEV)
    lens[0] = Nk; lens[1] = Nk // 2 + 3
    mask = torch.arange(Nk, device=DEV)[None, :] < lens[:, None]
    mk = mask.to(torch.uint8).contiguous()
    diag = rnd(H, Nq + Nk - 1, seed=3, dtype=torch.float32) if use_bias else None
    o = torch.empty(B, Nq, W, dtype=torch.bfloat16, device=DEV)
    ml = torch.empty(B, H, Nq, 2, dtype=torch.float32, device=DEV)
    a = L.attn_args(B, H, Nq, Nk, q, k, v, o, (Nq * 3 * W, 3 * W), (Nk * 3 * W, 3 * W), (Nk * 3 * W, 3 * W), (Nq * W, W),
                    ml=ml, scale=1.0, bias_diag=diag, key_mask=mk, dropout_p=0.1, dropout_seed=77)
    L.attn_fwd(a)
    d_o = rnd(B, Nq, W, seed=5)
    delta = torch.empty(B, H, Nq, 4, dtype=torch.float32, device=DEV)
    runs = []
    for _ in range(5):
        dq_ = torch.full((B, Nq, 3 * W), float("nan"), dtype=torch.bfloat16, device=DEV)
        dk_ = dq_ if Nq == Nk else torch.full((B, Nk, 3 * W), float("nan"), dtype=torch.bfloat16, device=DEV)
        ddiag = torch.zeros(H, Nq + Nk - 1, dtype=torch.float32, device=DEV) if use_bias else None
        L.attn_bwd(a, d_o, (Nq * W, W), delta, dq_[..., :W], dk_[..., W:2 * W], dk_[..., 2 * W:],
                   (Nq * 3 * W, 3 * W), (Nk * 3 * W, 3 * W), (Nk * 3 * W, 3 * W), dbias_diag=ddiag)
        runs.append((dq_[..., :W].clone(), dk_[..., W:2 * W].clone(), dk_[..., 2 * W:].clone()))
    torch.cuda.synchronize()
    for name, j in (("dq", 0), ("dk", 1), ("dv", 2)):
        assert torch.isfinite(runs[0][j].float()).all(), f"{name}: not finite"
        for r in runs[1:]:
            assert torch.equal(r[j], runs[0][j]), f"{name}: two launches on the same operands differ"
    # masked keys of rows that saw a real key get exactly zero
    dk0, dv0 = runs[0][1], runs[0][2]
    dead = ~mask
    assert float(dk0[dead].float().abs().max()) == 0.0 and float(dv0[dead].float().abs().max()) == 0.0
    # without dropout, the first two sequences against fp32 torch
    a2 = L.attn_args(B, H, Nq, Nk, q, k, v, o, (Nq * 3 * W, 3 * W), (Nk * 3 * W, 3 * W), (Nk * 3 * W, 3 * W), (Nq * W, W),
                     ml=ml, scale=1.0, bias_diag=diag, key_mask=mk)
    L.attn_fwd(a2)
    dq_ = torch.zeros(B, Nq, 3 * W, dtype=torch.bfloat16, device=DEV)
    dk_ = dq_ if Nq == Nk else torch.zeros(B, Nk, 3 * W, dtype=torch.bfloat16, device=DEV)
    ddiag = torch.zeros(H, Nq + Nk - 1, dtype=torch.float32, device=DEV) if use_bias else None
    L.attn_bwd(a2, d_o, (Nq * W, W), delta, dq_[..., :W], dk_[..., W:2 * W], dk_[..., 2 * W:],
               (Nq * 3 * W, 3 * W), (Nk * 3 * W, 3 * W), (Nk * 3 * W, 3 * W), dbias_diag=ddiag)
    nb = 2
    qf = q[:nb].float().reshape(nb, Nq, H, 64).requires_grad_(True)
    kf = k[:nb].float().reshape(nb, Nk, H, 64).requires_grad_(True)
    vf = v[:nb].float().reshape(nb, Nk, H, 64).requires_grad_(True)
    ref = attn_ref(qf, kf, vf, 1.0, bias_from_diag(diag, Nq, Nk) if use_bias else None, mask[:nb], False, 0)
    ref.backward(d_o[:nb].float().view(nb, Nq, H, 64))
    for name, got, want in (("dq", dq_[:nb, :, :W], qf.grad), ("dk", dk_[:nb, :, W:2 * W], kf.grad), ("dv", dk_[:nb, :, 2 * W:], vf.grad)):
        c = cos(got, want.reshape(nb, -1, W)); e = relerr(got, want.reshape(nb, -1, W))
        print(f"  {name}: cos {c:.5f} relerr {e:.2e}")
        assert c > 0.999 and e < 4e-2


def test_full_size_launches_are_repeatable():
    """The step's big single-pass launches at their cfg-2 sizes (every CU busy for several rounds), three times each into NaN-filled outputs:
    bit-identical, finite.  None of them reduces across blocks, so any difference is a race inside a block or between a block and its
    successor on the same CU (round 6: such a race in the dK / dV attention kernel passed every small-grid parity test)."""
    M = 32000
    x768, x3072 = rnd(M, 768, seed=1, scale=0.5), rnd(M, 3072, seed=2, scale=0.5)
    w_qkv, w_wi, w_wo = rnd(2304, 768, seed=3, scale=0.04), rnd(3072, 768, seed=4, scale=0.04), rnd(768, 3072, seed=5, scale=0.02)
    res = rnd(M, 768, seed=6)
    u = rnd(M, 3072, seed=7)                              # ReLU-mask operand of the masked dgrad
    cases = {
        "QKV forward (plain)": lambda c: L.gemm(x768, w_qkv, c(M, 2304), M, 2304, 768),
        "wi forward (ReLU + dropout)": lambda c: L.gemm(x768, w_wi, c(M, 3072), M, 3072, 768, act=L.ACT_RELU, dropout_p=0.1, dropout_seed=11),
        "wo forward (residual + dropout)": lambda c: L.gemm(x3072, w_wo, c(M, 768), M, 768, 3072, residual=res, dropout_p=0.1, dropout_seed=12),
        "wi dgrad (plain, NN)": lambda c: L.gemm(x3072, w_wi, c(M, 768), M, 768, 3072, transB=True, ldb=768),
        "wo dgrad (ReLU mask + dropout, NN)": lambda c: L.gemm(x768, w_wo, c(M, 3072), M, 3072, 768, transB=True, ldb=3072, dact=L.ACT_RELU, z=u,
                                                                dropout_p=0.1, dropout_seed=13),
    }
    for name, launch in cases.items():
        outs = []
        for _ in range(3):
            box = []
            def c(m, n):
                box.append(torch.full((m, n), float("nan"), dtype=torch.bfloat16, device=DEV)); return box[0]
            launch(c)
            outs.append(box[0])
        torch.cuda.synchronize()
        assert torch.isfinite(outs[0].float()).all(), name
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), name
    # attention forward at the encoder's grid (bias, ragged masks, dropout): output and row statistics
    B, H, N, W = 32, 12, 1000, 768
    qkv = rnd(B, N, 3 * W, seed=21, scale=0.5)
    g = torch.Generator(device="cpu").manual_seed(9)
    lens = torch.randint(int(0.7 * N), N + 1, (B,), generator=g).to(DEV)
    mk = (torch.arange(N, device=DEV)[None, :] < lens[:, None]).to(torch.uint8).contiguous()
    diag = rnd(H, 2 * N - 1, seed=22, dtype=torch.float32)
    outs = []
    for _ in range(3):
        o = torch.full((B, N, W), float("nan"), dtype=torch.bfloat16, device=DEV)
        ml = torch.full((B, H, N, 2), float("nan"), dtype=torch.float32, device=DEV)
        L.attn_fwd(L.attn_args(B, H, N, N, qkv[..., :W], qkv[..., W:2 * W], qkv[..., 2 * W:], o, (N * 3 * W, 3 * W), (N * 3 * W, 3 * W),
                               (N * 3 * W, 3 * W), (N * W, W), ml=ml, bias_diag=diag, key_mask=mk, dropout_p=0.1, dropout_seed=5))
        outs.append((o, ml))
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0][0].float()).all() and torch.isfinite(outs[0][1]).all()
    for o, ml in outs[1:]:
        assert torch.equal(o, outs[0][0]) and torch.equal(ml, outs[0][1])


@pytest.mark.parametrize("M,N,K,G,acc", [(768, 768, 512, 12, False), (136, 264, 192, 3, True), (2304, 768, 256, 16, False)])
def test_gemm_grouped_weight_gradients(M, N, K, G, acc):
    """v2s_gemm_grouped: G weight gradients of one shape (dY^T X, fp32 out) in one launch == the single launches / fp32 torch."""
    As = [rnd(K, M, seed=10 + i, scale=0.5) for i in range(G)]
    Bs = [rnd(K, N, seed=40 + i, scale=0.5) for i in range(G)]
    init = [rnd(M, N, seed=70 + i, dtype=torch.float32) for i in range(G)]
    Cs = [t.clone() for t in init]
    L.gemm_grouped(As, Bs, Cs, M, N, K, accumulate=acc)
    assert L.lib().v2s_last_gemm_kernel().decode() == "gemm_dma_grouped_kernel"
    for a, b, c, c0 in zip(As, Bs, Cs, init):
        ref = a.float().t() @ b.float() + (c0 if acc else 0)
        assert relerr(c, ref) < 2e-5
        single = c0.clone()
        L.gemm(a, b, single, M, N, K, transA=True, transB=True, accumulate=acc)
        assert torch.equal(single, c)          # same kernel body, same K order (no split-K without a workspace)
    with pytest.raises(ValueError):
        L.gemm_grouped(As[:1] * 17, Bs[:1] * 17, Cs[:1] * 17, M, N, K)


@pytest.mark.parametrize("kind,M,N,K,ep", [
    ("NT", 512, 512, 384, ""), ("NN", 768, 512, 512, ""), ("NT", 2048, 768, 768, ""), ("NN", 1024, 1536, 640, ""),      # persistent form (plain, whole tiles)
    ("NT", 1000, 520, 384, ""), ("NT", 2000, 768, 768, "res"), ("NT", 1300, 1544, 384, "act"), ("NT", 600, 520, 256, "f32"),
    ("NN", 777, 392, 256, "dact"), ("NT", 768, 768, 2048, "bias"),
    ("NN", 1024, 768, 640, "dact"), ("NN", 1000, 1544, 512, "dactdrop"),                                                # persistent form with the ReLU-mask epilogue
    ("NT", 1024, 768, 640, "act"), ("NT", 1300, 1544, 640, "actdrop"), ("NT", 2048, 3072, 768, "actdrop")])             # ... with the ReLU / ReLU + dropout forward epilogue
def test_gemm_a4_kernel(kind, M, N, K, ep):
    """gemm_a4_kernel / gemm_a4p_kernel (generated asm K loop, 4 waves, 128 x 128 wave tiles in AGPRs, v_mfma 32x32x16): against fp32 torch on
    the bf16 inputs, plain and fused epilogues, ragged edges, both weight layouts; where the persistent deferred-write-out form is legal it
    must agree BIT FOR BIT with the one-tile form (same K order, one rounding).  CPU counterpart on the generated text: test_gemm_a4_emu.py."""
    A = rnd(M, K, seed=M + 1, scale=0.5)
    B = rnd(*((K, N) if kind == "NN" else (N, K)), seed=N + 2, scale=0.5)
    kw = dict(transB=(kind == "NN"), ldb=N if kind == "NN" else K)
    ref = A.float() @ (B.float() if kind == "NN" else B.float().t())
    if ep == "res":
        r = rnd(M, N, seed=7); kw.update(residual=r); ref = ref + r.float()
    if ep == "act":
        kw.update(act=L.ACT_RELU); ref = torch.relu(ref)
    if ep == "bias":
        b = rnd(N, seed=9, dtype=torch.float32); kw.update(bias=b, act=L.ACT_GELU); ref = torch.nn.functional.gelu(ref + b)
    if ep == "dact":
        z = torch.relu(rnd(M, N, seed=9)); kw.update(dact=L.ACT_RELU, z=z); ref = ref * (z.float() > 0)
    if ep == "dactdrop":          # z = the forward's post-dropout activation: the mask is z > 0, the scale 1 / (1 - p) (p16-rounded like the library's)
        z = torch.relu(rnd(M, N, seed=9)) * (rnd(M, N, seed=10) > -1.0); kw.update(dact=L.ACT_RELU, z=z, dropout_p=0.1, dropout_seed=3)
        ref = ref * (z.float() > 0) * (1.0 / (1.0 - round(0.1 * 65536) / 65536.0))
    if ep == "actdrop":           # the mask is the library's counter-based generator: the reference is the default dispatch with the same seed
        kw.update(act=L.ACT_RELU, dropout_p=0.1, dropout_seed=5)
    outs = {}
    try:
        for mode in (2, 3) + ((0,) if ep == "actdrop" else ()):
            L.set_option("gemm_a4", mode)
            C_ = torch.full((M, N), float("nan"), dtype=torch.float32 if ep == "f32" else torch.bfloat16, device=DEV)
            L.gemm(A, B, C_, M, N, K, **kw)
            outs[mode] = (C_, L.lib().v2s_last_gemm_kernel().decode())
    finally:
        L.set_option("gemm_a4", 1)
    # the persistent form: plain bf16 epilogue (ragged edges: the last tile row / column overlaps), the ReLU-mask dgrad epilogue, or the forward's
    # ReLU (+ dropout: the mask regenerated inside the kernel)
    whole = (ep == "" and N >= 512 and K >= 384) or (ep in ("dact", "dactdrop") and kind == "NN" and N >= 512 and K >= 512) or \
        (ep == "act" and kind == "NT" and N >= 512 and K >= 384) or (ep == "actdrop" and kind == "NT" and N >= 512 and K >= 640)
    assert ("gemm_a4p_kernel" in outs[2][1]) == whole and "gemm_a4_kernel" in outs[3][1], (outs[2][1], outs[3][1])
    if ep == "actdrop":
        assert "gemm_a4" not in outs[0][1]
        keep = outs[0][0] != 0
        scale = 1.0 / (1.0 - round(0.1 * 65536) / 65536.0)
        assert 0.40 < float(keep.float().mean()) < 0.50                                    # half the pre-activations are negative, 10 % of the rest dropped
        assert torch.equal(outs[2][0], outs[0][0]) and torch.equal(outs[3][0], outs[0][0])  # same mask, same single rounding
        assert relerr(torch.where(keep, outs[2][0].float(), torch.zeros_like(ref)), torch.where(keep, torch.relu(ref) * scale, torch.zeros_like(ref))) < 5e-3
        return
    for mode in (2, 3):
        assert relerr(outs[mode][0], ref) < (2e-5 if ep == "f32" else 5e-3), (mode, kind, M, N, K, ep)      # bf16 half-ulp of the largest element: up to 2^-8
    assert torch.equal(outs[2][0], outs[3][0])


@pytest.mark.parametrize("M,N,K,split", [(768, 512, 1024, False), (520, 776, 384, False), (768, 768, 4096, True), (1536, 768, 2304, True)])
def test_gemm_a4_weight_gradient(M, N, K, split):
    """the TN form of gemm_a4_kernel (dY^T X: both operands [k][rows], tr-read fragments), fp32 accumulate into C, with and without the
    split-K workspace (slices of whole 128-wide iterations + the deterministic reduce)"""
    A = rnd(K, M, seed=M + 3, scale=0.5)
    B = rnd(K, N, seed=N + 4, scale=0.5)
    C0 = rnd(M, N, seed=5, dtype=torch.float32)
    ref = A.float().t() @ B.float() + C0
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV) if split else None
    try:
        L.set_option("gemm_a4", 2)                      # (default dispatch keeps weight gradients on the two-blocks-per-CU kernels: they share CUs with the main chain)
        C_ = C0.clone()
        L.gemm(A, B, C_, M, N, K, transA=True, transB=True, accumulate=True, workspace=ws)
        kern = L.lib().v2s_last_gemm_kernel().decode()
    finally:
        L.set_option("gemm_a4", 1)
    assert kern == "gemm_a4_kernel<true, true>", kern
    assert relerr(C_, ref) < 2e-5


def test_sum_n():
    """v2s_sum_n: fp32 sum of bf16 parts, one rounding (the decoder layers' d(memory) contributions)"""
    parts = rnd(5, 1000, 72, seed=3)
    y = torch.empty(1000, 72, dtype=torch.bfloat16, device=DEV)
    L.sum_n(parts, 1000 * 72, 5, y, 1000 * 72)
    assert torch.equal(y, parts.float().sum(0).to(torch.bfloat16))


def test_fp32_io_debug_mode_norm_ce_attention():
    """SURVEY 8c "tolerances to state": with fp32 activations in and out (library option fp32_io) the norm, cross-entropy and
    attention entry points must agree with fp32 torch to <= 1e-4 -- the debug mode that separates a kernel bug from bf16 rounding.
    (Norms / CE: the product kernels instantiated for fp32 I/O; attention: fp32-arithmetic reference kernels with the product
    kernels' semantics, incl. the dropout mask function, compared here with torch and with the MFMA kernels.)"""
    F = torch.nn.functional
    g = torch.Generator(device="cpu").manual_seed(11)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    worst = {}
    with L.fp32_io():
        # ---- RMSNorm / LayerNorm forward + backward (with the fused residual-gradient add)
        rows, cols, eps = 333, 768, 1e-6
        for ln in (False, True):
            x, w, b, dy, da = rn(rows, cols), rn(cols) * 0.2 + 1.0, rn(cols) * 0.1, rn(rows, cols), rn(rows, cols)
            y = torch.empty_like(x); rstd = torch.empty(rows, device=DEV); mean = torch.empty(rows, device=DEV)
            dx = torch.empty_like(x); dw = torch.zeros(cols, device=DEV); db = torch.zeros(cols, device=DEV)
            xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            if ln:
                L.layernorm_fwd(x, w, b, y, mean, rstd, rows, cols, 1e-5)
                L.layernorm_bwd(x, w, mean, rstd, dy, dx, da, dw, db, rows, cols)
                ref = F.layer_norm(xr, (cols,), wr, br, 1e-5)
            else:
                L.rmsnorm_fwd(x, w, y, rstd, rows, cols, eps)
                L.rmsnorm_bwd(x, w, rstd, dy, dx, da, dw, rows, cols)
                ref = wr * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + eps))
            ref.backward(dy)
            tag = "layernorm" if ln else "rmsnorm"
            worst[tag] = max(relerr(y, ref), relerr(dx, xr.grad + da), relerr(dw, wr.grad), relerr(db, br.grad) if ln else 0.0)
        # ---- cross-entropy (label smoothing, ignore_index) forward + backward with fp32 d(logits)
        R_, V, ld, sm = 37, 1000, 1000, 0.1
        logits = rn(R_, ld, sc=3.0)
        labels = torch.randint(0, V, (R_,), generator=g).to(DEV); labels[::5] = -100
        row = torch.empty(R_, 2, device=DEV); ls = torch.zeros(1, device=DEV); cnt = torch.zeros(1, device=DEV)
        L.ce_fwd(logits, ld, labels, R_, V, sm, row, ls, cnt)
        lr = logits.clone().requires_grad_(True)
        want = F.cross_entropy(lr, labels, ignore_index=-100, label_smoothing=sm)
        want.backward()
        gs = (1.0 / cnt).contiguous()
        dl = torch.empty(R_, ld, device=DEV)
        L.ce_bwd(logits, ld, labels, row, R_, V, sm, gs, dl, ld)
        worst["ce"] = max(abs(float(ls / cnt) - float(want)) / abs(float(want)), relerr(dl, lr.grad))
        # ---- attention: bias + key mask (with a fully masked row) / causal; torch fp32 reference
        for (B, H, Nq, Nk, scale, use_bias, use_mask, causal, drop) in ((2, 3, 70, 90, 1.0, True, True, False, 0.0), (2, 2, 65, 65, 1.0, True, False, True, 0.0),
                                                                     (1, 2, 50, 50, 0.125, False, False, False, 0.0)):
            W = H * 64
            q, k, v, d_o = rn(B, Nq, W, sc=0.5), rn(B, Nk, W, sc=0.5), rn(B, Nk, W), rn(B, Nq, W)
            mask = None
            if use_mask:
                mask = torch.arange(Nk, device=DEV)[None, :] < torch.tensor([Nk - 5, 0], device=DEV)[:, None]      # batch entry 1: NO visible key
            diag = rn(H, Nq + Nk - 1) if use_bias else None
            o = torch.empty(B, Nq, W, device=DEV); ml = torch.empty(B, H, Nq, 2, device=DEV)
            a = L.attn_args(B, H, Nq, Nk, q, k, v, o, (Nq * W, W), (Nk * W, W), (Nk * W, W), (Nq * W, W), ml=ml, scale=scale, bias_diag=diag,
                            key_mask=mask.to(torch.uint8).contiguous() if mask is not None else None, causal=causal, dropout_p=drop, dropout_seed=5)
            L.attn_fwd(a)
            qf, kf, vf = (t.view(B, -1, H, 64).clone().requires_grad_(True) for t in (q, k, v))
            df = diag.clone().requires_grad_(True) if use_bias else None
            ref = attn_ref(qf, kf, vf, scale, bias_from_diag(df, Nq, Nk) if use_bias else None, mask, causal, 0)
            ref.backward(d_o.view(B, Nq, H, 64))
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            delta = torch.empty(B, H, Nq, 4, device=DEV)
            ddiag = torch.zeros(H, Nq + Nk - 1, device=DEV) if use_bias else None
            L.attn_bwd(a, d_o, (Nq * W, W), delta, dq, dk, dv, (Nq * W, W), (Nk * W, W), (Nk * W, W), dbias_diag=ddiag)
            e = max(relerr(o.view(B, Nq, H, 64), ref), relerr(dq.view(B, Nq, H, 64), qf.grad), relerr(dk.view(B, Nk, H, 64), kf.grad),
                    relerr(dv.view(B, Nk, H, 64), vf.grad), relerr(ddiag, df.grad) if use_bias else 0.0)
            worst[f"attention {Nq}x{Nk}" + (" causal" if causal else "")] = e
    print("fp32_io debug mode, worst relative error vs fp32 torch:", {k: f"{v:.1e}" for k, v in worst.items()})
    assert all(v <= 1e-4 for v in worst.values()), worst
    # the MFMA kernels against the fp32 reference kernels on the same inputs, with attention dropout: same mask function -> same kept set
    B, H, N, W = 2, 2, 130, 128
    qkv = rnd(B, N, 3 * W, seed=1, scale=0.5); d_o = rnd(B, N, W, seed=2)
    diag = rnd(H, 2 * N - 1, seed=3, dtype=torch.float32)
    st3, st1 = (N * 3 * W, 3 * W), (N * W, W)
    o16 = torch.empty(B, N, W, dtype=torch.bfloat16, device=DEV); ml = torch.empty(B, H, N, 2, device=DEV)
    a16 = L.attn_args(B, H, N, N, qkv, qkv[..., W:], qkv[..., 2 * W:], o16, st3, st3, st3, st1, ml=ml, bias_diag=diag, dropout_p=0.1, dropout_seed=9)
    L.attn_fwd(a16)
    dqkv16 = torch.zeros_like(qkv); delta = torch.empty(B, H, N, 4, device=DEV)
    L.attn_bwd(a16, d_o, st1, delta, dqkv16, dqkv16[..., W:], dqkv16[..., 2 * W:], st3, st3, st3)
    with L.fp32_io():
        q32 = qkv.float(); o32 = torch.empty(B, N, W, device=DEV); ml2 = torch.empty(B, H, N, 2, device=DEV)
        a32 = L.attn_args(B, H, N, N, q32, q32[..., W:], q32[..., 2 * W:], o32, st3, st3, st3, st1, ml=ml2, bias_diag=diag, dropout_p=0.1, dropout_seed=9)
        L.attn_fwd(a32)
        dqkv32 = torch.zeros(B, N, 3 * W, device=DEV)
        L.attn_bwd(a32, d_o.float(), st1, delta, dqkv32, dqkv32[..., W:], dqkv32[..., 2 * W:], st3, st3, st3)
    e_o, c_g = relerr(o16, o32), cos(dqkv16, dqkv32)
    print(f"  MFMA bf16 kernels vs fp32 reference kernels with dropout 0.1: O relerr {e_o:.2e}, d(qkv) cosine {c_g:.5f}")
    assert e_o < 2e-2 and c_g > 0.999


@pytest.mark.parametrize("N,H,drop", [(200, 3, 0.0), (1000, 4, 0.1), (70, 2, 0.1)])
def test_attention_packed_equals_padded(N, H, drop):
    """Packed ("varlen", seq_off) self-attention == the padded + key-masked call on every row that exists: forward output,
    dQ/dK/dV and the bias gradient, bit for bit (same tiles, same arithmetic; the pad keys the padded call masks do not exist)."""
    B = 4
    W = H * 64
    lens = [N, max(1, N - 37), max(1, N // 2 + 3), 1]
    off = [0]
    for n in lens:
        off.append(off[-1] + n)
    T = off[-1]
    qkv_pad = rnd(B, N, 3 * W, seed=1, scale=0.5)
    do_pad = rnd(B, N, W, seed=2)
    for b_, n_ in enumerate(lens):
        do_pad[b_, n_:] = 0          # the gradient reaching pad rows is exactly zero in the model (their keys are masked downstream)
    diag = rnd(H, 2 * N - 1, seed=3, dtype=torch.float32)
    mask = (torch.arange(N, device=DEV)[None, :] < torch.tensor(lens, device=DEV)[:, None]).to(torch.uint8).contiguous()
    # padded reference call
    o_pad = torch.zeros(B, N, W, dtype=torch.bfloat16, device=DEV); ml = torch.zeros(B, H, N, 2, device=DEV)
    st3, st1 = (N * 3 * W, 3 * W), (N * W, W)
    a = L.attn_args(B, H, N, N, qkv_pad, qkv_pad[..., W:], qkv_pad[..., 2 * W:], o_pad, st3, st3, st3, st1, ml=ml, bias_diag=diag,
                    key_mask=mask, dropout_p=drop, dropout_seed=9)
    L.attn_fwd(a)
    dqkv_pad = torch.zeros_like(qkv_pad); delta = torch.zeros(B, H, N, 4, device=DEV); dd_pad = torch.zeros_like(diag)
    L.attn_bwd(a, do_pad, st1, delta, dqkv_pad, dqkv_pad[..., W:], dqkv_pad[..., 2 * W:], st3, st3, st3, dbias_diag=dd_pad, far=(-60, 60))
    # packed call on the rows that exist
    rows = torch.cat([torch.arange(n, device=DEV) + b * N for b, n in enumerate(lens)])
    qkv = qkv_pad.view(B * N, 3 * W)[rows].contiguous(); d_o = do_pad.view(B * N, W)[rows].contiguous()
    o = torch.zeros(T, W, dtype=torch.bfloat16, device=DEV); ml2 = torch.zeros(B, H, N, 2, device=DEV)
    so = torch.tensor(off, dtype=torch.int32, device=DEV)
    a2 = L.attn_args(B, H, N, N, qkv, qkv[:, W:], qkv[:, 2 * W:], o, (0, 3 * W), (0, 3 * W), (0, 3 * W), (0, W), ml=ml2, bias_diag=diag,
                     dropout_p=drop, dropout_seed=9, seq_off=so)
    L.attn_fwd(a2)
    dqkv = torch.zeros_like(qkv); delta2 = torch.zeros(B, H, N, 4, device=DEV); dd = torch.zeros_like(diag)
    L.attn_bwd(a2, d_o, (0, W), delta2, dqkv, dqkv[:, W:], dqkv[:, 2 * W:], (0, 3 * W), (0, 3 * W), (0, 3 * W), dbias_diag=dd, far=(-60, 60))
    assert torch.equal(o, o_pad.view(B * N, W)[rows])
    assert torch.equal(dqkv, dqkv_pad.view(B * N, 3 * W)[rows])
    assert relerr(dd, dd_pad) < 1e-5


@pytest.mark.parametrize("drop", [0.0, 0.1])
def test_cross_attention_query_packed_equals_padded(drop):
    """seq_q_only: the query side packed (a padding-free decoder), K / V dense with the memory's key mask == the padded call on every
    query row that exists (output, dQ) and on every key (dK, dV: pad query rows carry a zero upstream gradient in the model), bit for bit."""
    B, H, Nq, Nk = 3, 4, 200, 330
    W = H * 64
    lens = [Nq, 133, 1]
    off = [0]
    for n in lens:
        off.append(off[-1] + n)
    T = off[-1]
    q_pad = rnd(B, Nq, W, seed=1, scale=0.5); kv = rnd(B, Nk, 2 * W, seed=2, scale=0.5)
    do_pad = rnd(B, Nq, W, seed=3)
    for b_, n_ in enumerate(lens):
        do_pad[b_, n_:] = 0
    kmask = (torch.arange(Nk, device=DEV)[None, :] < torch.tensor([Nk, 250, 101], device=DEV)[:, None]).to(torch.uint8).contiguous()
    sq, skv = (Nq * W, W), (Nk * 2 * W, 2 * W)
    o_pad = torch.zeros(B, Nq, W, dtype=torch.bfloat16, device=DEV); ml = torch.zeros(B, H, Nq, 2, device=DEV)
    a = L.attn_args(B, H, Nq, Nk, q_pad, kv, kv[..., W:], o_pad, sq, skv, skv, sq, ml=ml, key_mask=kmask, dropout_p=drop, dropout_seed=4)
    L.attn_fwd(a)
    dq_pad = torch.zeros_like(q_pad); dkv_pad = torch.zeros_like(kv); delta = torch.zeros(B, H, Nq, 4, device=DEV)
    L.attn_bwd(a, do_pad, sq, delta, dq_pad, dkv_pad, dkv_pad[..., W:], sq, skv, skv)
    rows = torch.cat([torch.arange(n, device=DEV) + b * Nq for b, n in enumerate(lens)])
    q = q_pad.view(B * Nq, W)[rows].contiguous(); d_o = do_pad.view(B * Nq, W)[rows].contiguous()
    o = torch.zeros(T, W, dtype=torch.bfloat16, device=DEV); ml2 = torch.zeros(B, H, Nq, 2, device=DEV)
    so = torch.tensor(off, dtype=torch.int32, device=DEV)
    a2 = L.attn_args(B, H, Nq, Nk, q, kv, kv[..., W:], o, (0, W), skv, skv, (0, W), ml=ml2, key_mask=kmask, dropout_p=drop, dropout_seed=4,
                     seq_off=so, seq_q_only=True)
    L.attn_fwd(a2)
    dq = torch.zeros_like(q); dkv = torch.zeros_like(kv); delta2 = torch.zeros(B, H, Nq, 4, device=DEV)
    L.attn_bwd(a2, d_o, (0, W), delta2, dq, dkv, dkv[..., W:], (0, W), skv, skv)
    assert torch.equal(o, o_pad.view(B * Nq, W)[rows])
    assert torch.equal(dq, dq_pad.view(B * Nq, W)[rows])
    assert torch.equal(dkv, dkv_pad)
    # key side packed as well (kv_seq_off: the valid memory rows only, no key mask), query side packed or dense
    klens = [Nk, 250, 101]
    koff = [0]
    for n in klens:
        koff.append(koff[-1] + n)
    krows = torch.cat([torch.arange(n, device=DEV) + b * Nk for b, n in enumerate(klens)])
    kvp = kv.view(B * Nk, 2 * W)[krows].contiguous()
    ko = torch.tensor(koff, dtype=torch.int32, device=DEV)
    for q_packed in (True, False):
        o3 = torch.zeros_like(o if q_packed else o_pad); ml3 = torch.zeros(B, H, Nq, 2, device=DEV)
        qq, dd_o = (q, d_o) if q_packed else (q_pad, do_pad)
        qst = (0, W) if q_packed else sq
        a3 = L.attn_args(B, H, Nq, Nk, qq, kvp, kvp[:, W:], o3, qst, (0, 2 * W), (0, 2 * W), qst, ml=ml3, dropout_p=drop, dropout_seed=4,
                         seq_off=so if q_packed else None, seq_q_only=q_packed, kv_seq_off=ko)
        L.attn_fwd(a3)
        dq3 = torch.zeros_like(qq); dkv3 = torch.zeros_like(kvp); delta3 = torch.zeros(B, H, Nq, 4, device=DEV)
        L.attn_bwd(a3, dd_o, qst, delta3, dq3, dkv3, dkv3[:, W:], qst, (0, 2 * W), (0, 2 * W))
        if q_packed:
            assert torch.equal(o3, o) and torch.equal(dq3, dq)
        else:
            for b_, n_ in enumerate(lens):          # rows that exist; the padded call's pad query rows carry garbage-free but unused values
                assert torch.equal(o3[b_, :n_], o_pad[b_, :n_]) and torch.equal(dq3[b_, :n_], dq_pad[b_, :n_])
        assert torch.equal(dkv3, dkv_pad.view(B * Nk, 2 * W)[krows])
        assert dkv_pad.view(B * Nk, 2 * W)[krows].abs().sum() > 0


def test_attention_fully_masked_row_is_uniform():
    B, H, N = 1, 1, 70
    W = 64
    q, k, v = rnd(B, N, W, seed=1), rnd(B, N, W, seed=2), rnd(B, N, W, seed=3)
    mk = torch.zeros(B, N, dtype=torch.uint8, device=DEV)          # every key masked -> reference gives uniform weights
    o = torch.empty(B, N, W, dtype=torch.bfloat16, device=DEV)
    a = L.attn_args(B, H, N, N, q, k, v, o, (N * W, W), (N * W, W), (N * W, W), (N * W, W), key_mask=mk)
    L.attn_fwd(a)
    want = v.float().mean(1, keepdim=True).expand(B, N, W)
    assert relerr(o, want) < 2e-2
    # backward of the same degenerate rows: P = 1/N everywhere (1/l is folded into the exponent inside the kernels; all-masked rows
    # take the masked-element value -log2 l instead)
    ml = torch.empty(B, H, N, 2, dtype=torch.float32, device=DEV)
    a = L.attn_args(B, H, N, N, q, k, v, o, (N * W, W), (N * W, W), (N * W, W), (N * W, W), key_mask=mk, ml=ml)
    L.attn_fwd(a)
    d_o = rnd(B, N, W, seed=4)
    dq, dk, dv = (torch.zeros(B, N, W, dtype=torch.bfloat16, device=DEV) for _ in range(3))
    delta = torch.empty(B, H, N, 4, dtype=torch.float32, device=DEV)
    L.attn_bwd(a, d_o, (N * W, W), delta, dq, dk, dv, (N * W, W), (N * W, W), (N * W, W))
    P = torch.full((B, N, N), 1.0 / N, device=DEV)
    dP = d_o.float() @ v.float().transpose(1, 2)
    dS = P * (dP - (dP * P).sum(-1, keepdim=True))
    assert relerr(dv, P.transpose(1, 2) @ d_o.float()) < 2e-2
    assert relerr(dq, dS @ k.float()) < 3e-2 and relerr(dk, dS.transpose(1, 2) @ q.float()) < 3e-2


@pytest.mark.parametrize("use_bias", [False, True])
def test_attention_masks_with_holes_tails_and_an_invisible_sequence(use_bias):
    """Key masks that are not prefixes: whole 64-key tiles masked in the middle of a sequence, a visible key after them, a long masked
    tail (the tiles the forward / dQ loops cut off: tiles_to_visit), a sequence whose only visible key is its last one, and a sequence
    without any visible key (uniform over every key, finfo.min arithmetic of modeling_t5.py:559) -- forward and backward against fp32."""
    B, H, Nq, Nk, W = 5, 2, 200, 520, 128
    q, k, v = rnd(B, Nq, W, seed=1, scale=0.5), rnd(B, Nk, W, seed=2, scale=0.5), rnd(B, Nk, W, seed=3)
    mask = torch.zeros(B, Nk, dtype=torch.bool, device=DEV)
    mask[0, :70] = True                                    # seven trailing tiles masked
    mask[1, :10] = True; mask[1, 300:331] = True           # tiles 1-3 masked, visible keys again in tile 4-5, masked tail
    mask[2, Nk - 1] = True                                 # the only visible key is the very last one
    mask[3, :] = True                                      # nothing masked
    # sequence 4: no visible key
    diag = rnd(H, Nq + Nk - 1, seed=3, dtype=torch.float32) if use_bias else None
    o = torch.empty(B, Nq, W, dtype=torch.bfloat16, device=DEV)
    ml = torch.empty(B, H, Nq, 2, dtype=torch.float32, device=DEV)
    st = ((Nq * W, W), (Nk * W, W), (Nk * W, W), (Nq * W, W))
    a = L.attn_args(B, H, Nq, Nk, q, k, v, o, *st, ml=ml, bias_diag=diag, key_mask=mask.to(torch.uint8).contiguous())
    L.attn_fwd(a)
    qf = q.float().reshape(B, Nq, H, 64).requires_grad_(True)
    kf = k.float().reshape(B, Nk, H, 64).requires_grad_(True)
    vf = v.float().reshape(B, Nk, H, 64).requires_grad_(True)
    df = diag.clone().requires_grad_(True) if use_bias else None
    ref = attn_ref(qf, kf, vf, 1.0, bias_from_diag(df, Nq, Nk) if use_bias else None, mask, False, 0)
    for b in range(B):
        e = relerr(o.view(B, Nq, H, 64)[b], ref[b])
        print(f"  sequence {b}: forward relerr {e:.2e}")
        assert e < 2e-2
    d_o = rnd(B, Nq, W, seed=5)
    ref.backward(d_o.float().view(B, Nq, H, 64))
    dq, dk, dv = (torch.full((B, n, W), float("nan"), dtype=torch.bfloat16, device=DEV) for n in (Nq, Nk, Nk))
    delta = torch.empty(B, H, Nq, 4, dtype=torch.float32, device=DEV)
    ddiag = torch.zeros(H, Nq + Nk - 1, dtype=torch.float32, device=DEV) if use_bias else None
    L.attn_bwd(a, d_o, (Nq * W, W), delta, dq, dk, dv, (Nq * W, W), (Nk * W, W), (Nk * W, W), dbias_diag=ddiag)
    for name, got, want in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        for b in range(B):
            wb = want[b].reshape(-1, W)
            e = float((got[b].float() - wb).abs().max() / (want.abs().max() + 1e-12))
            c = cos(got[b], wb) if float(wb.abs().max()) > 0 else 1.0
            print(f"  {name}[{b}]: cos {c:.5f} err/global max {e:.2e}")
            assert c > 0.999 and e < 4e-2
    if use_bias:
        c = cos(ddiag, df.grad); e = relerr(ddiag, df.grad)
        print(f"  dbias_diag: cos {c:.5f} relerr {e:.2e}")
        assert c > 0.999 and e < 3e-2


@pytest.mark.parametrize("Nq,Nk,use_bias,causal", [(1000, 1000, True, False), (256, 1100, False, False), (256, 256, True, True)])
def test_attention_backward_skips_zero_gradient_rows_exactly(Nq, Nk, use_bias, causal):
    """Query rows whose dO is all zero (+0 or -0: the pad rows of a dense batch) add nothing to any gradient; the dQ kernel marks them, leaves
    early when its 128 rows are all marked, and the dK / dV kernel ends its query loop at the last unmarked row.  The same launch with the zeros
    replaced by +-2^-100 takes the full path: every output agrees to 1e-25 absolute (the only difference left is the 1e-30 the stand-ins add), with
    dropout, masks and the bias gradient on.  Plus a sequence whose dO is zero everywhere, and a zero block in the middle of a sequence."""
    B, H, W = 4, 3, 192
    q = rnd(B, Nq, W, seed=1, scale=0.5); k = rnd(B, Nk, W, seed=2, scale=0.5); v = rnd(B, Nk, W, seed=3)
    lens = torch.tensor([Nk, int(0.71 * Nk), int(0.5 * Nk) + 1, Nk - 5], device=DEV)
    mk = (torch.arange(Nk, device=DEV)[None, :] < lens[:, None]).to(torch.uint8).contiguous() if not causal else None
    diag = rnd(H, Nq + Nk - 1, seed=3, dtype=torch.float32) if use_bias else None
    o = torch.empty(B, Nq, W, dtype=torch.bfloat16, device=DEV)
    ml = torch.empty(B, H, Nq, 2, dtype=torch.float32, device=DEV)
    st = ((Nq * W, W), (Nk * W, W), (Nk * W, W), (Nq * W, W))
    a = L.attn_args(B, H, Nq, Nk, q, k, v, o, *st, ml=ml, bias_diag=diag, key_mask=mk, causal=causal, dropout_p=0.1, dropout_seed=31)
    L.attn_fwd(a)
    d_o = rnd(B, Nq, W, seed=5)
    zero = torch.zeros(B, Nq, dtype=torch.bool, device=DEV)
    zero[0, int(0.87 * Nq):] = True                       # a pad tail
    zero[1, int(0.4 * Nq):] = True                        # a long one: whole 128-row blocks of the dQ kernel
    zero[1, 10:12] = True
    zero[2, :] = True                                     # nothing to do for the whole sequence
    zero[3, 128:256] = True                               # a zero block in the middle, non-zero rows after it
    zero[3, Nq - 1] = True
    zero[3, 32:64] = True                                 # one wave's 32 rows inside a block that has work
    sign = torch.where(torch.arange(W, device=DEV) % 3 == 0, -1.0, 1.0).to(torch.bfloat16)
    d_zero = torch.where(zero[..., None], (0.0 * sign)[None, None, :].expand(B, Nq, W), d_o).contiguous()      # +0 and -0
    d_tiny = torch.where(zero[..., None], (2.0 ** -100 * sign)[None, None, :].expand(B, Nq, W), d_o).contiguous()
    assert bool((d_zero[2] == 0).all()) and bool((d_zero.view(torch.int16)[2] != 0).any())                     # -0 bit patterns are in
    res = []
    for d in (d_zero, d_tiny):
        dq, dk, dv = (torch.full((B, n, W), float("nan"), dtype=torch.bfloat16, device=DEV) for n in (Nq, Nk, Nk))
        delta = torch.empty(B, H, Nq, 4, dtype=torch.float32, device=DEV)
        ddiag = torch.zeros(H, Nq + Nk - 1, dtype=torch.float32, device=DEV) if use_bias else None
        L.attn_bwd(a, d, (Nq * W, W), delta, dq, dk, dv, (Nq * W, W), (Nk * W, W), (Nk * W, W), dbias_diag=ddiag)
        res.append((dq, dk, dv, ddiag))
    for name, x, y in zip(("dq", "dk", "dv", "dbias"), res[0], res[1]):
        if x is None:
            continue
        assert torch.isfinite(x.float()).all() and torch.isfinite(y.float()).all(), name
        e = float((x.float() - y.float()).abs().max())
        print(f"  {name}: max |skip - full| = {e:.1e}")
        # (the bias gradient is summed over blocks with fp32 atomics: the order of arrival moves its last bits from launch to launch)
        assert e < (1e-6 * float(y.abs().max()) if name == "dbias" else 1e-25), (name, e)
    assert float(res[0][0][zero].float().abs().max()) == 0.0          # dQ of a zero row is zero
    assert float(res[0][1][2].float().abs().max()) == 0.0 and float(res[0][2][2].float().abs().max()) == 0.0


def test_attention_dropout_consistency():
    B, H, Nq, Nk, W = 2, 2, 128, 192, 128
    q, k, v = rnd(B, Nq, W, seed=1, scale=0.3), rnd(B, Nk, W, seed=2, scale=0.3), rnd(B, Nk, W, seed=3)
    o0 = torch.empty(B, Nq, W, dtype=torch.bfloat16, device=DEV); o1 = torch.empty_like(o0); o2 = torch.empty_like(o0)
    ml = torch.empty(B, H, Nq, 2, dtype=torch.float32, device=DEV)
    st = ((Nq * W, W), (Nk * W, W), (Nk * W, W), (Nq * W, W))
    L.attn_fwd(L.attn_args(B, H, Nq, Nk, q, k, v, o0, *st, ml=ml))
    a = L.attn_args(B, H, Nq, Nk, q, k, v, o1, *st, ml=ml, dropout_p=0.1, dropout_seed=77)
    L.attn_fwd(a)
    L.attn_fwd(L.attn_args(B, H, Nq, Nk, q, k, v, o2, *st, ml=ml, dropout_p=0.1, dropout_seed=77))
    assert torch.equal(o1, o2) and not torch.equal(o0, o1)
    # O is linear in V for a fixed mask: <dO, O(V=D)> == <dV, D>  ties the forward mask to the dK/dV kernel's mask
    d_o = rnd(B, Nq, W, seed=5)
    dq, dk, dv = (torch.zeros(B, n, W, dtype=torch.bfloat16, device=DEV) for n in (Nq, Nk, Nk))
    delta = torch.empty(B, H, Nq, 4, dtype=torch.float32, device=DEV)
    L.attn_bwd(a, d_o, (Nq * W, W), delta, dq, dk, dv, (Nq * W, W), (Nk * W, W), (Nk * W, W))
    D = rnd(B, Nk, W, seed=6)
    oD = torch.empty_like(o0)
    L.attn_fwd(L.attn_args(B, H, Nq, Nk, q, k, D, oD, *st, dropout_p=0.1, dropout_seed=77))
    lhs = (d_o.float() * oD.float()).sum().item(); rhs = (dv.float() * D.float()).sum().item()
    print(f"dropout linearity: {lhs:.4f} vs {rhs:.4f}")
    assert abs(lhs - rhs) < 2e-2 * (abs(lhs) + abs(rhs) + 1.0)


def test_attention_dropout_backward_with_extracted_mask():
    """All three kernels must use ONE dropout mask: the forward kernel's mask is read back exactly (V = one-hot rows, one 64-key
    chunk at a time, so O = dropped probabilities), then dQ / dK / dV are compared with the closed-form backward under that mask."""
    B, H, Nq, Nk, W, pdrop, scale = 2, 1, 128, 192, 64, 0.25, 0.5
    q, k, v = rnd(B, Nq, W, seed=1, scale=0.5), rnd(B, Nk, W, seed=2, scale=0.5), rnd(B, Nk, W, seed=3)
    st = ((Nq * W, W), (Nk * W, W), (Nk * W, W), (Nq * W, W))
    ml = torch.empty(B, H, Nq, 2, dtype=torch.float32, device=DEV)
    pd = torch.zeros(B, Nq, Nk, device=DEV)
    for c in range(Nk // 64):
        onehot = torch.zeros(B, Nk, W, dtype=torch.bfloat16, device=DEV)
        onehot[:, c * 64:(c + 1) * 64] = torch.eye(64, dtype=torch.bfloat16, device=DEV)
        oc = torch.empty(B, Nq, W, dtype=torch.bfloat16, device=DEV)
        L.attn_fwd(L.attn_args(B, H, Nq, Nk, q, k, onehot, oc, *st, scale=scale, dropout_p=pdrop, dropout_seed=123))
        pd[:, :, c * 64:(c + 1) * 64] = oc.float()
    keep = (pd > 0).float()
    assert abs(keep.mean().item() - (1 - pdrop)) < 0.02
    P = torch.softmax(scale * q.float() @ k.float().transpose(1, 2), -1)
    assert relerr(pd, P * keep / (1 - pdrop)) < 1e-2
    o = torch.empty(B, Nq, W, dtype=torch.bfloat16, device=DEV)
    a = L.attn_args(B, H, Nq, Nk, q, k, v, o, *st, ml=ml, scale=scale, dropout_p=pdrop, dropout_seed=123)
    L.attn_fwd(a)
    d_o = rnd(B, Nq, W, seed=5)
    dq, dk, dv = (torch.zeros(B, n, W, dtype=torch.bfloat16, device=DEV) for n in (Nq, Nk, Nk))
    delta = torch.empty(B, H, Nq, 4, dtype=torch.float32, device=DEV)
    L.attn_bwd(a, d_o, (Nq * W, W), delta, dq, dk, dv, (Nq * W, W), (Nk * W, W), (Nk * W, W))
    Pd = P * keep / (1 - pdrop)
    dP = (d_o.float() @ v.float().transpose(1, 2)) * keep / (1 - pdrop)
    dS = P * (dP - (dP * P).sum(-1, keepdim=True))
    for name, got, want in (("dv", dv, Pd.transpose(1, 2) @ d_o.float()), ("dq", dq, scale * dS @ k.float()),
                            ("dk", dk, scale * dS.transpose(1, 2) @ q.float())):
        assert relerr(got, want) < 2e-2, (name, relerr(got, want))


def test_bias_diag_roundtrip():
    H, n, nb = 12, 1999, 32
    table = rnd(nb, H, seed=1, dtype=torch.float32)
    lut = (torch.arange(n, device=DEV) % nb).to(torch.int32)
    diag = torch.empty(H, n, dtype=torch.float32, device=DEV)
    L.bias_diag_fwd(table, lut, diag, H, n, nb)
    assert torch.equal(diag, table[lut.long()].T.contiguous())
    dd = rnd(H, n, seed=2, dtype=torch.float32)
    dt = torch.zeros(nb, H, dtype=torch.float32, device=DEV)
    L.bias_bucket_bwd(dd, lut, dt, H, n, nb)
    ref = torch.zeros(nb, H, device=DEV).index_add_(0, lut.long(), dd.T.contiguous())
    assert relerr(dt, ref) < 1e-5


# ----------------------------------------------------------------------------------------------- misc
def test_embedding_and_elementwise():
    V, d, n = 612, 64, 300
    table = rnd(V, d, seed=1)
    ids = torch.randint(0, V, (n,), device=DEV)
    out = torch.empty(n, d, dtype=torch.bfloat16, device=DEV)
    L.embed_fwd(ids, table, out, n, d, V)
    assert torch.equal(out, table[ids])
    dy = rnd(n, d, seed=2)
    dt = torch.zeros(V, d, dtype=torch.float32, device=DEV)
    L.embed_bwd(ids, dy, dt, n, d, V)
    ref = torch.zeros(V, d, device=DEV).index_add_(0, ids, dy.float())
    assert relerr(dt, ref) < 1e-5
    x, pos = rnd(6, 10, 64, seed=3), rnd(10, 64, seed=4)
    y = torch.empty_like(x)
    L.add_bcast(x, pos, y, x.numel(), pos.numel())
    assert relerr(y, x.float() + pos.float()) < 1e-2
    g = torch.zeros(10, 64, dtype=torch.float32, device=DEV)
    L.bcast_grad(x, g, x.numel(), pos.numel())
    assert relerr(g, x.float().sum(0)) < 1e-5


@pytest.mark.parametrize("rows,V", [(24, 612), (512, 32200)])
def test_cross_entropy(rows, V):
    logits = rnd(rows, V, seed=1, scale=3, dtype=torch.float32)
    labels = torch.randint(0, V, (rows,), device=DEV)
    labels[::5] = -100
    row = torch.empty(rows, 2, dtype=torch.float32, device=DEV)
    ls = torch.zeros(1, dtype=torch.float32, device=DEV); cnt = torch.zeros(1, dtype=torch.float32, device=DEV)
    L.ce_fwd(logits, V, labels, rows, V, 0.1, row, ls, cnt)
    lf = logits.clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lf, labels, ignore_index=-100, label_smoothing=0.1)
    got = (ls / cnt).item()
    print(f"CE {rows}x{V}: {got:.6f} vs {ref.item():.6f}")
    assert abs(got - ref.item()) < 2e-5 * abs(ref.item())      # fp32 both sides
    ref.backward()
    gs = (1.0 / cnt).contiguous()
    ldd = (V + 7) // 8 * 8
    dl = torch.full((rows, ldd), float("nan"), dtype=torch.bfloat16, device=DEV)
    L.ce_bwd(logits, V, labels, row, rows, V, 0.1, gs, dl, ldd)
    assert relerr(dl[:, :V], lf.grad) < 1e-2 and cos(dl[:, :V], lf.grad) > 0.9999
    assert (dl[:, V:] == 0).all()


@pytest.mark.parametrize("rows,V,d", [(300, 612, 128), (2048 + 40, 32200, 768), (130, 1000, 64)])
def test_lm_head_cross_entropy_without_logits(rows, V, d):
    """v2s_lmhead_ce_fwd / _bwd (round 6): tied LM head + label-smoothed CE with the logits reduced / recomputed inside the GEMM epilogue, against
    fp32 torch on the same bf16 operands (modeling_t5.py:1709-1721); ragged rows, a vocabulary that is not a multiple of the tile, ignored rows,
    bf16 d(logits) of a row CHUNK (as the engine calls it) incl. its zero pad columns."""
    Vpad = (V + 63) // 64 * 64
    h = rnd(rows, d, seed=1, scale=1.0)
    E = torch.zeros(Vpad, d, dtype=torch.bfloat16, device=DEV)
    E[:V] = rnd(V, d, seed=2, scale=1.5)
    alpha = d ** -0.5
    labels = torch.randint(0, V, (rows,), device=DEV)
    labels[::7] = -100
    labels[1] = V - 1                                                     # a target in the last, partial column tile
    part = torch.empty(L.lmhead_ce_workspace_floats(rows, Vpad), dtype=torch.float32, device=DEV)
    row = torch.full((rows, 2), float("nan"), dtype=torch.float32, device=DEV)
    acc = torch.zeros(2, dtype=torch.float32, device=DEV)
    L.lmhead_ce_fwd(h, d, E, rows, V, Vpad, d, alpha, labels, 0.1, part, row, acc[0:1], acc[1:2])
    hf = h.float().requires_grad_(True)
    lg = (hf * alpha) @ E[:V].float().T
    ref = torch.nn.functional.cross_entropy(lg, labels, ignore_index=-100, label_smoothing=0.1)
    got = (acc[0] / acc[1]).item()
    keep = labels >= 0
    lse_err = float((row[keep, 0] - torch.logsumexp(lg.detach(), -1)[keep]).abs().max())
    print(f"LM head + CE {rows} x {V} x {d}: loss {got:.6f} vs {ref.item():.6f}; count {acc[1].item():.0f}; log-sum-exp max error {lse_err:.2e}")
    assert acc[1].item() == float(keep.sum()) and abs(got - ref.item()) < 2e-5 * abs(ref.item()) and lse_err < 2e-4
    assert (row[~keep] == 0).all()
    ref.backward(retain_graph=True)
    dlg = torch.autograd.grad(ref, lg, retain_graph=True)[0]
    gs = (1.0 / acc[1:2]).contiguous()
    r0, n = 3, rows - 5                                                   # a chunk that starts and ends off the tile grid
    dl = torch.full((n, Vpad), float("nan"), dtype=torch.bfloat16, device=DEV)
    L.lmhead_ce_bwd(h[r0:r0 + n], d, E, n, V, Vpad, d, alpha, labels[r0:r0 + n], row[r0:r0 + n], 0.1, gs, dl, Vpad)
    assert relerr(dl[:, :V], dlg[r0:r0 + n]) < 1e-2 and cos(dl[:, :V], dlg[r0:r0 + n]) > 0.9999
    assert (dl[:, V:] == 0).all() and (dl[~keep[r0:r0 + n]] == 0).all()
    # the split-K reduction that rounds to bf16 itself (d(hidden) = d(logits) E alpha without the fp32 chunk + cast)
    ws = torch.empty(16 * n * d, dtype=torch.float32, device=DEV)
    dh = torch.full((n, d), float("nan"), dtype=torch.bfloat16, device=DEV)
    L.gemm(dl, E, dh, n, d, Vpad, transB=True, lda=Vpad, ldb=d, alpha=alpha, workspace=ws)
    want = (dl.float() @ E.float()) * alpha
    assert relerr(dh, want) < 1e-2 and cos(dh, want) > 0.99999


def test_optimizer_kernels():
    n = 100003 * 4
    p = rnd(n, seed=1, dtype=torch.float32); g = rnd(n, seed=2, dtype=torch.float32, scale=0.01)
    m = torch.zeros_like(p); v = torch.zeros_like(p); pb = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    ws = torch.empty(1024, dtype=torch.float32, device=DEV); sq = torch.zeros(1, dtype=torch.float32, device=DEV)
    L.sqnorm(g, n, ws, sq)
    assert abs(sq.item() - (g.double() ** 2).sum().item()) < 1e-5 * sq.item()
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=3e-4)
    for step in (1, 2):
        pr.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([pr], 0.1)
        opt.step()
        sq.zero_(); L.sqnorm(g, n, ws, sq)
        L.adam_step(p, m, v, g, pb, n, 3e-4, 0.9, 0.999, 1e-8, 0.0, step, gnorm_sq=sq, max_norm=0.1)
    assert (p - pr.detach()).abs().max().item() < 1e-6        # <= 2 ulp of fp32 at |p| ~ 4
    assert torch.equal(pb, p.to(torch.bfloat16))
    # time-token renorm (dvc.py:118-126)
    V, d, nb = 612, 64, 100
    emb = rnd(V, d, seed=3, dtype=torch.float32); embb = torch.empty(V, d, dtype=torch.bfloat16, device=DEV)
    ref = emb.clone()
    ref[-nb:] /= (ref[-nb:].norm(dim=1).mean() / ref[:-nb].norm(dim=1).mean())
    wsr = torch.empty(V + 2, dtype=torch.float32, device=DEV)
    L.cast_bf16(emb, embb, V * d)
    L.timetoken_renorm(emb, embb, V, d, nb, wsr)
    assert relerr(emb, ref) < 1e-6 and torch.equal(embb, emb.to(torch.bfloat16))


@pytest.mark.parametrize("K", [2, 8, 16, 40, 66])
def test_topk_logprob_and_kv_gather(K):
    rows, V, ld = 12, 3211, 3216
    g = torch.Generator().manual_seed(K)
    logits = torch.zeros(rows, ld)
    logits[:, :V] = torch.randn(rows, V, generator=g) * 3
    logits[:, V:] = 1e9                                            # pad columns must never be read
    bs = torch.randn(rows, generator=g)
    val = torch.zeros(rows, K, device=DEV); idx = torch.zeros(rows, K, dtype=torch.int32, device=DEV)
    L.topk_logprob(logits.to(DEV), ld, rows, V, K, bs.to(DEV), val, idx)
    ref = torch.log_softmax(logits[:, :V].double(), -1) + bs[:, None].double()
    rv, ri = torch.topk(ref, K, dim=1)
    assert torch.equal(idx.cpu().long(), ri)
    assert (val.cpu().double() - rv).abs().max() < 1e-5
    # min_length: the banned token stays in the softmax but is no candidate while pos + 1 < min_length
    ban = int(ri[0, 0]); pos = torch.tensor([2], dtype=torch.int32, device=DEV)
    ref2 = ref.clone(); ref2[:, ban] = -float("inf")
    rv2, ri2 = torch.topk(ref2, K, dim=1)
    L.topk_logprob(logits.to(DEV), ld, rows, V, K, bs.to(DEV), val, idx, ban_token=ban, pos_dev=pos, min_length=4)
    assert torch.equal(idx.cpu().long(), ri2) and (val.cpu().double() - rv2).abs().max() < 1e-5
    L.topk_logprob(logits.to(DEV), ld, rows, V, K, bs.to(DEV), val, idx, ban_token=ban, pos_dev=pos, min_length=3)
    assert torch.equal(idx.cpu().long(), ri)
    if K <= 8 or K > 32:
        # (K > 32: the generic rounds kernel, same tie order)  the 1024-thread threshold path (aligned rows of <= 32768 logits: candidates = elements >= the K-th largest thread maximum),
        # with rows that overflow its 64-slot list (flat row, a wide plateau of equal maxima) falling back to the sorted-list path;
        # equal values: lower token first
        rows2, V2, ld2 = 9, 32200, 32256
        lg2 = torch.zeros(rows2, ld2)
        lg2[:, :V2] = torch.randn(rows2, V2, generator=g) * 3
        lg2[:, V2:] = 1e9
        lg2[5, :V2] = 0.25                                           # flat: every element ties
        lg2[6, 100:400] = 50.0                                       # 300 equal maxima
        lg2[7, 7] = lg2[7, 31000] = 60.0                             # a tie between far-apart columns
        bs2 = torch.randn(rows2, generator=g)
        val2 = torch.zeros(rows2, K, device=DEV); idx2 = torch.zeros(rows2, K, dtype=torch.int32, device=DEV)
        L.topk_logprob(lg2.to(DEV), ld2, rows2, V2, K, bs2.to(DEV), val2, idx2)
        ref3 = torch.log_softmax(lg2[:, :V2].double(), -1) + bs2[:, None].double()
        order = torch.argsort(-lg2[:, :V2].double(), dim=1, stable=True)[:, :K]          # value descending, lower token first
        assert torch.equal(idx2.cpu().long(), order)
        assert (val2.cpu().double() - torch.gather(ref3, 1, order)).abs().max() < 1e-5
        ban2 = int(order[0, 0]); ref4 = lg2[:, :V2].double().clone(); ref4[:, ban2] = -float("inf")
        order4 = torch.argsort(-ref4, dim=1, stable=True)[:, :K]
        L.topk_logprob(lg2.to(DEV), ld2, rows2, V2, K, bs2.to(DEV), val2, idx2, ban_token=ban2, pos_dev=pos, min_length=4)
        assert torch.equal(idx2.cpu().long(), order4)
        assert (val2.cpu().double() - torch.gather(ref3, 1, order4)).abs().max() < 1e-5
    # gather
    B, maxlen, W, n = 6, 20, 256, 13
    src = rnd(B, maxlen, W, seed=5); dst = torch.zeros_like(src)
    sel = torch.tensor([3, 3, 0, 5, 1, 1], dtype=torch.int32, device=DEV)
    L.kv_gather(src, dst, sel, maxlen * W, W, B, n, W)
    assert torch.equal(dst[:, :n], src[sel.long(), :n]) and dst[:, n:].abs().sum().item() == 0



def test_span_corrupt_vs_reference_golden(golden_dir):
    """v2s_span_corrupt against the outputs of the reference's util/t5.py (bit-exact), as ONE ragged batch incl. the
    lens <= 1 rule of dataset/dvc_dataset.py:141-144."""
    import os
    g = np.load(os.path.join(golden_dir, "data_pipeline.npz"))
    lens = [int(x) for x in g["sc_lens"]] + [1]
    Lm = max(lens); B = len(lens)
    ids = torch.zeros(B, Lm, dtype=torch.int64); noise = torch.zeros(B, Lm, dtype=torch.uint8)
    for i, n in enumerate(lens[:-1]):
        ids[i, :n] = torch.from_numpy(g[f"sc_ids_{n}"]); noise[i, :n] = torch.from_numpy(g[f"sc_mask_{n}"].astype(np.uint8))
    ids[-1, 0] = 1
    want_in = [g[f"sc_in_{n}"] for n in lens[:-1]] + [np.array([0])]
    want_out = [g[f"sc_out_{n}"] for n in lens[:-1]] + [np.array([1])]
    li, lo = max(len(x) for x in want_in), max(len(x) for x in want_out)
    den_in = torch.full((B, li), -7, dtype=torch.int64, device=DEV); den_out = torch.full((B, lo), -7, dtype=torch.int64, device=DEV)
    out_lens = torch.zeros(B, 2, dtype=torch.int32, device=DEV)
    L.span_corrupt(ids.to(DEV), torch.tensor(lens, dtype=torch.int32, device=DEV), noise.to(DEV), Lm, 32100, 1, den_in, den_out, out_lens)
    den_in, den_out, out_lens = den_in.cpu().numpy(), den_out.cpu().numpy(), out_lens.cpu().numpy()
    for i in range(B):
        assert out_lens[i].tolist() == [len(want_in[i]), len(want_out[i])]
        assert np.array_equal(den_in[i, :len(want_in[i])], want_in[i]) and not den_in[i, len(want_in[i]):].any()
        assert np.array_equal(den_out[i, :len(want_out[i])], want_out[i]) and not den_out[i, len(want_out[i]):].any()


def test_device_batcher_vs_oracle():
    from oracle import data_ref as D
    from vidchapters_amd.data import DeviceBatcher
    rng = np.random.RandomState(3)
    samples = []
    for n_frames, n_in, n_out in ((40, 60, 9), (100, 1, 5), (333, 250, 31), (1200, 17, 2)):
        ins = rng.randint(2, 32100, size=n_in).astype(np.int64); ins[-1] = 1
        outs = rng.randint(2, 32200, size=n_out).astype(np.int64); outs[-1] = 1
        samples.append({"video": rng.randn(n_frames, 64).astype(np.float32), "input_tokens": ins, "output_tokens": outs})
    np.random.seed(11)
    masks = [D.random_spans_noise_mask(len(s["input_tokens"]), 0.25, 5) if len(s["input_tokens"]) > 1 else np.zeros(1, bool) for s in samples]
    np.random.seed(11)
    batcher = DeviceBatcher(DEV, max_feats=100, num_text_tokens=32100)
    b = batcher(samples)                                        # draws the same masks from the seeded global numpy RNG
    batcher.ready.synchronize()
    want_video = np.stack([D.get_video(s["video"], 100) for s in samples])
    assert torch.equal(b["video"].cpu(), torch.from_numpy(want_video).to(torch.bfloat16))
    assert np.array_equal(b["input_ids"].cpu().numpy(), D.collate([s["input_tokens"] for s in samples]))
    assert np.array_equal(b["output_ids"].cpu().numpy(), D.collate([s["output_tokens"] for s in samples]))
    pairs = [D.span_corrupt(s["input_tokens"], m, 32100, 1) for s, m in zip(samples, masks)]
    assert np.array_equal(b["den_input_ids"].cpu().numpy(), D.collate([p[0] for p in pairs]))
    assert np.array_equal(b["den_output_ids"].cpu().numpy(), D.collate([p[1] for p in pairs]))


def test_topp_sampling_step_distribution():
    """v2s_topp_sample_step: (1) the filtered, renormalised distribution equals HF's temperature + top-p warpers (oracle.top_p_probs,
    itself identical to the installed transformers' warpers); (2) draws only ever hit kept tokens and their frequencies follow that
    distribution; (3) finished rows emit pad; (4) min_length bans EOS."""
    from oracle import vid2seq_ref as R
    rows, V, ld = 6, 612, 616
    g = torch.Generator().manual_seed(3)
    logits = torch.zeros(rows, ld); logits[:, :V] = torch.randn(rows, V, generator=g) * 2.5; logits[:, V:] = 50.0
    for top_p, temp, top_k in ((0.9, 1.0, 0), (0.5, 0.7, 0), (0.95, 1.2, 50), (1.0, 1.0, 7)):
        want = R.top_p_probs(logits[:, :V], top_p, temp, top_k)
        probs = torch.zeros(rows, V, device=DEV)
        nxt = torch.zeros(rows, dtype=torch.long, device=DEV); unf = torch.ones(rows, dtype=torch.int32, device=DEV)
        L.topp_sample_step(logits.to(DEV), ld, rows, V, top_p, temp, 1, nxt, unf, -1, 0, probs_out=probs, top_k=top_k)
        got = probs.cpu()
        edge = ((got > 0) != (want > 0)).sum().item()          # ties at the nucleus boundary may fall either way
        assert edge <= rows and (got - want).abs().max() < 2e-3
        if top_k:
            assert ((got > 0).sum(1) <= top_k).all() and (top_p < 1.0 or ((got > 0).sum(1) == top_k).all())
        counts = torch.zeros(rows, V)
        n_draw = 3000 if top_k in (0, 7) else 300
        lg = logits.to(DEV)
        for sd in range(n_draw):
            unf.fill_(1)
            L.topp_sample_step(lg, ld, rows, V, top_p, temp, 1000 + sd, nxt, unf, -1, 0, top_k=top_k)
            counts[torch.arange(rows), nxt.cpu()] += 1
        if n_draw < 3000:
            assert (counts[got == 0] == 0).all()
            continue
        assert (counts[got == 0] == 0).all()                  # never outside the nucleus
        freq = counts / n_draw
        assert (freq - got).abs().max() < 0.04, float((freq - got).abs().max())
    # finished rows emit pad; EOS banned below min_length
    unf = torch.tensor([1, 0, 1, 0, 1, 1], dtype=torch.int32, device=DEV)
    big = logits.clone(); big[:, 1] = 40.0                         # EOS overwhelmingly likely
    pos = torch.tensor([2], dtype=torch.int32, device=DEV); seq = torch.zeros(rows, 8, dtype=torch.long, device=DEV)
    L.topp_sample_step(big.to(DEV), ld, rows, V, 0.9, 1.0, 7, nxt, unf, 1, 0, seq_out=seq, seq_ld=8, pos_dev=pos, min_length=1)
    assert nxt.cpu().tolist() == [1, 0, 1, 0, 1, 1] and unf.cpu().tolist() == [0, 0, 0, 0, 0, 0] and seq[:, 3].cpu().tolist() == [1, 0, 1, 0, 1, 1]
    unf.fill_(1)
    L.topp_sample_step(big.to(DEV), ld, rows, V, 0.9, 1.0, 7, nxt, unf, 1, 0, seq_out=seq, seq_ld=8, pos_dev=pos, min_length=5)
    assert (nxt.cpu() != 1).all()


def test_beam_sample_candidates_vs_oracle():
    """v2s_beam_sample_cand against the oracle's restatement of one HF 4.28 beam_sample step (log-softmax, EOS ban, beam score,
    temperature / top-k / top-p warpers with min_tokens_to_keep 2, draws without replacement as the largest score + Gumbel keys) with
    the kernel's own counter-based noise restated on the host: kept sets, scores and keys per row, and -- after the host merge
    (beam.BeamScorer ordering) -- the same 2*nb (score, token, beam) triples per entry."""
    from oracle import vid2seq_ref as R
    B, nb, V, ld, K = 3, 4, 1500, 1504, 8
    rows = B * nb
    g = torch.Generator().manual_seed(9)
    logits = torch.zeros(rows, ld); logits[:, :V] = torch.randn(rows, V, generator=g) * 3; logits[:, V:] = 60.0
    bscore = torch.randn(rows, generator=g) * 2
    pos = torch.tensor([3], dtype=torch.int32, device=DEV)
    for top_p, temp, top_k, min_length in ((0.9, 1.0, 50, 1), (0.6, 0.7, 50, 9), (1.0, 1.3, 5, 1), (0.05, 1.0, 50, 1)):
        ban = 1 if 3 + 1 < min_length else -1
        val = torch.zeros(rows, K, device=DEV); key = torch.zeros(rows, K, device=DEV); tok = torch.zeros(rows, K, dtype=torch.int32, device=DEV)
        L.beam_sample_cand(logits.to(DEV), ld, rows, V, K, bscore.to(DEV), top_p, temp, top_k, 77, val, tok, key, ban_token=1, pos_dev=pos,
                           min_length=min_length)
        val, key, tok = val.cpu(), key.cpu(), tok.cpu().long()
        sc = torch.log_softmax(logits[:, :V], -1)
        if ban >= 0:
            sc[:, ban] = -float("inf")
        w = R.warp_scores(sc + bscore[:, None], top_p, temp, top_k, 2)
        noise = R.beam_sample_gumbel(77, 3, rows, V)
        kept = torch.isfinite(w).sum(1)
        assert (kept >= 2).all()
        if top_p <= 0.05:
            assert (kept == 2).all()                          # min_tokens_to_keep
        for r in range(rows):
            n = min(int(kept[r]), K)
            wk, wi = torch.topk(w[r] + noise[r], n)
            assert tok[r, :n].tolist() == wi.tolist(), (r, tok[r], wi)
            assert (val[r, :n] - w[r, wi]).abs().max() < 2e-4 and (key[r, :n] - wk).abs().max() < 2e-3
            assert torch.isinf(key[r, n:]).all() and torch.isinf(val[r, n:]).all()
        # host merge == the oracle's joint draw
        from vidchapters_amd.beam import BeamScorer
        scorer = BeamScorer(B, nb, 1.0, -5, 0, 0, 8, sample=True)
        new_tok, src, _ = scorer.advance(val.numpy(), tok.numpy().astype(np.int32), key.numpy())
        wv, wt, wb = R.beam_sample_step(logits[:, :V], bscore, nb, top_p, temp, top_k, noise, ban_token=ban)
        assert new_tok.reshape(B, nb).tolist() == wt[:, :nb].tolist()
        assert src.reshape(B, nb).tolist() == (wb[:, :nb] + torch.arange(B)[:, None] * nb).tolist()
        assert np.abs(scorer.scores - wv[:, :nb].numpy()).max() < 2e-4


def test_beam_sample_candidates_plateau_is_deterministic():
    """a plateau of ties at the k-th value puts more than 64 candidates at or above it; the kept 64 must not depend on
    atomic arrival order: everything strictly above the plateau plus the LOWEST token ids of the ties, identical from run to run."""
    rows, V, ld, K = 4, 3000, 3008, 8
    logits = torch.zeros(rows, ld)
    logits[:, :V] = 1.0                                   # plateau over the whole vocabulary ...
    logits[:, 100:110] = 5.0                              # ... with ten tokens strictly above it
    logits[:, V:] = 60.0
    bscore = torch.zeros(rows)
    pos = torch.tensor([3], dtype=torch.int32, device=DEV)
    outs = []
    for _ in range(3):
        val = torch.zeros(rows, K, device=DEV); key = torch.zeros(rows, K, device=DEV); tok = torch.zeros(rows, K, dtype=torch.int32, device=DEV)
        L.beam_sample_cand(logits.to(DEV), ld, rows, V, K, bscore.to(DEV), 1.0, 1.0, 50, 77, val, tok, key, ban_token=1, pos_dev=pos, min_length=1)
        outs.append((val.cpu(), tok.cpu(), key.cpu()))
    for v, t, k in outs[1:]:
        assert torch.equal(t, outs[0][1]) and torch.equal(v, outs[0][0]) and torch.equal(k, outs[0][2])
    tok = outs[0][1].long()
    allowed = set(range(100, 110)) | set(range(0, 54))    # the ten above the plateau + the 54 lowest token ids of the ties = 64 kept
    assert all(int(x) in allowed for x in tok.flatten()), tok


def test_decode_kernels():
    B, H, Nk = 3, 4, 333
    W = H * 64
    q = rnd(B, W, seed=1, scale=0.5); kc = rnd(B, 400, W, seed=2, scale=0.5); vc = rnd(B, 400, W, seed=3)
    bias = rnd(H, Nk, seed=4, dtype=torch.float32)
    mask = (torch.arange(Nk, device=DEV)[None, :] < torch.tensor([333, 200, 5], device=DEV)[:, None])
    o = torch.empty(B, W, dtype=torch.bfloat16, device=DEV)
    L.decode_attn(B, H, Nk, q, W, kc, vc, 400 * W, W, o, W, bias_row=bias, key_mask=mask.to(torch.uint8).contiguous(), mask_ld=Nk)
    s = torch.einsum("bhd,bkhd->bhk", q.float().view(B, H, 64), kc[:, :Nk].float().view(B, Nk, H, 64)) + bias[None]
    s = s + (~mask)[:, None, :].float() * torch.finfo(torch.float32).min
    ref = torch.einsum("bhk,bkhd->bhd", torch.softmax(s, -1), vc[:, :Nk].float().view(B, Nk, H, 64)).reshape(B, W)
    assert relerr(o, ref) < 1e-2
    # masked keys are never fetched: holes between valid keys, a padded tail, and a row without any valid key (which keeps the
    # reference's uniform average over its masked keys).  NaN planted in the K/V rows of masked keys must not reach the output.
    B2, Nk2 = 4, 1100
    g = torch.Generator().manual_seed(7)
    mask2 = torch.rand(B2, Nk2, generator=g) < 0.7
    mask2[0] = True; mask2[1, 640:] = False; mask2[3] = False
    mask2 = mask2.to(DEV)
    q2 = rnd(B2, W, seed=11, scale=0.5); k2 = rnd(B2, Nk2, W, seed=12, scale=0.5); v2 = rnd(B2, Nk2, W, seed=13)
    s2 = torch.einsum("bhd,bkhd->bhk", q2.float().view(B2, H, 64), k2.float().view(B2, Nk2, H, 64))
    s2 = s2 + (~mask2)[:, None, :].float() * torch.finfo(torch.float32).min
    ref2 = torch.einsum("bhk,bkhd->bhd", torch.softmax(s2, -1), v2.float().view(B2, Nk2, H, 64)).reshape(B2, W)
    poison = (~mask2) & mask2.any(1, keepdim=True)
    k2p, v2p = k2.clone(), v2.clone()
    k2p[poison] = float("nan"); v2p[poison] = float("nan")
    o2 = torch.empty(B2, W, dtype=torch.bfloat16, device=DEV)
    L.decode_attn(B2, H, Nk2, q2, W, k2p, v2p, Nk2 * W, W, o2, W, key_mask=mask2.to(torch.uint8).contiguous(), mask_ld=Nk2)
    assert torch.isfinite(o2.float()).all()
    assert relerr(o2, ref2) < 1e-2
    o3 = torch.empty_like(o2)                        # same result as with the real K/V in place (nothing depends on the skipped rows)
    L.decode_attn(B2, H, Nk2, q2, W, k2, v2, Nk2 * W, W, o3, W, key_mask=mask2.to(torch.uint8).contiguous(), mask_ld=Nk2)
    assert torch.equal(o2, o3)
    # beams of a batch entry share the encoder K/V (kv_group): one block per (entry, head) scores all of them (2 / 4 / 8), any other
    # group size keeps one block per row; both against a per-row reference
    for grp in (2, 3, 4, 8, 12, 16):           # (round 3: every width 2..16 runs the MFMA kernel -- scores and P.V on the matrix pipe)
        Bq = B2 * grp
        q4 = rnd(Bq, W, seed=31 + grp, scale=0.5)
        s4 = torch.einsum("bghd,bkhd->bghk", q4.float().view(B2, grp, H, 64), k2.float().view(B2, Nk2, H, 64))
        s4 = s4 + (~mask2)[:, None, None, :].float() * torch.finfo(torch.float32).min
        ref4 = torch.einsum("bghk,bkhd->bghd", torch.softmax(s4, -1), v2.float().view(B2, Nk2, H, 64)).reshape(Bq, W)
        o4 = torch.empty(Bq, W, dtype=torch.bfloat16, device=DEV)
        L.decode_attn(Bq, H, Nk2, q4, W, k2p, v2p, Nk2 * W, W, o4, W, key_mask=mask2.to(torch.uint8).contiguous(), mask_ld=Nk2, kv_group=grp)
        assert torch.isfinite(o4.float()).all()
        assert relerr(o4, ref4) < 1e-2, grp
        if grp in (4, 16):                      # no key mask at all, and a key count that is not a multiple of the 32-key chunk
            Nk3 = 333
            s5 = torch.einsum("bghd,bkhd->bghk", q4.float().view(B2, grp, H, 64), k2[:, :Nk3].float().view(B2, Nk3, H, 64))
            ref5 = torch.einsum("bghk,bkhd->bghd", torch.softmax(s5, -1), v2[:, :Nk3].float().view(B2, Nk3, H, 64)).reshape(Bq, W)
            o5 = torch.empty(Bq, W, dtype=torch.bfloat16, device=DEV)
            L.decode_attn(Bq, H, Nk3, q4, W, k2, v2, Nk2 * W, W, o5, W, kv_group=grp)
            assert relerr(o5, ref5) < 1e-2, grp
    # beam search without moving the cache: key k of row b is read from cache row row_map[b][k]; the step's own key comes from the
    # projection output, is appended to cache row b, and row_map[b][pos] becomes b
    Bm, maxlen, pv = 6, 40, 17
    cache = rnd(Bm, maxlen, 2 * W, seed=61, scale=0.5)
    rmap = torch.randint(0, Bm, (Bm, maxlen), generator=torch.Generator().manual_seed(5), dtype=torch.int32).to(DEV)
    qkv = rnd(Bm, 3 * W, seed=62, scale=0.5)
    posd = torch.tensor([pv], dtype=torch.int32, device=DEV)
    before = cache.clone()
    om = torch.empty(Bm, W, dtype=torch.bfloat16, device=DEV)
    L.decode_attn(Bm, H, maxlen, qkv, 3 * W, cache, cache[:, :, W:], maxlen * 2 * W, 2 * W, om, W, pos_dev=posd, bias_maxlen=maxlen,
                  new_k=qkv[:, W:], new_v=qkv[:, 2 * W:], new_bs=3 * W, row_map=rmap, row_map_ld=maxlen)
    ar = torch.arange(pv, device=DEV)
    hist = before[rmap[:, :pv].long(), ar[None, :]]                                  # [Bm, pv, 2W]
    Kf = torch.cat([hist[:, :, :W], qkv[:, None, W:2 * W]], 1).float().view(Bm, pv + 1, H, 64)
    Vf = torch.cat([hist[:, :, W:], qkv[:, None, 2 * W:]], 1).float().view(Bm, pv + 1, H, 64)
    sm = torch.softmax(torch.einsum("bhd,bkhd->bhk", qkv[:, :W].float().view(Bm, H, 64), Kf), -1)
    assert relerr(om, torch.einsum("bhk,bkhd->bhd", sm, Vf).reshape(Bm, W)) < 1e-2
    assert torch.equal(cache[:, pv], qkv[:, W:]) and torch.equal(cache[:, :pv], before[:, :pv])
    assert rmap[:, pv].tolist() == list(range(Bm))
    # argmax: 16-byte and scalar paths, ties resolve to the lowest index
    for V2 in (32128, 32201, 7):
        lg = rnd(B, V2, seed=21, dtype=torch.float32)
        lg[0, V2 - 1] = 50.0; lg[0, V2 // 2] = 50.0; lg[1, 3] = 60.0; lg[2, V2 - 1] = 70.0
        nx = torch.empty(B, dtype=torch.int64, device=DEV); un = torch.ones(B, dtype=torch.int32, device=DEV)
        L.argmax_step(lg, V2, B, V2, nx, un, -1, 0)
        assert nx.tolist() == [V2 // 2, 3, V2 - 1]
        # torch.argmax's answer on NaN / -inf rows (the first NaN wins; a row of -inf gives 0): never an out-of-range token
        lg[0] = float("nan"); lg[1] = float("-inf"); lg[2, V2 - 2] = float("nan"); lg[2, 2] = float("nan")
        un.fill_(1)
        L.argmax_step(lg, V2, B, V2, nx, un, -1, 0)
        assert nx.tolist() == [0, 0, 2] == lg.argmax(-1).tolist()
    logits = rnd(B, 32200, seed=5, dtype=torch.float32)
    nxt = torch.empty(B, dtype=torch.int64, device=DEV); unf = torch.tensor([1, 0, 1], dtype=torch.int32, device=DEV)
    logits[2, 1] = 100.0                                          # row 2 emits EOS
    L.argmax_step(logits, 32200, B, 32200, nxt, unf, 1, 0)
    assert nxt.tolist() == [logits[0].argmax().item(), 0, 1] and unf.tolist() == [1, 0, 0]
    src = rnd(B, W, seed=6)
    L.kv_append(src, W, kc, 400 * W, W, B, W, 399)
    assert torch.equal(kc[:, 399], src)


@pytest.mark.parametrize("G", [1, 2, 3, 4])
def test_decode_cross_attention_on_the_shared_memory(G):
    """v2s_decode_qfold + v2s_decode_memattn + v2s_decode_ctxfold (the decode step's cross-attention without per-layer K / V caches)
    against the plain formulation: q = RMSNorm(x) Wq^T, K = mem Wk^T, V = mem Wv^T, softmax(q_h K_h^T) V_h over the valid key prefix
    (modeling_t5.py:484-561; T5: no scaling, no position bias in cross-attention).  Key counts around the 32-key tile, one key, and
    more splits than tiles."""
    H, d, S = 12, 768, 700
    klen_l = [700, 1, 31, 32, 33, 389]
    E = len(klen_l)
    rows = E * G
    x = rnd(rows, d, seed=101)
    lnw = 1.0 + 0.1 * rnd(d, seed=102, dtype=torch.float32)
    wq = rnd(H * 64, d, seed=103, scale=0.25 * d ** -0.5)
    wk = rnd(H * 64, d, seed=104, scale=d ** -0.5)
    wv = rnd(H * 64, d, seed=105, scale=d ** -0.5)
    mem = rnd(E, S, d, seed=106)
    klen = torch.tensor(klen_l, dtype=torch.int32, device=DEV)
    eps = 1e-6
    wqf = torch.empty_like(wq)
    L.scale_cols(wq, lnw, wqf, H * 64, d)
    # reference (fp32 on the bf16 inputs; q rounded to bf16 like the K/V-cache path's projection output)
    xf = x.float()
    xn = xf * torch.rsqrt((xf * xf).mean(1, keepdim=True) + eps)
    qr = (xn @ wqf.float().t()).bfloat16().float().view(E, G, H, 64)
    K = (mem.float() @ wk.float().t()).view(E, S, H, 64)
    V = (mem.float() @ wv.float().t()).view(E, S, H, 64)
    sc = torch.einsum("eghd,ekhd->eghk", qr, K)
    valid = torch.arange(S, device=DEV)[None, :] < klen[:, None]
    sc = sc.masked_fill(~valid[:, None, None, :], float("-inf"))
    ref = torch.einsum("eghk,ekhd->eghd", torch.softmax(sc, -1), V).reshape(rows, H * 64)
    # folded queries alone
    qp = torch.empty(rows, H, d, dtype=torch.bfloat16, device=DEV)
    L.decode_qfold(x, rows, wqf, wk.t().contiguous(), eps, qp, H, d)
    qp_ref = torch.einsum("rhj,hjc->rhc", qr.view(rows, H, 64), wk.float().view(H, 64, d))
    assert relerr(qp, qp_ref) < 6e-3
    # the unfused form the engine uses beyond 64 rows: RMSNorm kernel first, then the fold without a row scale (rms_eps = 0)
    nrm = torch.empty_like(x); rstd = torch.empty(rows, dtype=torch.float32, device=DEV)
    L.rmsnorm_fwd(x, lnw, nrm, rstd, rows, d, eps)
    qp_u = torch.empty_like(qp)
    L.decode_qfold(nrm, rows, wq, wk.t().contiguous(), 0.0, qp_u, H, d)
    assert relerr(qp_u, qp_ref) < 1e-2
    memp = mem.clone()
    for e, n in enumerate(klen_l):                    # rows past the valid prefix are never read into a result
        memp[e, n:] = float("nan")
    for tpp in (1000, 8, 3, 1):                      # one piece per entry ... one tile per piece
        plan = L.MemAttnPlan(klen_l, G * H, DEV, tiles_per_piece=tpp)
        want = sum(-(-(-(-n // 32)) // tpp) for n in klen_l)
        assert plan.slot_off_host[-1] == plan.nblk == want
        plan.part.fill_(float("nan")); plan.ml.fill_(float("nan"))
        ctx = torch.empty(rows, H * 64, dtype=torch.bfloat16, device=DEV)
        L.decode_memattn(qp, memp, S * d, plan, d)
        L.decode_ctxfold(plan, rows, G, H, wv, ctx, d)
        assert torch.isfinite(ctx.float()).all(), tpp
        assert relerr(ctx, ref) < 1.5e-2, (tpp, relerr(ctx, ref))
    # scores that jump by far more than the 44-nat headroom of the kernel's fixed exponent reference (late keys 40 x larger: hundreds
    # of nats, twice): the reference moves up and the sums of that query restart from zero (what was summed before weighs < 2^-54).
    # Logits that large amplify any rounding of the queries, so the
    # reference here starts from the kernel's own folded queries: softmax(qp . mem) mem, then Wv per head.
    mem2 = mem.clone()
    mem2[:, 300:] *= 40.0
    mem2[:, 500:] *= 8.0                              # a second, larger jump further on (references move more than once per stream)
    sc2 = torch.einsum("eghc,ekc->eghk", qp.float().view(E, G, H, d), mem2.float()).masked_fill(~valid[:, None, None, :], float("-inf"))
    assert float((sc2.amax(-1) - sc2[..., :16].amax(-1)).max()) > 60.0
    accn = torch.einsum("eghk,ekc->eghc", torch.softmax(sc2, -1), mem2.float()).bfloat16().float()
    ref2 = torch.einsum("eghc,hjc->eghj", accn, wv.float().view(H, 64, d)).reshape(rows, H * 64)
    plan = L.MemAttnPlan(klen_l, G * H, DEV, tiles_per_piece=16)
    ctx2 = torch.empty(rows, H * 64, dtype=torch.bfloat16, device=DEV)
    L.decode_memattn(qp, mem2, S * d, plan, d)
    L.decode_ctxfold(plan, rows, G, H, wv, ctx2, d)
    assert torch.isfinite(ctx2.float()).all() and relerr(ctx2, ref2) < 2e-2, relerr(ctx2, ref2)
    # an entry's result does not depend on the batch it sits in: entries 0 and 5 alone, in the other order
    sub = [5, 0]
    rows_s = torch.tensor([e * G + gg for e in sub for gg in range(G)], device=DEV)
    plan_s = L.MemAttnPlan([klen_l[e] for e in sub], G * H, DEV)
    plan_f = L.MemAttnPlan(klen_l, G * H, DEV)
    ctx_f = torch.empty(rows, H * 64, dtype=torch.bfloat16, device=DEV)
    L.decode_memattn(qp, memp, S * d, plan_f, d)
    L.decode_ctxfold(plan_f, rows, G, H, wv, ctx_f, d)
    qp_s = torch.empty(len(sub) * G, H, d, dtype=torch.bfloat16, device=DEV)
    L.decode_qfold(x[rows_s].contiguous(), len(sub) * G, wqf, wk.t().contiguous(), eps, qp_s, H, d)
    assert torch.equal(qp_s, qp[rows_s])
    ctx_s = torch.empty(len(sub) * G, H * 64, dtype=torch.bfloat16, device=DEV)
    L.decode_memattn(qp_s, memp[sub].contiguous(), S * d, plan_s, d)
    L.decode_ctxfold(plan_s, len(sub) * G, G, H, wv, ctx_s, d)
    assert torch.equal(ctx_s, ctx_f[rows_s])
    # scores that keep climbing (every 16-key group 3 x the previous over the last 20 groups: ~20 reference moves per stream): the cost of a
    # jump must stay O(1) -- an earlier version redid the stream on every jump and took 4 x longer on a model trained for a few steps
    if G == 1:
        mem3 = mem.clone()
        for k in range(20):
            mem3[:, 380 + 16 * k:] *= 3.0
        mem3 = mem3.clamp(-3e4, 3e4)
        plan3 = L.MemAttnPlan(klen_l, G * H, DEV)
        ctx3 = torch.empty(rows, H * 64, dtype=torch.bfloat16, device=DEV)
        def timed(m):
            L.decode_memattn(qp, m, S * d, plan3, d); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                L.decode_memattn(qp, m, S * d, plan3, d)
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1)
        # (a wall-clock comparison: the minimum of five interleaved measurements each -- one 0.5 ms loop alone lost a full-suite run of round 6 to
        # a hiccup of the box; the defect this guards against was a factor 4)
        tp, tc = [], []
        for _ in range(5):
            tp.append(timed(mem)); tc.append(timed(mem3))
        t_plain, t_climb = min(tp), min(tc)
        L.decode_ctxfold(plan3, rows, G, H, wv, ctx3, d)
        assert torch.isfinite(ctx3.float()).all()
        assert t_climb < 2.0 * t_plain + 0.2, (t_plain, t_climb, tp, tc)
    with pytest.raises(RuntimeError):
        L.MemAttnPlan([5, 0, 7], G * H, DEV)
    with pytest.raises(ValueError):                   # a query buffer that does not match the plan
        L.decode_memattn(qp[:-1], memp, S * d, plan, d)
    plan.R = 49                                       # more than 48 query rows per entry: refused by the library
    with pytest.raises(RuntimeError):
        L.decode_memattn(torch.empty(E * 49 * d, dtype=torch.bfloat16, device=DEV), memp, S * d, plan, d)


@pytest.mark.parametrize("nb,K,lp", [(4, 8, 1.0), (2, 4, 0.6), (3, 8, 1.0), (8, 16, 2.0), (16, 32, 1.0), (1, 2, 1.0)])
def test_beam_advance_on_device_equals_the_host_scorer(nb, K, lp):
    """v2s_beam_advance (BeamSearchScorer.process + BeamHypotheses.add on the device) against vidchapters_amd/beam.py, the host
    restatement of transformers 4.28's scorer: random per-beam sorted candidate lists with many EOS candidates and many ties, several
    entries, until every entry is done or the length runs out -- next tokens, scores, source rows, token history, row map, done flags,
    every step; the finished hypotheses (in insertion order) and the finalized sequences at the end."""
    from vidchapters_amd.beam import BeamScorer
    rng = np.random.default_rng(nb * 100 + K)
    B, maxlen, eos, pad, start, V = 5, 24, 1, 0, 0, 40
    R = B * nb
    sc = BeamScorer(B, nb, lp, eos, pad, start, maxlen + 1)
    st = L.BeamState(B, nb, maxlen + 1, DEV, lp)
    hist = torch.from_numpy(sc.seqs.copy()).to(DEV)
    row_map = torch.zeros(R, maxlen, dtype=torch.int32, device=DEV)
    host_map = np.zeros((R, maxlen), dtype=np.int32)
    nxt = torch.zeros(R, dtype=torch.long, device=DEV)
    bscore = torch.from_numpy(sc.scores.reshape(-1).copy()).to(DEV)
    src = torch.zeros(R, dtype=torch.int32, device=DEV)
    pos = torch.zeros(1, dtype=torch.int32, device=DEV)
    steps = 0
    for t in range(maxlen):
        # per-row candidates: running score + sorted steps on a coarse grid (ties across beams), EOS with probability ~ 1/4 from step 2
        base = sc.scores.reshape(-1).astype(np.float32)
        inc = np.sort(rng.integers(1, 12, size=(R, K)), axis=1).astype(np.float32) * np.float32(-0.25)
        val = (np.maximum(base, np.float32(-50.0))[:, None] + inc).astype(np.float32)
        tok = np.stack([rng.permutation(np.arange(2, V))[:K] for _ in range(R)]).astype(np.int32)
        if t >= 2:
            put = rng.random(R) < 0.3
            tok[put, rng.integers(0, min(K, 3))] = eos
        host_map[:, t] = np.arange(R)
        row_map[:, t] = torch.arange(R, dtype=torch.int32, device=DEV)
        was_done = sc.done.copy()
        ntok, nsrc, fin = sc.advance(val.copy(), tok.copy())
        L.beam_advance(torch.from_numpy(val).to(DEV), torch.from_numpy(tok).to(DEV), K, st, eos, pad, pos, hist, row_map, nxt, bscore, src)
        L.counter_add(pos, 1)
        steps += 1
        live = np.repeat(~was_done, nb)              # finished entries: pad tokens, rows nobody reads any more
        assert np.array_equal(nxt.cpu().numpy(), ntok)
        assert np.array_equal(bscore.cpu().numpy()[live], sc.scores.reshape(-1)[live])
        assert np.array_equal(src.cpu().numpy()[live], nsrc[live])
        host_map[live] = host_map[nsrc][live]
        assert np.array_equal(hist.cpu().numpy()[live], sc.seqs[live]), t
        assert np.array_equal(row_map.cpu().numpy()[live][:, :t + 1], host_map[live][:, :t + 1])
        assert np.array_equal(st.done.cpu().numpy().astype(bool), sc.done)
        assert int(st.ndone.item()) == int(sc.done.sum())
        if fin:
            break
    assert sum(len(h.items) for h in sc.heaps) >= B and steps > 3          # (finished hypotheses everywhere; most parametrisations also finish entries)
    hn, ht, hl = st.heap_n.cpu().numpy(), st.hyp_tok.cpu().numpy(), st.hyp_len.cpu().numpy()
    hs, ho = st.hyp_score.cpu().numpy(), st.hyp_order.cpu().numpy()
    for b in range(B):
        slots = sorted(range(int(hn[b])), key=lambda i: int(ho[b, i]))
        assert len(slots) == len(sc.heaps[b].items)
        for i, (score, toks) in zip(slots, sc.heaps[b].items):
            assert hs[b, i] == score
            assert np.array_equal(ht[b, i, :hl[b, i]], toks)
        assert st.heap_worst.cpu().numpy()[b] == sc.heaps[b].worst
    with pytest.raises(RuntimeError):
        L.beam_advance(torch.zeros(17 * 64, device=DEV), torch.zeros(17 * 64, dtype=torch.int32, device=DEV), 64, L.BeamState(1, 17, 8, DEV),
                       eos, pad, pos, torch.zeros(17, 8, dtype=torch.long, device=DEV), None, torch.zeros(17, dtype=torch.long, device=DEV),
                       torch.zeros(17, device=DEV), torch.zeros(17, dtype=torch.int32, device=DEV))
