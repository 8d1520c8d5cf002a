"""CPU tests of the GENERATED K loop of gemm_a4_kernel (vidchapters_amd/csrc/gen_gemm_a4.py -> v2s_gemm_a4.inc): the generated
instruction text is executed on the functional model of tools/a4_emu.py (4 waves x 64 lanes, LDS, LDS-DMA, ds_read_b128 /
ds_read_b64_tr_b16, v_mfma_f32_32x32x16_bf16, scalar loop control) and must reproduce X . W^T for both operand layouts under every
combination of lazy / eager completion of LDS reads and DMA writes and several wave interleavings -- which is what checks the counted
s_waitcnt values and the barrier protocol, not just the address arithmetic.  (The -m gpu counterpart is
tests/test_kernels_gpu.py::test_gemm_a4_kernel.)"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import a4_emu as E  # noqa: E402

GEN = os.path.join(ROOT, "vidchapters_amd", "csrc", "gen_gemm_a4.py")
MODES = ((False, False, "fwd"), (True, True, "random"), (True, False, "random"), (False, True, "rev"), (True, True, "fwd"))


@pytest.fixture(scope="module")
def inc(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("a4") / "v2s_gemm_a4.inc")
    subprocess.run([sys.executable, GEN, out], check=True)
    return out


def test_committed_inc_is_what_the_generator_emits(inc):
    committed = os.path.join(ROOT, "vidchapters_amd", "csrc", "v2s_gemm_a4.inc")
    assert open(committed).read() == open(inc).read(), "v2s_gemm_a4.inc is stale: run gen_gemm_a4.py"


@pytest.mark.parametrize("tb", [False, True])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (600, 512, 384)])
def test_generated_loop_computes_the_product(inc, tb, M, N, K):
    tile = ((M - 1) // 256, (N - 1) // 256)          # the ragged last tile row
    for lazy_ds, lazy_dma, sched in MODES:
        err = E.check(inc, tb, M, N, K, tile=tile, lazy_ds=lazy_ds, lazy_dma=lazy_dma, sched=sched)
        assert err < 2e-5, (tb, M, N, K, lazy_ds, lazy_dma, sched, err)


def test_weight_gradient_form(inc):
    """A stored [K][M] (dY^T X): tr-read fragments for both operands"""
    for lazy_ds, lazy_dma, sched in MODES[1:4]:
        assert E.check(inc, True, 520, 304, 256, tile=(2, 1), lazy_ds=lazy_ds, lazy_dma=lazy_dma, sched=sched, ta=True) < 2e-5


@pytest.mark.parametrize("tb,M,N,K,grid", [(False, 512, 768, 512, 2), (True, 768, 512, 384, 1), (False, 600, 520, 384, 2)])
def test_persistent_deferred_writeout(inc, tb, M, N, K, grid):
    """gemm_a4p: `grid` blocks walk all tiles; conversion at the first step of the next tile, wave-private LDS transposition, buffer stores
    (the first tile's through a zero-length descriptor), DMA stream across tile edges, drain after the last tile: every bf16 output exact"""
    for lazy_ds, lazy_dma, sched in MODES[1:4]:
        assert E.check_p(inc, tb, M, N, K, grid, lazy_ds=lazy_ds, lazy_dma=lazy_dma, sched=sched) == 0


def test_persistent_grouped_tile_walk(inc):
    """the scalar tile walk of gemm_a4p: groups of GM tile rows walked column by column (GM = 1: row-major); a short last group, a last group of
    ONE row (no 32-bit division magic for 1) and GM > tile rows; every tile must be visited once: any miss leaves NaNs, any wrong address wrong values"""
    for GM, (M, N, grid) in zip((2, 4, 3, 8), ((1280, 768, 3), (1280, 768, 2), (1024, 1024, 3), (768, 768, 2))):
        assert E.check_p(inc, False, M, N, 384, grid, lazy_ds=True, lazy_dma=True, sched="random", GM=GM) == 0
    # the host-side arithmetic on its own, exhaustively: (tile rows, tile columns, GM) -> a permutation of the tiles
    for tm in range(1, 24):
        for tn in (2, 3, 12, 16, 32):
            for GM in (1, 2, 3, 4, 5, 8, 40):
                gsz, mg, mm, ml, wk = E.walk_args(tm, tn, GM)
                seen = set()
                for t in range(tm * tn):
                    grp = (t * mg) >> 32; r = t - grp * gsz
                    gm, m = ((wk >> 8) & 0xFF, ml) if grp == wk >> 16 else (wk & 0xFF, mm)
                    c = r if gm == 1 else (r * m) >> 32
                    seen.add((grp * (wk & 0xFF) + r - c * gm, c))
                assert seen == {(i, j) for i in range(tm) for j in range(tn)}, (tm, tn, GM)


def test_persistent_relu_mask_epilogue(inc):
    """gemm_a4p with dact = RELU: the z tile prefetched into the held registers during the tile's third iteration, masked + scaled at the
    conversion (z > 0 incl. -0 and negative z; scale 1 / 0.9); K = 512 is the shortest contraction it takes (T, S, Z and one loop iteration)"""
    for (M, N, K, grid), (lazy_ds, lazy_dma, sched) in zip(((512, 768, 640, 2), (600, 520, 512, 1)), (MODES[1], MODES[3])):
        assert E.check_p(inc, True, M, N, K, grid, lazy_ds=lazy_ds, lazy_dma=lazy_dma, sched=sched, dact=True, scale=1.0 / 0.9) == 0


def test_persistent_relu_and_dropout_forward_epilogues(inc):
    """gemm_a4p with act = RELU (bf16 sign test after the rounding) and with ReLU + dropout: the keep masks of the library's counter-based
    generator (v2s_keep8, restated in numpy: a4_emu.keep_mask) are computed into the held registers in the MFMA gaps of the tile's second to
    fourth iteration and applied at the conversion; ragged tiles overlap, so their rows are masked by GLOBAL index twice, identically"""
    assert E.check_p(inc, False, 512, 768, 384, 2, lazy_ds=True, lazy_dma=True, sched="random", epi="relu") == 0
    for (M, N, K, grid, p16, hs), (lazy_ds, lazy_dma, sched) in zip(((512, 768, 640, 2, 6554, 0x1234567), (600, 520, 768, 1, 32768, 0xdeadbeef)), (MODES[1], MODES[3])):
        assert E.check_p(inc, False, M, N, K, grid, lazy_ds=lazy_ds, lazy_dma=lazy_dma, sched=sched, epi="reludrop", scale=1.0 / 0.9, p16=p16, hseed=hs) == 0


def test_the_model_catches_a_wrong_wait_a_missing_barrier_and_a_wrong_slot(inc, tmp_path):
    """the emulator is only worth something if it rejects broken schedules: three mutations of the generated text"""
    src = open(inc).read()

    def nth(s, pat, rep, n):
        idx = [m.start() for m in re.finditer(re.escape(pat), s)]
        return s[:idx[n]] + rep + s[idx[n] + len(pat):]

    muts = {"vmcnt": nth(src, "s_waitcnt vmcnt(16) lgkmcnt(0)", "s_waitcnt vmcnt(28) lgkmcnt(0)", 2),
            "barrier": nth(src, "s_barrier", "s_nop 0", 3),
            "lgkm": nth(src, "s_waitcnt lgkmcnt(6)", "s_waitcnt lgkmcnt(7)", 10),
            "slot": nth(src, "s_add_u32 m0, s40, 0x1c400", "s_add_u32 m0, s40, 0x14400", 1)}
    for name, text in muts.items():
        p = str(tmp_path / f"{name}.inc")
        open(p, "w").write(text)
        caught = False
        for lazy_ds, lazy_dma, sched in MODES:
            try:
                err = E.check(p, False, 256, 256, 256, lazy_ds=lazy_ds, lazy_dma=lazy_dma, sched=sched)
                caught |= not (err < 1e-3)           # NaN (poisoned register read) or a wrong product
            except (AssertionError, RuntimeError):
                caught = True
        assert caught, name
