"""HIP logits against the REFERENCE's (VERDICT r05 missing #3 / weak #1; SURVEY 8c: "logits cosine >= 0.999").

The loss of the synthetic init is ln V to three digits (near-uniform logits), so "loss rel <= 2e-3" alone would pass with a decoder that
outputs noise.  These tests compare the logits themselves (model/modeling_t5.py:1709-1714) -- captured from the engine's LM head through
the test-only tap Engine.dbg_logits -- with the fixtures written by the real reference (oracle/make_golden.py: case_tiny, case_full,
case_sharp), and add a configuration whose loss is far below ln V and moves with every logit (tests/golden/sharp_cfg1.npz: the decoder's
final norm gain x 16, targets = the reference's own greedy continuation)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vid2seq_ref as R                               # noqa: E402  (checker only)
from oracle.make_golden import SHARP_KEY, grad_sample            # noqa: E402  (sampling rule / constants of the fixtures)
from vidchapters_amd import SyntheticTokenizer, Vid2Seq, synth   # noqa: E402
from vidchapters_amd.train import Trainer                        # noqa: E402

DEV = "cuda"


def tok(ids):
    ids = ids.to(DEV)
    return {"input_ids": ids, "attention_mask": ids != 0}


def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float(a @ b / (a.norm() * b.norm() + 1e-30))


def hip_logits(model, video, input_ids, output_ids, pack):
    """([B, Lo, V] fp32 logits of the HIP LM head, loss tensor, video_dict): dense rows, or -- pack = True, the engine's default padding-free decoder -- the
    rows of real targets scattered back (the rest NaN)"""
    eng = model.engine()
    was = eng.pack
    eng.pack = pack
    eng.dbg_logits = []
    try:
        out, vd = model(video.to(DEV), tok(input_ids), tok(output_ids))
        rows = torch.cat(eng.dbg_logits, 0).float()
    finally:
        eng.dbg_logits = None
        eng.pack = was
    B, Lo = output_ids.shape
    if rows.shape[0] == B * Lo:
        return rows.view(B, Lo, -1).cpu(), out["loss"], vd
    plan = eng._pack_plan_dec(output_ids.to(DEV) != 0)
    assert plan is not None and plan[1] == rows.shape[0]
    full = torch.full((B * Lo, rows.shape[1]), float("nan"), device=DEV)
    full[plan[2]] = rows
    return full.view(B, Lo, -1).cpu(), out["loss"], vd


def compare_rows(tag, got, want, margin_min):
    """per-row cosine, worst |difference|, arg-max agreement on the rows the reference decides by more than ``margin_min``"""
    ok = ~torch.isnan(got).any(-1)
    g, w = got[ok], want[ok]
    rc = torch.nn.functional.cosine_similarity(g.double(), w.double(), dim=-1)
    top2 = w.topk(2, -1).values
    firm = (top2[:, 0] - top2[:, 1]) > margin_min
    agree_all = float((g.argmax(-1) == w.argmax(-1)).float().mean())
    agree_firm = float((g.argmax(-1) == w.argmax(-1))[firm].float().mean()) if firm.any() else float("nan")
    print(f"[{tag}] {int(ok.sum())} rows: cosine overall {cos(g, w):.7f}, worst row {float(rc.min()):.7f}; max |hip - reference| {float((g - w).abs().max()):.4f} "
          f"(logit rms {float(w.pow(2).mean().sqrt()):.3f}); arg-max equal on {100 * agree_all:.2f} % of all rows, {100 * agree_firm:.2f} % of the "
          f"{int(firm.sum())} rows the reference decides by > {margin_min}")
    return float(rc.min()), agree_firm


@pytest.mark.parametrize("tag,cfg,seed", [
    ("small", R.RefConfig.small(), 7),
    ("small_resize_proj", R.RefConfig.small(vit_dim=64, vit_heads=1, num_features=10), 9),
])
def test_logits_vs_reference_golden_small(golden_dir, tag, cfg, seed):
    g = np.load(os.path.join(golden_dir, f"{tag}_forward_backward.npz"))
    t5 = dict(d_model=cfg.d_model, d_kv=cfg.d_kv, heads=cfg.heads, d_ff=cfg.d_ff, n_enc=cfg.n_enc, n_dec=cfg.n_dec)
    model = Vid2Seq(t5, num_features=cfg.num_features, embed_dim=cfg.vit_dim, depth=cfg.vit_depth, heads=cfg.vit_heads, mlp_dim=cfg.vit_mlp,
                    tokenizer=SyntheticTokenizer(cfg.vocab - cfg.num_bins, cfg.num_bins), vis_drop=0.0, enc_drop=0.0, dec_drop=0.0, num_bins=cfg.num_bins,
                    init_seed=seed).to(DEV).eval()
    want = torch.from_numpy(g["logits"])
    # the oracle's bf16 mode on the same inputs: what rounding alone does to these logits (at d_model 128 a row's cosine against the fp32
    # reference drops to 0.9993 from rounding alone), and a second, tighter target
    P = synth.init_params(R.param_shapes(cfg), seed, cfg.d_model, cfg.inner, cfg.d_ff)
    v, i_, o_ = (torch.from_numpy(g[k]) for k in ("video", "input_ids", "output_ids"))
    with torch.no_grad(), R.bf16_mode():
        wantb, _, _ = R.vid2seq_logits(P, cfg, v, i_, i_ != 0, o_, o_ != 0)
    compare_rows(f"{tag}: the oracle's bf16 mode against the reference (rounding alone)", wantb.view(-1, wantb.shape[-1]), want.view(-1, want.shape[-1]), 0.05)
    for pack in (False, True):
        got, _, _ = hip_logits(model, v, i_, o_, pack)
        flat = got.view(-1, got.shape[-1])
        worst, firm = compare_rows(f"{tag}, pack={pack}", flat, want.view(-1, want.shape[-1]), 0.05)
        ok = ~torch.isnan(flat).any(-1)
        overall = cos(flat[ok], want.view(-1, want.shape[-1])[ok])
        assert overall > 0.999 and worst > 0.998     # cosine >= 0.999; single rows at d_model 128: measured 0.9990, rounding alone 0.9993
        assert not firm < 1.0                        # every row decided by more than the bf16 noise of a logit has the reference's arg-max
        worstb, _ = compare_rows(f"{tag}, pack={pack}, against the oracle's bf16 mode", flat, wantb.view(-1, wantb.shape[-1]), 0.05)
        assert worstb > 0.998


def test_logits_cfg1_vs_reference_golden(golden_dir):
    """t5-base at cfg-1 (B = 2, 100 frames, 256 ASR tokens, 256 targets): four whole logit rows, the [:, :4, :64] slice, every row's maximum and
    arg-max of the REAL reference's logits (full_cfg1_scalars.npz)"""
    g = np.load(os.path.join(golden_dir, "full_cfg1_scalars.npz"))
    seed, B, L, Lo = (int(g[k]) for k in ("seed", "B", "L", "Lo"))
    model = Vid2Seq("t5-base", tokenizer=SyntheticTokenizer(32100, 100), vis_drop=0.0, enc_drop=0.0, dec_drop=0.0, init_seed=seed, device=DEV).eval()
    b = synth.make_batch(B, 100, L, Lo, 32200, seed, 768)
    got, loss, _ = hip_logits(model, b["video"], b["input_ids"], b["output_ids"], pack=False)
    rows = torch.stack([got[int(bi), int(j)] for bi, j in g["logits_rows_pos"]])
    worst, _ = compare_rows("cfg-1 whole rows", rows, torch.from_numpy(g["logits_rows"]), 0.02)
    c_slice = cos(got[:, :4, :64], torch.from_numpy(g["logits_slice"]))
    dmax = float((got.max(-1).values - torch.from_numpy(g["logits_rowmax"])).abs().max())
    margin = torch.from_numpy(g["logits_margin"])
    same = got.argmax(-1) == torch.from_numpy(g["logits_argmax"])
    firm = margin > 0.02
    print(f"[cfg-1] slice cosine {c_slice:.7f}; row maxima within {dmax:.4f} (rms of a logit ~0.2); arg-max equal on {100 * float(same.float().mean()):.2f} % of the "
          f"{same.numel()} rows, on {100 * float(same[firm].float().mean()):.2f} % of the {int(firm.sum())} rows with a reference margin > 0.02 "
          f"(median margin {float(margin.median()):.3f}: the synthetic init's logits are near-uniform); loss hip {loss.item():.6f} reference {float(g['loss']):.6f}")
    # the oracle's OWN bf16 mode against these fp32 rows (8 / 3 BLAS threads, round 6): cosine 0.9984 - 0.9993, row maxima within 0.030 -- 24 bf16 layers
    # of rounding; SURVEY 8c's 0.999 holds for the slice and overall, single whole rows sit at the rounding floor
    assert worst > 0.998 and c_slice > 0.999
    assert dmax < 0.06
    assert float(same[firm].float().mean()) > 0.97


def test_sharp_logits_and_loss_vs_reference_golden(golden_dir):
    """A loss that can see the decoder: final-norm gain x 16 (peaked logits), targets = the reference's greedy continuation -> loss 3.5 << ln V = 10.4.
    Loss, per-row log-sum-exp / target logit / maximum, whole rows and gradients against the REAL reference (sharp_cfg1.npz)."""
    g = np.load(os.path.join(golden_dir, "sharp_cfg1.npz"))
    seed, B, T, L, Lo = (int(g[k]) for k in ("seed", "B", "T", "L", "Lo"))
    model = Vid2Seq("t5-base", tokenizer=SyntheticTokenizer(32100, 100), vis_drop=0.0, enc_drop=0.0, dec_drop=0.0, init_seed=seed, device=DEV).eval()
    with torch.no_grad():
        dict(model.named_parameters())[SHARP_KEY].mul_(float(g["scale"]))
    b = synth.make_batch(B, T, L, Lo, 32200, seed, 768)
    out_ids = torch.from_numpy(g["output_ids"])
    got, loss, _ = hip_logits(model, b["video"], b["input_ids"], out_ids, pack=False)
    ref, refb = float(g["loss"]), float(g["loss_bf16mode"])
    rel, relb = abs(loss.item() - ref) / ref, abs(loss.item() - refb) / refb
    print(f"[sharp] loss hip {loss.item():.6f} reference {ref:.6f} (rel {rel:.1e}) bf16-mode oracle {refb:.6f} (rel {relb:.1e}); ln V = {np.log(32200):.4f}")
    assert ref < 0.5 * np.log(32200)
    # the oracle's bf16 mode gives 3.5201 with 8 BLAS threads and 3.5340 with 3 or 5 (another summation order): this loss moves by 4e-3 under
    # rounding alone -- and by O(1) under a wrong logit, which is the point of the configuration
    assert rel < 1e-2 and relb < 1e-2
    rows = torch.stack([got[int(bi), int(j)] for bi, j in g["logits_rows_pos"]])
    worst, _ = compare_rows("sharp whole rows", rows, torch.from_numpy(g["logits_rows"]), 0.25)
    assert worst > 0.998                         # the oracle's bf16 mode: 0.9990 - 0.9997 on these rows
    lse = torch.logsumexp(got.double(), -1).float()
    tgt = got.gather(-1, out_ids[..., None])[..., 0]
    d_lse = float((lse - torch.from_numpy(g["logits_lse"])).abs().max())
    d_tgt = float((tgt - torch.from_numpy(g["logits_target"])).abs().max())
    d_max = float((got.max(-1).values - torch.from_numpy(g["logits_rowmax"])).abs().max())
    margin = torch.from_numpy(g["logits_margin"])
    same = got.argmax(-1) == torch.from_numpy(g["logits_argmax"])
    firm = margin > 0.25
    print(f"  per row (logit rms ~3): log-sum-exp within {d_lse:.4f}, target logit within {d_tgt:.4f}, maximum within {d_max:.4f}; arg-max equal on "
          f"{100 * float(same.float().mean()):.1f} % of the rows, {100 * float(same[firm].float().mean()):.1f} % of the {int(firm.sum())} with margin > 0.25")
    assert d_lse < 0.7 and d_tgt < 0.7 and d_max < 0.7       # the oracle's bf16 mode: log-sum-exp within 0.32 - 0.35 of the reference's
    assert float(same[firm].float().mean()) > 0.9
    loss.backward()
    grads = {k: p.grad.detach().float().cpu() for k, p in model.named_parameters()}
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    cs = sorted((cos(grad_sample(k[3:], grads[k[3:]].view(*shapes[k[3:]])), torch.from_numpy(g[k])), k[3:]) for k in g.files if k.startswith("gs:"))
    tot = float(torch.sqrt(sum((v.double() ** 2).sum() for v in grads.values())))
    print(f"  {len(cs)} sampled gradient tensors: worst cosines {[(round(c_, 4), n_) for c_, n_ in cs[:3]]}; total grad norm hip {tot:.3f} reference {float(g['grad_norm']):.3f}")
    assert cs[0][0] > 0.95 and abs(tot - float(g["grad_norm"])) < 3e-2 * float(g["grad_norm"])


def test_fused_head_logits_equal_unfused():
    """the Trainer's chunked head (LM head + CE + backward inside the forward) sees the same logits as the autograd route's whole-tensor head"""
    model = Vid2Seq("t5-base", tokenizer=SyntheticTokenizer(32100, 100), vis_drop=0.0, enc_drop=0.0, dec_drop=0.0, init_seed=11, device=DEV).train()
    b = {k: v.to(DEV) for k, v in synth.make_batch(4, 100, 200, 96, 32200, 11, 768).items()}
    eng = model.engine()
    eng.pack = False
    eng.head_rows = 128                      # three chunks
    eng.dbg_logits = []
    model(b["video"], tok(b["input_ids"]), tok(b["output_ids"]))        # (first: the step below renormalises the time-token rows of the tied embedding)
    whole = torch.cat(eng.dbg_logits, 0)
    eng.dbg_logits = []
    tr = Trainer(model, lr=0.0, clip_max_norm=1.0, generative=1.0, denoising=0.0)
    tr.step(b)
    fused = torch.cat(eng.dbg_logits, 0)
    n_chunks = len(eng.dbg_logits)
    eng.dbg_logits = None
    print(f"fused head: {n_chunks} chunks, {fused.shape[0]} rows; max |fused - whole| = {float((fused - whole).abs().max()):.2e}")
    assert n_chunks == 3 and fused.shape == whole.shape
    assert torch.equal(fused, whole) or float((fused - whole).abs().max()) < 1e-5
