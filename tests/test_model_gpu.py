"""End-to-end parity of the HIP engine (bf16 activations, fp32 accumulation) on the GPU against
  (a) golden vectors captured from the REAL reference (tests/golden/*.npz, made by oracle/make_golden.py), and
  (b) the CPU oracle (oracle/vid2seq_ref.py, itself pinned against the reference) on other seeded inputs.

Stated tolerances, set just under what the engine measures (round-2 numbers next to each assert): loss rel <= 2e-2;
greedy tokens equal up to a first divergence; gradients: cosine >= 0.978 per tensor on the reduced ("small") shapes and gradient-norm
ratio within [0.95, 1.06].  Why 0.978 and not 0.99:
on these tiny shapes (3 sequences x 24 tokens) the attention q/k/bias gradients are cancellation-dominated, and the
reference math itself run under torch's bf16 autocast scores 0.977-0.990 against its own fp32 gradients on exactly
these inputs (tests/tools/bf16_noise.py); the HIP path measures 0.980-0.995.  At full size (t5-base, cfg-1) the per-tensor
gradient norms are within 10 % and the total norm within 5 % of the reference.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vid2seq_ref as R            # noqa: E402  (checker only)
from vidchapters_amd import SyntheticTokenizer, Vid2Seq, synth  # noqa: E402
from vidchapters_amd.train import Trainer      # noqa: E402

DEV = "cuda"


def build(cfg: R.RefConfig, seed: int, **kw) -> Vid2Seq:
    t5 = dict(d_model=cfg.d_model, d_kv=cfg.d_kv, heads=cfg.heads, d_ff=cfg.d_ff, n_enc=cfg.n_enc, n_dec=cfg.n_dec)
    args = dict(num_features=cfg.num_features, embed_dim=cfg.vit_dim, depth=cfg.vit_depth, heads=cfg.vit_heads,
                mlp_dim=cfg.vit_mlp, tokenizer=SyntheticTokenizer(cfg.vocab - cfg.num_bins, cfg.num_bins), vis_drop=0.0,
                enc_drop=0.0, dec_drop=0.0, num_bins=cfg.num_bins, label_smoothing=cfg.label_smoothing, init_seed=seed)
    args.update(kw)
    return Vid2Seq(t5, **args).to(DEV)


def tok(ids):
    ids = ids.to(DEV)
    return {"input_ids": ids, "attention_mask": ids != 0}


def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float(a @ b / (a.norm() * b.norm() + 1e-30))


def named_grads(model):
    return {k: p.grad.detach().float().cpu() for k, p in model.named_parameters()}


@pytest.mark.parametrize("tag,cfg,seed", [
    ("small", R.RefConfig.small(), 7),
    ("small_resize_proj", R.RefConfig.small(vit_dim=64, vit_heads=1, num_features=10), 9),
])
def test_forward_backward_vs_reference_golden(golden_dir, tag, cfg, seed):
    g = np.load(os.path.join(golden_dir, f"{tag}_forward_backward.npz"))
    model = build(cfg, seed).eval()
    video = torch.from_numpy(g["video"]).to(DEV)
    out, vd = model(video, tok(torch.from_numpy(g["input_ids"])), tok(torch.from_numpy(g["output_ids"])))
    loss = out["loss"]
    ref = float(g["loss"])
    print(f"[{tag}] loss hip={loss.item():.6f} reference={ref:.6f}")
    assert abs(loss.item() - ref) <= 2e-2 * abs(ref)
    mem = torch.from_numpy(g["memory"])
    T = video.shape[1]
    assert vd["video"].dtype == torch.float32 and vd["atts_vis"].dtype == torch.long          # the reference's video_dict (vid2seq.py:60-67)
    c = cos(vd["video"].float().cpu(), mem[:, :T])
    print(f"  ViT output cosine vs reference: {c:.6f}")
    assert c > 0.999
    loss.backward()
    grads = named_grads(model)
    worst, bad = 1.0, []
    for k in g.files:
        if not k.startswith("grad:"):
            continue
        name = k[5:]
        want = torch.from_numpy(g[k])
        got = grads[name].view_as(want)
        cs = cos(got, want)
        rn = float(got.norm() / (want.norm() + 1e-30))
        worst = min(worst, cs) if cs == cs else float("nan")
        if not (cs > 0.995):
            print(f"  {name}: cos {cs:.4f} norm ratio {rn:.3f} finite={bool(torch.isfinite(got).all())}")
        if not (cs > 0.978 and 0.95 < rn < 1.06):            # measured r02: worst cosine 0.9803 / 0.9889, norm ratios 0.973 .. 1.039
            bad.append((name, cs, rn))
    print(f"  worst gradient cosine over all tensors: {worst:.5f}; failing tensors: {len(bad)}")
    assert not bad, bad[:10]


@pytest.mark.parametrize("use_video,use_speech", [(True, False), (False, True)])
def test_ablation_flags_vs_oracle(use_video, use_speech):
    cfg = R.RefConfig.small(use_video=use_video, use_speech=use_speech)
    model = build(cfg, 21, use_video=use_video, use_speech=use_speech).eval()
    P = synth.init_params(R.param_shapes(cfg), 21, cfg.d_model, cfg.inner, cfg.d_ff)
    b = synth.make_batch(2, 10, 40, 17, cfg.vocab, 33, cfg.vit_dim)
    with torch.no_grad():
        want, _ = R.vid2seq_forward(P, cfg, b["video"], b["input_ids"], b["input_ids"] != 0, b["output_ids"], b["output_ids"] != 0)
        got, _ = model(b["video"].to(DEV), tok(b["input_ids"]), tok(b["output_ids"]))
    print(f"ablation video={use_video} speech={use_speech}: hip {got['loss'].item():.5f} oracle {want['loss'].item():.5f}")
    assert abs(got["loss"].item() - want["loss"].item()) <= 2e-2 * abs(want["loss"].item())


def test_vc_variant_num_bins_zero_vs_oracle():
    """vc.py's use of the same module: no time tokens (num_bins=0, vocab = text tokens only), single pass, and an evaluation
    batch made of the chapters of ONE video (vc.py:140-154: video[0] -> [n_chap, T, 768])."""
    cfg = R.RefConfig.small(vocab=512, num_bins=0)
    model = build(cfg, 23, tokenizer=SyntheticTokenizer(512, 0)).eval()
    P = synth.init_params(R.param_shapes(cfg), 23, cfg.d_model, cfg.inner, cfg.d_ff)
    b = synth.make_batch(3, 10, 30, 12, cfg.vocab, 35, cfg.vit_dim)
    video = b["video"][:1].expand(3, -1, -1).contiguous()          # chapters of one video share its features
    with torch.no_grad():
        want, _ = R.vid2seq_forward(P, cfg, video, b["input_ids"], b["input_ids"] != 0, b["output_ids"], b["output_ids"] != 0)
        got, _ = model(video.to(DEV), tok(b["input_ids"]), tok(b["output_ids"]))
    assert abs(got["loss"].item() - want["loss"].item()) <= 2e-2 * abs(want["loss"].item())
    toks = model.engine().greedy(video.to(DEV), tok(b["input_ids"]), max_new_tokens=6).cpu()
    ref = R.greedy_generate(P, cfg, video, b["input_ids"], b["input_ids"] != 0, 6)
    n = min(toks.shape[1], ref.shape[1])
    assert (toks[:, :n] == ref[:, :n]).float().mean() > 0.8



def test_ragged_batch_and_two_pass_vs_oracle():
    """Ragged lengths (a 1-token row, a full row, >64 and >128 keys so several attention tiles are partly masked) and
    the cached video_dict path of dvc.py:78-92."""
    cfg = R.RefConfig.small()
    model = build(cfg, 11).eval()
    P = synth.init_params(R.param_shapes(cfg), 11, cfg.d_model, cfg.inner, cfg.d_ff)
    for v in P.values():
        v.requires_grad_(True)
    b = synth.make_batch(3, 10, 150, 70, cfg.vocab, 5, cfg.vit_dim, denoising=True)
    b["input_ids"][0, 1:] = 0; b["input_ids"][0, 0] = 1
    b["output_ids"][1, 1:] = 0; b["output_ids"][1, 0] = 1
    o1, vd = R.vid2seq_forward(P, cfg, b["video"], b["input_ids"], b["input_ids"] != 0, b["output_ids"], b["output_ids"] != 0)
    o2, _ = R.vid2seq_forward(P, cfg, vd, b["den_input_ids"], b["den_input_ids"] != 0, b["den_output_ids"], b["den_output_ids"] != 0)
    tot = o1["loss"] + o2["loss"]
    names = list(P)
    gw = torch.autograd.grad(tot, [P[k] for k in names])
    l1, vdict = model(b["video"].to(DEV), tok(b["input_ids"]), tok(b["output_ids"]))
    l2, _ = model(vdict, tok(b["den_input_ids"]), tok(b["den_output_ids"]))
    (l1["loss"] + l2["loss"]).backward()
    print(f"two-pass losses hip ({l1['loss'].item():.5f}, {l2['loss'].item():.5f}) oracle ({o1['loss'].item():.5f}, {o2['loss'].item():.5f})")
    assert abs(l1["loss"].item() - o1["loss"].item()) <= 2e-2 * o1["loss"].item()
    assert abs(l2["loss"].item() - o2["loss"].item()) <= 2e-2 * o2["loss"].item()
    grads = named_grads(model)
    for k, w in zip(names, gw):
        cs = cos(grads[k], w)
        assert cs > 0.975, (k, cs)


def test_train_recipe_vs_reference_golden(golden_dir):
    """dvc.py:train_one_epoch x2 steps (captured from the real reference) vs Trainer.step x2."""
    g = np.load(os.path.join(golden_dir, "small_train_recipe.npz"))
    cfg = R.RefConfig.small()
    model = build(cfg, 5).train()
    pre = {k: p.detach().float().cpu().clone() for k, p in model.named_parameters()}
    tr = Trainer(model, lr=3e-4, clip_max_norm=0.1, generative=1.0, denoising=1.0)
    for i in range(2):
        batch = {k.split(":", 1)[1]: torch.from_numpy(g[k]).to(DEV) for k in g.files if k.startswith(f"b{i}:")}
        losses = tr.step(batch)
        if i == 0:
            l0, d0, gn0 = losses["loss"].item(), losses["denoising_loss"].item(), tr.grad_norm().item()
            print(f"step0 loss {l0:.5f}/{float(g['loss0']):.5f} den {d0:.5f}/{float(g['den0']):.5f} gnorm {gn0:.4f}/{float(g['gnorm0']):.4f}")
            assert abs(l0 - float(g["loss0"])) <= 2e-2 * float(g["loss0"])
            assert abs(d0 - float(g["den0"])) <= 2e-2 * float(g["den0"])
            assert abs(gn0 - float(g["gnorm0"])) <= 5e-2 * float(g["gnorm0"])
    post = {k: p.detach().float().cpu() for k, p in model.named_parameters()}
    for k in g.files:
        if not k.startswith("post:"):
            continue
        name = k[5:]
        want = torch.from_numpy(g[k]).view_as(post[name])
        du_ref, du = want - pre[name], post[name] - pre[name]
        cs = cos(du, du_ref)
        mx = float((post[name] - want).abs().max())
        print(f"  {name}: update cosine {cs:.4f}, max|w - w_ref| {mx:.2e}")
        # Adam's first steps move every weight by ~lr*sign(g): elements whose tiny gradient changes sign under bf16
        # rounding differ by up to 2*lr per step, so compare the update direction and bound the distance by 2 steps * 2 lr
        assert cs > 0.93 and mx <= 4.2 * 3e-4 + 1e-6, (name, cs, mx)      # measured r02: update cosine 0.943 .. 0.9998
    # the bf16 shadow tracks the fp32 master after the fused step
    eng = model.engine()
    assert torch.equal(eng.arena.shadow, eng.arena.master.to(torch.bfloat16))


def test_dropin_optimizer_path_matches_trainer():
    """The reference's caller idiom: loss.backward(); clip_grad_norm_; torch.optim.Adam.step(); zero_grad()."""
    cfg = R.RefConfig.small()
    b = synth.make_batch(2, 10, 33, 12, cfg.vocab, 3, cfg.vit_dim)
    m1, m2 = build(cfg, 4).train(), build(cfg, 4).train()
    opt = torch.optim.Adam(m1.parameters(), lr=3e-4)
    for _ in range(2):
        opt.zero_grad()
        out, _ = m1(b["video"].to(DEV), tok(b["input_ids"]), tok(b["output_ids"]))
        out["loss"].backward()
        torch.nn.utils.clip_grad_norm_(m1.parameters(), 1.0)
        opt.step()
    tr = Trainer(m2, lr=3e-4, clip_max_norm=1.0, generative=1.0, denoising=0.0)
    m2.num_bins = 0                      # no renorm, like the plain optimizer loop above
    for _ in range(2):
        tr.step({k: v.to(DEV) for k, v in b.items()})
    # same kernels, same inputs: the two paths differ only by the order of fp32 atomics (embedding scatter, norm-weight
    # and bias column sums, split-K order), which Adam's sign-like first steps amplify on near-zero-gradient elements:
    # compare the update direction, bound the distance by 2 steps x 2 lr
    init = build(cfg, 4)
    for (k, p1), (_, p2), (_, p0) in zip(m1.named_parameters(), m2.named_parameters(), init.named_parameters()):
        u1, u2 = (p1.detach() - p0.detach().to(DEV)).flatten(), (p2.detach() - p0.detach().to(DEV)).flatten()
        assert (u1 - u2).abs().max().item() <= 4.1 * 3e-4, k
        assert cos(u1, u2) > 0.95, (k, cos(u1, u2))      # noise-dominated tensors (e.g. the k-part of a qkv bias, whose true gradient is 0)


def test_greedy_vs_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "small_greedy.npz"))
    cfg = R.RefConfig.small()
    model = build(cfg, 7).eval()
    toks = model.engine().greedy(torch.from_numpy(g["video"]).to(DEV), tok(torch.from_numpy(g["input_ids"])),
                                 max_new_tokens=int(g["max_new"])).cpu()
    want = torch.from_numpy(g["tokens"])
    n = min(toks.shape[1], want.shape[1])
    same = (toks[:, :n] == want[:, :n])
    first_div = [int((~r).nonzero()[0]) if (~r).any() else n for r in same]
    print("greedy first divergence per row:", first_div, "of", n)
    assert toks[:, 0].eq(0).all()
    assert min(first_div) >= 16, (toks, want)              # measured r02: [17, 25, 25] of 25 (bf16 argmax flips only on <1e-2 logit margins)
    text = model.generate(torch.from_numpy(g["video"]).to(DEV), tok(torch.from_numpy(g["input_ids"])), num_beams=1,
                          max_length=int(g["max_new"]))
    assert isinstance(text, list) and len(text) == want.shape[0] and all(isinstance(t, str) for t in text)


@pytest.mark.parametrize("nb", [2, 3, 5, 8, 12, 20])
def test_beam_search_other_widths(golden_dir, nb):
    """num_beams other than the default 4: 2 and 8 take the one-block-per-entry cross-attention (kv_group), 3, 5 and 12 one block per
    beam row (12: the 32-candidate top-k; 20: more than 16 beams -- the generic K = 2*nb rounds top-k, one cross-attention block per
    beam row; the reference forwards any num_beams, vid2seq.py:150-162); all of them the row-map instead of a cache reorder.  Reference = the fp32 oracle run here on the fixture's inputs; same bar as
    the golden test (valid hypotheses, >= 3/4 of the rows identical: bf16 logits can flip a near-tie)."""
    g = np.load(os.path.join(golden_dir, "small_beam.npz"))
    cfg = R.RefConfig.small()
    max_new = int(g["max_new"])
    rows = same = 0
    for i in range(min(2, len(g["seed"]))):
        model = build(cfg, int(g["seed"][i])).eval()
        with torch.no_grad():
            E = model.t5_model.shared.weight
            E.mul_(6.0)
            E[1] = E[int(g["fav"][i])] * float(g["fac"][i])
        P = synth.init_params(R.param_shapes(cfg), int(g["seed"][i]), cfg.d_model, cfg.inner, cfg.d_ff)
        Ew = P["t5_model.shared.weight"] * 6.0
        Ew[1] = Ew[int(g["fav"][i])] * float(g["fac"][i])
        P["t5_model.shared.weight"] = Ew
        video, ids = torch.from_numpy(g["video"][i]), torch.from_numpy(g["input_ids"][i])
        want = R.beam_generate(P, cfg, video, ids, ids != 0, nb, max_new, 1.0)
        out = model.engine().beam_search(video.to(DEV), tok(ids), num_beams=nb, max_new_tokens=max_new, length_penalty=1.0).cpu()
        assert out[:, 0].eq(0).all() and out.shape[1] <= max_new + 1
        for r in range(out.shape[0]):
            o = out[r].tolist()
            if 1 in o:
                assert all(t == 0 for t in o[o.index(1) + 1:])
            w = want[r].tolist()          # (the oracle returns a tensor [B, len])
            n = max(len(o), len(w))
            rows += 1
            same += (o + [0] * (n - len(o))) == (w + [0] * (n - len(w)))
    assert same >= 0.75 * rows, (nb, same, rows)


def test_beam_search_vs_golden(golden_dir):
    """num_beams=4 (vid2seq.py:150-162 default).  Fixture = oracle outputs that agree with the installed transformers'
    generate (the 4.28 scorer itself is un-vendored: parity unpinned).  bf16 logits can flip a near-tie between two beams, so the
    bar is: every returned row is a valid hypothesis (start token, pad after EOS), and at least 3/4 of all rows are identical to
    the fp32 fixture token for token."""
    g = np.load(os.path.join(golden_dir, "small_beam.npz"))
    cfg = R.RefConfig.small()
    nb, max_new = int(g["num_beams"]), int(g["max_new"])
    rows = same = 0
    for i in range(len(g["seed"])):
        model = build(cfg, int(g["seed"][i])).eval()
        with torch.no_grad():
            E = model.t5_model.shared.weight
            E.mul_(6.0)
            E[1] = E[int(g["fav"][i])] * float(g["fac"][i])
        video, ids = torch.from_numpy(g["video"][i]).to(DEV), torch.from_numpy(g["input_ids"][i])
        want = torch.from_numpy(g["tokens"][i])
        for use_graph in (True, False):
            out = model.engine().beam_search(video, tok(ids), num_beams=nb, max_new_tokens=max_new, length_penalty=1.0,
                                             use_graph=use_graph).cpu()
            assert out[:, 0].eq(0).all() and out.shape[1] <= max_new + 1
            if use_graph:
                first = out
            else:
                assert torch.equal(out, first)                     # graph replay == eager launches
        for r in range(out.shape[0]):
            o = out[r].tolist()
            if 1 in o:
                assert all(t == 0 for t in o[o.index(1) + 1:])
            w = want[r, :out.shape[1]].tolist()
            rows += 1
            same += (o == w and int(want[r, out.shape[1]:].abs().sum()) == 0)
    # min_length (MinLengthLogitsProcessor): EOS banned until the decoder sequence has min_length tokens
    P = synth.init_params(R.param_shapes(cfg), int(g["seed"][i]), cfg.d_model, cfg.inner, cfg.d_ff)
    Ew = P["t5_model.shared.weight"] * 6.0
    Ew[1] = Ew[int(g["fav"][i])] * float(g["fac"][i])
    P["t5_model.shared.weight"] = Ew
    want_ml = R.beam_generate(P, cfg, video.cpu(), ids, ids != 0, nb, max_new, 1.0, min_length=7)
    got_ml = model.engine().beam_search(video, tok(ids), num_beams=nb, max_new_tokens=max_new, min_length=7).cpu()
    assert all(1 not in row[:6] for row in got_ml.tolist())
    print("min_length=7:", "identical" if torch.equal(got_ml, want_ml) else (got_ml.tolist(), want_ml.tolist()))
    assert got_ml.shape == want_ml.shape and (got_ml == want_ml).float().mean() > 0.9
    print(f"beam search rows identical to the fp32 fixture: {same}/{rows}")
    assert same >= rows - 1                                # measured r02: 24/24
    text = model.generate(video, tok(ids), num_beams=nb, max_length=max_new)
    assert isinstance(text, list) and len(text) == out.shape[0]
    text2 = model.generate(video, tok(ids), num_beams=nb, max_length=max_new, num_captions=2)     # num_return_sequences = 2
    assert len(text2) == 2 * out.shape[0] and text2[0::2] == text


@pytest.mark.parametrize("nb,kw", [(4, {}), (4, {"length_penalty": 0.6}), (4, {"repetition_penalty": 1.3}), (4, {"min_length": 7, "num_return": 3}),
                                   (2, {}), (3, {"num_return": 2}), (8, {}), (12, {"length_penalty": 2.0})])
def test_beam_search_device_scorer_equals_host_scorer(golden_dir, nb, kw):
    """The beam bookkeeping on the device (v2s_beam_advance inside the replayed graph, no host round trip per step) returns the very
    tokens of the host scorer (vidchapters_amd/beam.py, the restatement of transformers 4.28's BeamSearchScorer that the golden and
    oracle tests pin): same kernels before it, so the comparison is exact -- EOS-heavy fixture (entries finish at different steps),
    length / repetition penalties, min_length, several returned sequences, widths with and without the grouped cross-attention."""
    g = np.load(os.path.join(golden_dir, "small_beam.npz"))
    cfg = R.RefConfig.small()
    max_new = int(g["max_new"])
    for i in range(min(3, len(g["seed"]))):
        model = build(cfg, int(g["seed"][i])).eval()
        with torch.no_grad():
            E = model.t5_model.shared.weight
            E.mul_(6.0)
            E[1] = E[int(g["fav"][i])] * float(g["fac"][i])
        video, ids = torch.from_numpy(g["video"][i]).to(DEV), torch.from_numpy(g["input_ids"][i])
        eng = model.engine()
        outs = {}
        for dev in (True, False):
            eng.beam_on_device = dev
            outs[dev] = eng.beam_search(video, tok(ids), num_beams=nb, max_new_tokens=max_new, **kw).cpu()
        eng.beam_on_device = True
        assert torch.equal(outs[True], outs[False]), (nb, kw, i, outs[True].tolist(), outs[False].tolist())
        eager = eng.beam_search(video, tok(ids), num_beams=nb, max_new_tokens=max_new, use_graph=False, **kw).cpu()
        assert torch.equal(eager, outs[True])



def test_checkpoint_load_after_engine_build_takes_effect():
    """load_state_dict (dvc.py:354-361) into a model whose arena / bf16 shadow already exist: the next forward must see the
    loaded weights (the shadow refresh is keyed on the parameters' version counters)."""
    cfg = R.RefConfig.small()
    b = synth.make_batch(2, 10, 40, 17, cfg.vocab, 9, cfg.vit_dim)
    args = (b["video"].to(DEV), tok(b["input_ids"]), tok(b["output_ids"]))
    a = build(cfg, 41).eval()
    ref = build(cfg, 42).eval()
    with torch.no_grad():
        la = a(*args)[0]["loss"].item()
        want = ref(*args)[0]["loss"].item()
        ck = {k: v.detach().cpu().clone() for k, v in ref.state_dict().items()}
        res = a.load_state_dict(ck, strict=False)
        assert not res.missing_keys and not res.unexpected_keys
        got = a(*args)[0]["loss"].item()
    assert la != want and got == want



def test_dropout_training_mode_runs_and_differs():
    cfg = R.RefConfig.small()
    model = build(cfg, 8, vis_drop=0.1, enc_drop=0.1, dec_drop=0.1)
    b = synth.make_batch(2, 10, 40, 17, cfg.vocab, 9, cfg.vit_dim)
    model.eval()
    with torch.no_grad():
        l_eval = model(b["video"].to(DEV), tok(b["input_ids"]), tok(b["output_ids"]))[0]["loss"].item()
    model.train()
    out, _ = model(b["video"].to(DEV), tok(b["input_ids"]), tok(b["output_ids"]))
    out["loss"].backward()
    l_tr = out["loss"].item()
    print(f"dropout: eval loss {l_eval:.5f} train loss {l_tr:.5f}")
    assert np.isfinite(l_tr) and l_tr != l_eval and abs(l_tr - l_eval) < 0.5
    for k, p in model.named_parameters():
        assert torch.isfinite(p.grad).all(), k


def test_full_size_cfg1_vs_reference_golden(golden_dir):
    """t5-base Vid2Seq (289 M parameters), B=2, T=100, L=256, Lo=256: loss, logits-free scalars and gradient norms
    captured from the real reference in fp32."""
    g = np.load(os.path.join(golden_dir, "full_cfg1_scalars.npz"))
    seed, B, Lx, Lo = int(g["seed"]), int(g["B"]), int(g["L"]), int(g["Lo"])
    model = Vid2Seq("t5-base", tokenizer=SyntheticTokenizer(32100, 100), vis_drop=0.0, enc_drop=0.0, dec_drop=0.0,
                    init_seed=seed, device=DEV).eval()
    b = synth.make_batch(B, 100, Lx, Lo, 32200, seed, 768)
    out, vd = model(b["video"].to(DEV), tok(b["input_ids"]), tok(b["output_ids"]))
    ref = float(g["loss"])
    print(f"full cfg-1 loss hip={out['loss'].item():.5f} reference={ref:.5f}")
    assert abs(out["loss"].item() - ref) <= 2e-2 * ref
    ms = torch.from_numpy(g["memory_slice"])
    got = vd["video"].float().cpu()[:, ::37, :32]
    assert cos(got, ms[:, :got.shape[1]]) > 0.999
    out["loss"].backward()
    keys = [str(k) for k in g["grad_norm_keys"]]
    vals = g["grad_norm_vals"]
    grads = {k: p.grad for k, p in model.named_parameters()}
    nonfinite = [k for k, v in grads.items() if not torch.isfinite(v).all()]
    print(f"  non-finite gradient tensors: {len(nonfinite)} {nonfinite[:12]}")
    tot = float(torch.sqrt(sum((v.float() ** 2).sum() for v in grads.values())))
    print(f"  total grad norm hip={tot:.4f} reference={float(g['grad_norm']):.4f}")
    assert abs(tot - float(g["grad_norm"])) <= 2e-2 * float(g["grad_norm"])      # measured r02: 0.7 %
    bad = []
    for k, v in zip(keys, vals):
        r = float(grads[k].float().norm()) / (float(v) + 1e-12)
        if not 0.9 < r < 1.1:
            bad.append((k, r))
    print(f"  per-tensor grad-norm ratio outside [0.9,1.1]: {bad[:8]}")
    assert not bad                                          # measured r02: every tensor inside [0.9, 1.1]


def test_full_size_cfg2_size_independent_properties():
    """BASELINE cfg-2 shapes (t5-base, 100 frames, 1000 ASR tokens, 256 target tokens) where the fp32 oracle is too slow to run
    in a test: properties that hold for the reference at any size.
      (1) token-weighted decomposition: loss(batch) * n_tokens(batch) == sum_i loss(sample i) * n_tokens(i)   (mean CE over
          non-ignored targets, modeling_t5.py:1721);
      (2) padding invariance: extra all-pad columns on the speech and target ids change nothing (masks = ids != 0);
      (3) the cached video_dict (dvc.py:78-92) gives the same loss as re-encoding the frames;
      (4) no speech (dvc.py:47-50: a single EOS token per row) runs and is finite."""
    model = Vid2Seq("t5-base", tokenizer=SyntheticTokenizer(32100, 100), vis_drop=0.0, enc_drop=0.0, dec_drop=0.0,
                    init_seed=5, device=DEV).eval()
    B = 4
    b = synth.make_batch(B, 100, 1000, 256, 32200, 77, 768)
    video, ids, out = b["video"].to(DEV), b["input_ids"], b["output_ids"]
    with torch.no_grad():
        full, vd = model(video, tok(ids), tok(out))
        full = full["loss"].item()
        n_tok = [(out[i] != 0).sum().item() for i in range(B)]
        per = [model(video[i:i + 1], tok(ids[i:i + 1]), tok(out[i:i + 1]))[0]["loss"].item() for i in range(B)]
        recomposed = sum(l * n for l, n in zip(per, n_tok)) / sum(n_tok)
        print(f"cfg-2 loss {full:.6f}; recomposed from per-sample losses {recomposed:.6f}")
        assert abs(full - recomposed) <= 2e-3 * abs(full)
        ids_p = torch.cat([ids, torch.zeros(B, 24, dtype=ids.dtype)], 1)
        out_p = torch.cat([out, torch.zeros(B, 8, dtype=out.dtype)], 1)
        padded = model(video, tok(ids_p), tok(out_p))[0]["loss"].item()
        print(f"  with 24/8 extra pad columns {padded:.6f}")
        assert abs(padded - full) <= 2e-3 * abs(full)
        cached = model(vd, tok(ids), tok(out))[0]["loss"].item()
        assert abs(cached - full) <= 1e-5 * abs(full)
        eos_only = torch.ones(B, 1, dtype=ids.dtype)
        nospeech = model(video, tok(eos_only), tok(out))[0]["loss"].item()
        assert np.isfinite(nospeech) and abs(nospeech - full) > 0


def test_padding_free_encoder_is_exact():
    """Engine.pack / Engine.pack_dec: the text encoder runs on the non-pad tokens only, the decoder on the rows of real targets only
    (q-packed cross-attention), the [video ; text] memory without its pad rows (key-packed cross-attention).  Loss and every parameter gradient must equal the dense (reference-like) path up
    to the order of floating-point accumulation in the weight-gradient GEMMs."""
    cfg = R.RefConfig.small()
    b = synth.make_batch(4, 10, 150, 40, cfg.vocab, 19, cfg.vit_dim)
    b["input_ids"][2, 5:] = 0                      # a short row; row 0..3 have different lengths
    b["output_ids"][1, 9:] = 0; b["output_ids"][1, 8] = 1            # short target rows: the decoder plan must find room for its filler rows
    b["output_ids"][3, 15:] = 0; b["output_ids"][3, 14] = 1
    args = (b["video"].to(DEV), tok(b["input_ids"]), tok(b["output_ids"]))
    res = {}
    for mode in ("enc+dec+mem", "enc+dec", "enc", "dense"):
        model = build(cfg, 31).eval()
        eng = model.engine()
        eng.pack, eng.pack_dec, eng.pack_mem = mode != "dense", mode.startswith("enc+dec"), mode == "enc+dec+mem"
        if mode == "enc+dec":
            plan = eng._pack_plan_dec(b["output_ids"].to(DEV) != 0)
            assert plan is not None and plan[1] % 64 == 0 and plan[1] < 4 * 40, plan
        out, _ = model(*args)
        out["loss"].backward()
        res[mode] = (out["loss"].item(), named_grads(model))
    ld, gd = res["dense"]
    for mode in ("enc+dec+mem", "enc+dec", "enc"):
        lp, gp = res[mode]
        print(f"loss packed ({mode}) {lp:.7f} dense {ld:.7f}")
        assert abs(lp - ld) <= 1e-5 * abs(ld)
        worst = min(cos(gp[k], gd[k]) for k in gp if gd[k].abs().max() > 0)
        print(f"  worst gradient cosine packed ({mode}) vs dense: {worst:.6f}")
        assert worst > 0.9995
    # generate() goes through the same encoder
    model.engine().pack = True
    t1 = model.engine().greedy(args[0], args[1], max_new_tokens=8).cpu()
    model.engine().pack = False
    t2 = model.engine().greedy(args[0], args[1], max_new_tokens=8).cpu()
    n = min(t1.shape[1], t2.shape[1])
    assert torch.equal(t1[:, :n], t2[:, :n])


def test_repetition_penalty_vs_golden(golden_dir):
    """generate(repetition_penalty=1.3) for greedy and 4 beams; fixture = oracle outputs that agree with the installed transformers'
    generate (the 4.28 processor is un-vendored: parity unpinned)."""
    g = np.load(os.path.join(golden_dir, "small_repetition_penalty.npz"))
    cfg = R.RefConfig.small()
    pen, max_new = float(g["penalty"]), int(g["max_new"])
    rows = same = 0
    for i in range(int(g["n"])):
        seed, fav = (int(x) for x in g[f"meta_{i}"])
        model = build(cfg, seed).eval()
        with torch.no_grad():
            E = model.t5_model.shared.weight
            E.mul_(6.0)
            E[1] = E[fav] * float(g[f"fac_{i}"])
        video, ids = torch.from_numpy(g[f"video_{i}"]).to(DEV), torch.from_numpy(g[f"ids_{i}"])
        for nb in (1, 4):
            want = torch.from_numpy(g[f"tok_{i}_{nb}"])
            if nb == 1:
                out = model.engine().greedy(video, tok(ids), max_new_tokens=max_new, repetition_penalty=pen).cpu()
            else:
                out = model.engine().beam_search(video, tok(ids), num_beams=nb, max_new_tokens=max_new, repetition_penalty=pen).cpu()
            for r in range(out.shape[0]):
                rows += 1
                same += (out[r].tolist() == want[r, :out.shape[1]].tolist() and int(want[r, out.shape[1]:].abs().sum()) == 0)
    print(f"repetition penalty: rows identical to the fp32 fixture: {same}/{rows}")
    assert same >= rows - 1                                # measured r02: 16/16


def test_generate_with_nucleus_sampling_runs():
    """generate(use_nucleus_sampling=True) (dvc.py:177 with --num_beams 0): valid sequences, different draws for different calls, and with
    a near-deterministic distribution (top_p tiny) the same tokens as greedy decoding."""
    cfg = R.RefConfig.small()
    model = build(cfg, 7).eval()
    b = synth.make_batch(3, 10, 24, 12, cfg.vocab, 7, cfg.vit_dim)
    video, ids = b["video"].to(DEV), tok(b["input_ids"])
    t1 = model.generate(video, ids, use_nucleus_sampling=True, num_beams=0, max_length=12, top_p=0.9)
    t2 = model.generate(video, ids, use_nucleus_sampling=True, num_beams=0, max_length=12, top_p=0.9)
    assert len(t1) == len(t2) == 3 and all(isinstance(x, str) for x in t1) and t1 != t2
    greedy = model.engine().greedy(video, ids, max_new_tokens=12).cpu()
    samp = model.engine().greedy(video, ids, max_new_tokens=12, sample=(1e-6, 1.0, 5)).cpu()      # nucleus = the argmax token only
    n = min(greedy.shape[1], samp.shape[1])
    assert torch.equal(greedy[:, :n], samp[:, :n])


def test_generate_beam_sample_vs_oracle():
    """generate(use_nucleus_sampling=True, num_beams=4) = HF 4.28 beam_sample (vid2seq.py:150-162 forwards do_sample together with
    num_beams): strings come back, a call is reproducible for a given sampling seed and differs for another, num_captions expands the
    rows; and, step by step along the engine's trajectory, the oracle's restatement of beam_sample (warpers + draws without replacement
    as Gumbel keys, the kernel's noise restated on the host) reproduces the beam scores and the final sequences."""
    from vidchapters_amd.beam import BeamScorer
    cfg = R.RefConfig.small()
    model = build(cfg, 21).eval()
    P = synth.init_params(R.param_shapes(cfg), 21, cfg.d_model, cfg.inner, cfg.d_ff)
    b = synth.make_batch(4, cfg.num_features, 24, 12, cfg.vocab, 21, cfg.vit_dim)
    video, ids = b["video"].to(DEV), tok(b["input_ids"])
    nb, max_new, top_p, temp, top_k = 4, 10, 0.9, 0.8, 50
    model.sampling_seed = 100
    t1 = model.generate(video, ids, use_nucleus_sampling=True, num_beams=nb, max_length=max_new, top_p=top_p, temperature=temp)
    model.sampling_seed = 100
    t2 = model.generate(video, ids, use_nucleus_sampling=True, num_beams=nb, max_length=max_new, top_p=top_p, temperature=temp)
    t3 = model.generate(video, ids, use_nucleus_sampling=True, num_beams=nb, max_length=max_new, top_p=top_p, temperature=temp)
    assert len(t1) == 4 and all(isinstance(x, str) for x in t1) and t1 == t2 and t1 != t3
    caps = model.generate(video, ids, use_nucleus_sampling=True, num_beams=2, max_length=6, num_captions=2)
    assert len(caps) == 8
    # Along the engine's own trajectory (a free-running fp32 oracle diverges at the first near-tied key, and HF's accumulation of
    # temperature-scaled scores amplifies any difference by 1/T per step): record the logits / beam scores / step counter every step
    # hands to the kernel, recompute that step with the oracle (warpers, keys from the restated noise) and drive a second host scorer
    # with it -- beam scores of every following step, and the final sequences, must agree.
    import vidchapters_amd.engine as E
    seed, rec, orig = 555, [], E.L.beam_sample_cand

    def spy(logits, ld, rows, V, K, bscore, *a, **kw):
        rec.append((logits.view(-1, ld)[:rows, :V].float().cpu().clone(), bscore.cpu().clone(), int(kw["pos_dev"].item())))
        return orig(logits, ld, rows, V, K, bscore, *a, **kw)
    E.L.beam_sample_cand = spy
    try:
        got = model.engine().beam_search(video, ids, num_beams=nb, max_new_tokens=max_new, sample=(top_p, temp, seed, top_k), use_graph=False).cpu()
    finally:
        E.L.beam_sample_cand = orig
    B = 4
    sc = BeamScorer(B, nb, 1.0, cfg.eos_id, cfg.pad_id, cfg.dec_start_id, max_new + 1, sample=True)
    assert [r[2] for r in rec] == list(range(len(rec)))
    for t, (logits, bscore, pos) in enumerate(rec):
        assert np.abs(bscore.numpy() - sc.scores.reshape(-1)).max() < 1e-3, t
        noise = R.beam_sample_gumbel(seed, pos, B * nb, logits.shape[1])
        w = R.warp_scores(torch.log_softmax(logits, -1) + bscore[:, None], top_p, temp, top_k, 2)
        kk, ki = torch.topk(w + noise, 2 * nb, dim=1)
        _, _, finished = sc.advance(torch.gather(w, 1, ki).numpy(), ki.numpy().astype(np.int32), kk.numpy())
        assert finished == (t == len(rec) - 1)
    want = torch.from_numpy(sc.finalize(1))
    print(f"beam-sample: {len(rec)} steps; hip {got[0].tolist()} oracle {want[0].tolist()}")
    assert got.shape == want.shape and torch.equal(got, want)


def test_greedy_min_length_and_sampled_num_captions_vs_oracle():
    """generate(num_beams=1, min_length=k) (EOS banned by v2s_ban_token from the device step counter, inside the replayed graph) against
    the oracle's greedy loop (itself checked against the installed transformers), and num_captions > 1 with nucleus sampling (HF expands
    every input row num_return_sequences times)."""
    cfg = R.RefConfig.small()
    model = build(cfg, 40).eval()
    P = synth.init_params(R.param_shapes(cfg), 40, cfg.d_model, cfg.inner, cfg.d_ff)
    b = synth.make_batch(4, cfg.num_features, 24, 12, cfg.vocab, 40, cfg.vit_dim)
    Ew = P["t5_model.shared.weight"] * 6.0
    P["t5_model.shared.weight"] = Ew
    g0 = R.greedy_generate(P, cfg, b["video"], b["input_ids"], b["input_ids"] != 0, 6)
    fav = int(torch.mode(g0[:, 1:].flatten()).values)
    Ew[1] = Ew[fav] * 1.3
    with torch.no_grad():
        model.t5_model.shared.weight.copy_(Ew.to(DEV))
    video, ids = b["video"].to(DEV), tok(b["input_ids"])
    for ml in (1, 5, 9):
        want = R.greedy_generate(P, cfg, b["video"], b["input_ids"], b["input_ids"] != 0, 12, min_length=ml)
        got = model.engine().greedy(video, ids, max_new_tokens=12, min_length=ml).cpu()
        n = min(got.shape[1], want.shape[1])
        print(f"min_length={ml}: hip {got[0].tolist()} oracle {want[0].tolist()}")
        assert (got[:, :n] == want[:, :n]).float().mean() > 0.9
        if ml > 1:
            assert not (got[:, 1:ml - 1] == 1).any()
    text = model.generate(video, ids, num_beams=1, max_length=12, min_length=5)
    assert len(text) == 4
    caps = model.generate(video, ids, use_nucleus_sampling=True, num_beams=0, max_length=8, top_p=0.9, num_captions=3)
    assert len(caps) == 12 and all(isinstance(t, str) for t in caps)


def test_greedy_beyond_64_rows_vs_oracle():
    """Greedy decoding at 72 rows: the projections stay on the fused RMSNorm + GEMM kernels (decode rows <= 512), the LM head leaves
    the 64-row wide kernel for norm kernel + GEMM.  Against the oracle's greedy loop (bf16 may flip a near-tie: >= 90 % of the tokens),
    and the rows of an 8-row run against the same rows inside the 72 (different head kernels: same bar)."""
    cfg = R.RefConfig.small()
    model = build(cfg, 41).eval()
    P = synth.init_params(R.param_shapes(cfg), 41, cfg.d_model, cfg.inner, cfg.d_ff)
    B = 72
    b = synth.make_batch(B, cfg.num_features, 24, 12, cfg.vocab, 41, cfg.vit_dim)
    Ew = P["t5_model.shared.weight"] * 6.0
    P["t5_model.shared.weight"] = Ew
    with torch.no_grad():
        model.t5_model.shared.weight.copy_(Ew.to(DEV))
    video, ids = b["video"].to(DEV), tok(b["input_ids"])
    want = R.greedy_generate(P, cfg, b["video"], b["input_ids"], b["input_ids"] != 0, 10)
    got = model.engine().greedy(video, ids, max_new_tokens=10).cpu()
    n = min(got.shape[1], want.shape[1])
    agree = (got[:, :n] == want[:, :n]).float().mean().item()
    print(f"greedy at {B} rows vs oracle: {agree:.3f} of the tokens identical")
    assert got.shape[0] == B and agree > 0.9
    small = model.engine().greedy(video[:8], {k: v[:8] for k, v in ids.items()}, max_new_tokens=10).cpu()
    m = min(small.shape[1], got.shape[1])
    assert (small[:, :m] == got[:8, :m]).float().mean().item() > 0.9


def test_greedy_fused_tail_equals_separate_launches():
    """The greedy step's tail as ONE launch (v2s_argmax_step_tail: argmax + the next step's embedding row + the step counter's increment by the
    last block to finish) against the three launches it replaces: the same tokens, step for step, in the replayed graph and eagerly, incl. rows that
    finish early (pads after EOS) and the min_length / repetition-penalty processors that run before it; two launches fewer per step."""
    cfg = R.RefConfig.small()
    model = build(cfg, 43).eval()
    eng = model.engine()
    B = 9
    b = synth.make_batch(B, cfg.num_features, 24, 12, cfg.vocab, 43, cfg.vit_dim)
    video, ids = b["video"].to(DEV), tok(b["input_ids"])
    for kw in (dict(), dict(use_graph=False), dict(min_length=5, repetition_penalty=1.3), dict(stop_at_eos=False)):
        out, launches = {}, {}
        for fused in (True, False):
            eng.decode_fuse_tail = fused
            out[fused] = eng.greedy(video, ids, max_new_tokens=14, **kw).cpu()
            launches[fused] = eng.last_decode_launches
        eng.decode_fuse_tail = True
        assert torch.equal(out[True], out[False]), kw
        assert launches[False] - launches[True] == 2, launches


def test_fused_lm_head_equals_unfused():
    """Trainer path: LM head + label-smoothed CE + their backward run chunk by chunk inside the forward (Engine.fused_head; no
    [B*Lo, vocab] logits / d(logits) tensor).  Same kernels on row chunks: loss and every gradient must match the unfused head up to
    the accumulation order of the chunked embedding weight gradient."""
    cfg = R.RefConfig.small()
    b = {k: v.to(DEV) for k, v in synth.make_batch(3, 10, 40, 23, cfg.vocab, 29, cfg.vit_dim, denoising=True).items()}
    res = {}
    # "ce": round 6, the logits are never written (v2s_lmhead_ce_fwd / _bwd: statistics in the GEMM epilogue, tiles recomputed for d(logits));
    # "chunk": round 2, an fp32 logits chunk per 16 rows through v2s_ce_fwd / v2s_ce_bwd; "whole": the unfused head
    for mode in ("ce", "chunk", "whole"):
        model = build(cfg, 13).train()
        eng = model.engine()
        eng.fused_head, eng.head_ce_fused, eng.head_rows = mode != "whole", mode == "ce", 16           # 69 decoder rows -> 5 chunks, the last one ragged
        tr = Trainer(model, lr=1e-3, clip_max_norm=1.0, generative=1.0, denoising=0.5)
        losses = tr.step(b)
        res[mode] = (losses["loss"].item(), losses["denoising_loss"].item(), tr.grad_norm().item(),
                     {k: p.grad.detach().float().cpu().clone() for k, p in model.named_parameters()})
    l0, d0, n0, g0 = res["whole"]
    for mode in ("ce", "chunk"):
        l1, d1, n1, g1 = res[mode]
        print(f"fused head ({mode}): loss {l1:.6f}/{l0:.6f} den {d1:.6f}/{d0:.6f} gnorm {n1:.5f}/{n0:.5f}")
        assert abs(l1 - l0) <= 1e-5 * abs(l0) and abs(d1 - d0) <= 1e-5 * abs(d0) and abs(n1 - n0) <= 1e-3 * n0
        worst = min(cos(g1[k], g0[k]) for k in g0 if g0[k].abs().max() > 0)
        print(f"  worst gradient cosine {mode} vs unfused: {worst:.6f}")
        assert worst > 0.9995


def test_skip_grad_memset_equals_zeroed_arena_and_survives_an_exception():
    """the skip-grad-memset path (a matrix's first weight-gradient GEMM of a step overwrites its gradient) against a zeroed
    arena: identical gradients over two two-pass steps; and an exception inside a step must leave overwrite mode (Trainer's try/finally),
    so that a later autograd-path backward accumulates again."""
    cfg = R.RefConfig.small()
    b = {k: v.to(DEV) for k, v in synth.make_batch(3, 10, 40, 23, cfg.vocab, 31, cfg.vit_dim, denoising=True).items()}
    res = {}
    for skip in (True, False):
        model = build(cfg, 13).train()
        model.engine().skip_grad_memset = skip
        tr = Trainer(model, lr=1e-3, clip_max_norm=1.0, generative=1.0, denoising=1.0)
        for _ in range(2):
            tr.step(b)
        res[skip] = {k: p.grad.detach().float().cpu().clone() for k, p in model.named_parameters()}
    for k in res[True]:
        assert torch.equal(res[True][k], res[False][k]) or cos(res[True][k], res[False][k]) > 0.99999, k
    model = build(cfg, 13).train()
    tr = Trainer(model, lr=1e-3)
    bad = dict(b); bad["output_ids"] = b["output_ids"][:1]            # batch-size mismatch: the step raises somewhere inside
    with pytest.raises(Exception):
        tr.step(bad)
    assert model.engine()._fresh_grads is None


def test_grouped_weight_gradients_equal_single_launches():
    """Engine.group_wgrads: the decoder's / ViT's weight gradients of one projection across the layers are launched as ONE grouped GEMM
    (v2s_gemm_grouped) -- same gradients as one launch per layer, in a two-pass step (second pass accumulates) and with a flush cadence
    that does not divide the layer count."""
    cfg = R.RefConfig.small(n_dec=3)
    b = {k: v.to(DEV) for k, v in synth.make_batch(32, 10, 40, 12, cfg.vocab, 31, cfg.vit_dim, denoising=True).items()}     # 384 decoder rows, 320 ViT rows
    res = {}
    for grouped in (True, False):
        model = build(cfg, 13).train()
        eng = model.engine()
        eng.group_wgrads, eng.group_flush_layers = grouped, 2
        tr = Trainer(model, lr=1e-3, clip_max_norm=1.0, generative=1.0, denoising=1.0)
        tr.step(b)
        res[grouped] = {k: p.grad.detach().float().cpu().clone() for k, p in model.named_parameters()}
    worst = min(cos(res[True][k], res[False][k]) for k in res[True] if res[False][k].abs().max() > 0)
    print(f"grouped vs single weight-gradient launches: worst gradient cosine {worst:.7f}")
    assert worst > 0.99999


def test_beam_sample_rejects_top_k_outside_the_kept_list():
    """beam-sample keeps at most 64 warped candidates per beam row; sampling_top_k = 0 ("no filter") or > 64 would be truncated,
    i.e. a different distribution from HF's beam_sample -- generate() refuses it instead (greedy-loop sampling has no such limit)."""
    cfg = R.RefConfig.small()
    model = build(cfg, 7).eval()
    b = synth.make_batch(2, 10, 24, 12, cfg.vocab, 7, cfg.vit_dim)
    video, ids = b["video"].to(DEV), tok(b["input_ids"])
    for k in (0, 65):
        model.sampling_top_k = k
        with pytest.raises(ValueError):
            model.generate(video, ids, use_nucleus_sampling=True, num_beams=4, max_length=6)
        assert len(model.generate(video, ids, use_nucleus_sampling=True, num_beams=0, max_length=6)) == 2
    model.sampling_top_k = 50
    assert len(model.generate(video, ids, use_nucleus_sampling=True, num_beams=4, max_length=6)) == 2


def test_captured_step_equals_eager_steps():
    """Trainer.step_graph: the whole step (three streams, ~2500 launches at full size) replayed from one hipGraph; batch, dropout salt and
    Adam's lr / bias corrections are read from device memory.  Without dropout three graph-path steps (eager warm-up, capture, replay)
    on three different batches must land where three eager steps land (up to the fp32-atomics order, like the other optimizer tests);
    with dropout and lr = 0 two replays of the SAME batch must give different losses (new masks per replay)."""
    cfg = R.RefConfig.small()
    batches = [{k: v.to(DEV) for k, v in synth.make_batch(3, 10, 40, 17, cfg.vocab, 50 + i, cfg.vit_dim, denoising=True).items()} for i in range(4)]
    m_e, m_g = build(cfg, 19).train(), build(cfg, 19).train()
    m_e.engine().pack = False
    tr_e = Trainer(m_e, lr=1e-3, clip_max_norm=1.0, generative=1.0, denoising=1.0, schedule="linear_with_warmup", num_training_steps=10)
    tr_g = Trainer(m_g, lr=1e-3, clip_max_norm=1.0, generative=1.0, denoising=1.0, schedule="linear_with_warmup", num_training_steps=10)
    le, lg = [], []
    for b in batches:
        le.append(tr_e.step(b)["loss"].item())
        lg.append(tr_g.step_graph(b)["loss"].item())
    print("losses eager", [round(x, 5) for x in le], "graph", [round(x, 5) for x in lg])
    assert tr_g._g["graph"] is not None and tr_g.step_count == tr_e.step_count == 4
    for a, b in zip(le, lg):
        assert abs(a - b) <= 2e-3 * abs(a)
    init = build(cfg, 19)
    worst, maxdiff = 1.0, 0.0
    for (k, p), (_, q), (_, p0) in zip(m_e.named_parameters(), m_g.named_parameters(), init.named_parameters()):
        u1, u2 = (p.detach() - p0.detach().to(DEV)).double().flatten(), (q.detach() - p0.detach().to(DEV)).double().flatten()
        if k.endswith("attn.qkv.bias"):
            n3 = u1.numel() // 3
            u1, u2 = torch.cat([u1[:n3], u1[2 * n3:]]), torch.cat([u2[:n3], u2[2 * n3:]])
        if float(u1.norm()) == 0.0:
            continue
        worst = min(worst, float(u1 @ u2 / (u1.norm() * u2.norm() + 1e-30)))
        maxdiff = max(maxdiff, float((u1 - u2).abs().max()))
    print(f"  captured vs eager after 4 steps: worst update cosine {worst:.4f}, max |dw| diff {maxdiff:.2e}")
    assert worst > 0.9 and maxdiff <= 4 * 2.1e-3
    # dropout: new masks on every replay
    m_d = build(cfg, 19, vis_drop=0.1, enc_drop=0.1, dec_drop=0.1).train()
    tr_d = Trainer(m_d, lr=0.0, clip_max_norm=1.0, generative=1.0, denoising=0.0)
    b = {k: v for k, v in batches[0].items() if not k.startswith("den_")}
    vals = [tr_d.step_graph(b)["loss"].item() for _ in range(4)]
    print("  dropout 0.1, lr 0, same batch:", [round(v, 5) for v in vals])
    assert len({round(v, 6) for v in vals[1:]}) == 3


def test_dropout_stream_follows_torch_seed_and_resumes():
    """the dropout stream is derived from torch.manual_seed() (and the data-parallel rank), not a constant, and its
    position is part of the training state: same seed -> same masks, other seed -> other masks, rng_state()/set_rng_state() resumes."""
    cfg = R.RefConfig.small()
    b = {k: v.to(DEV) for k, v in synth.make_batch(2, 10, 40, 17, cfg.vocab, 9, cfg.vit_dim).items()}

    def run(seed, steps=2, resume_from=None):
        torch.manual_seed(seed)
        model = build(cfg, 8, vis_drop=0.1, enc_drop=0.1, dec_drop=0.1).train()
        tr = Trainer(model, lr=0.0, clip_max_norm=1.0, generative=1.0, denoising=0.0)
        model.num_bins = 0                      # no time-token renorm: with lr = 0 the weights then stay exactly where they are
        if resume_from is not None:
            tr.load_state_dict(resume_from)
        out = [tr.step(b)["loss"].item() for _ in range(steps)]
        return out, tr.state_dict()

    a1, _ = run(1)
    a2, _ = run(1)
    c1, _ = run(2)
    assert a1 == a2 and a1 != c1 and a1[0] != a1[1]
    first, st = run(1, steps=1)
    second, _ = run(5, steps=1, resume_from=st)          # a different torch seed, but the saved stream position wins
    assert first + second == a1
    eng = build(cfg, 8).engine()
    v = eng.arena._seen_version
    eng.mark_dirty()
    assert eng.arena._seen_version != v or v == -1
