"""CPU ORACLE -- test infrastructure, NOT product code.

A plain fp32 PyTorch restatement of the Vid2Seq hot path of antoyang/VidChapters,
written in functional style over a flat ``{state_dict_key: tensor}`` dictionary.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file; the product package ``vidchapters_amd`` never does.

Parity pinning: ``oracle/make_golden.py`` imports the real reference from
``/root/reference`` (build container only), runs it on seeded inputs and checks
this restatement against it (<=1e-5 abs on logits, <=1e-6 rel on loss); the
vectors it emits are committed under ``tests/golden/``.  ``generate()`` arithmetic
lives in the un-vendored dependency ``transformers==4.28.0`` (requirements.txt:9):
its greedy rule is restated in :func:`greedy_generate` and is *parity unpinned* by
any reference test (checked against a hand-rolled loop over the reference's own
cached ``T5ForConditionalGeneration.forward`` instead).

Each function cites the reference file:line it follows (paths relative to the
reference repo root).
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

FMIN = torch.finfo(torch.float32).min

# ----------------------------------------------------------------------------------------------
# "bf16 mode": the same arithmetic with a round-to-bf16 wherever the HIP engine stores a bf16 tensor (vidchapters_amd/engine.py): the bf16
# shadow of every weight matrix, every GEMM / norm / attention output (the residual stream included), the attention probabilities as
# MFMA operands, and -- in the backward pass -- the gradient of each of those tensors, d(scores) and d(logits).  Accumulation stays fp32
# like the MFMA's.  With it, a -m gpu test can hold the engine to a per-tensor gradient cosine near 1 (rounding noise is reproduced instead
# of tolerated), next to the >= 0.97 it reaches against the fp32 reference; tests/test_oracle_cpu.py pins this mode against the fp32
# golden within that measured noise.  Off (exact fp32 restatement of the reference) unless `with bf16_mode():` is active.
# ----------------------------------------------------------------------------------------------
BF16_MODE = False


class bf16_mode:
    def __enter__(self):
        global BF16_MODE
        self.was, BF16_MODE = BF16_MODE, True
        return self

    def __exit__(self, *a):
        global BF16_MODE
        BF16_MODE = self.was


class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return x.bfloat16().float() if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (g.bfloat16().float() if ctx.bwd else g), None, None


def _ra(x):      # a stored activation: value and gradient are bf16 tensors in the engine
    return _Round.apply(x, True, True) if BF16_MODE else x


def _rf(x):      # rounded going forward only: weight shadows (their gradients accumulate in fp32), attention probabilities
    return _Round.apply(x, True, False) if BF16_MODE else x


def _rb(x):      # rounded going backward only: d(scores), d(logits)
    return _Round.apply(x, False, True) if BF16_MODE else x


def _bf(x):
    return x.bfloat16().float()


class _AttnCoreBf16(torch.autograd.Function):
    """softmax(scale * q k^T + bias) v as the engine's flash kernels compute it (csrc/v2s_attn.hip), bf16 mode only: probabilities rounded to
    bf16 as MFMA operands, the context stored in bf16; backward: dO in bf16, delta = rowsum(dO * O) from the STORED (rounded) context -- not
    sum_k P dP --, dP = dO v^T in fp32, dS = P (dP - delta) rounded to bf16 for the dQ / dK products, dV = P_bf16^T dO; the bias gradient sums
    the fp32 dS."""
    @staticmethod
    def forward(ctx, q, k, v, bias, scale):
        s = (q @ k.transpose(-1, -2)) * scale + bias
        p = torch.softmax(s.float(), dim=-1)
        o = _bf(_bf(p) @ v)
        ctx.save_for_backward(q, k, v, p, o)
        ctx.scale, ctx.bias_shape = scale, bias.shape
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, p, o = ctx.saved_tensors
        do = _bf(do)
        delta = (do * o).sum(-1, keepdim=True)
        dp = do @ v.transpose(-1, -2)
        ds = p * (dp - delta)
        dsb = _bf(ds)
        dq = (dsb @ k) * ctx.scale
        dk = (dsb.transpose(-1, -2) @ q) * ctx.scale
        dv = _bf(p).transpose(-1, -2) @ do
        db = ds
        for d_, n in enumerate(ctx.bias_shape):            # reduce to the (broadcast) bias shape
            if n == 1 and db.shape[d_] != 1:
                db = db.sum(d_, keepdim=True)
        return dq, dk, dv, db, None


@dataclass
class RefConfig:
    """Shapes of the path.  Defaults = t5-base + args.py:107-329 defaults."""
    vocab: int = 32200            # 32100 sentencepiece ids + num_bins time tokens (vid2seq.py:39-40)
    d_model: int = 768
    d_kv: int = 64
    heads: int = 12
    d_ff: int = 3072
    n_enc: int = 12
    n_dec: int = 12
    buckets: int = 32
    max_distance: int = 128
    rms_eps: float = 1e-6
    num_features: int = 100       # args.max_feats
    vit_dim: int = 768            # args.embedding_dim
    vit_depth: int = 12
    vit_heads: int = 12
    vit_mlp: int = 2048
    ln_eps: float = 1e-5
    num_bins: int = 100
    label_smoothing: float = 0.1
    pad_id: int = 0
    eos_id: int = 1
    dec_start_id: int = 0
    use_video: bool = True
    use_speech: bool = True

    @property
    def inner(self) -> int:
        return self.heads * self.d_kv

    @staticmethod
    def small(**kw) -> "RefConfig":
        """Reduced shapes for fast parity tests.  head_dim stays 64 (d_kv = vit_dim/vit_heads = 64) because the HIP
        attention kernels are built for the head size of every real T5 / CLIP-ViT configuration."""
        base = dict(vocab=612, d_model=128, d_kv=64, heads=2, d_ff=256, n_enc=2, n_dec=2,
                    num_features=10, vit_dim=128, vit_depth=2, vit_heads=2, vit_mlp=256,
                    num_bins=100)
        base.update(kw)
        return RefConfig(**base)


# ----------------------------------------------------------------------------------------------
# parameter inventory (state-dict keys and shapes)
# ----------------------------------------------------------------------------------------------
def param_shapes(cfg: RefConfig) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    d, inner, ff = cfg.d_model, cfg.inner, cfg.d_ff
    s["t5_model.shared.weight"] = (cfg.vocab, d)
    for stack, n in (("encoder", cfg.n_enc), ("decoder", cfg.n_dec)):
        for i in range(n):
            p = f"t5_model.{stack}.block.{i}.layer."
            for w in "qkv":
                s[p + f"0.SelfAttention.{w}.weight"] = (inner, d)
            s[p + "0.SelfAttention.o.weight"] = (d, inner)
            if i == 0:
                s[p + "0.SelfAttention.relative_attention_bias.weight"] = (cfg.buckets, cfg.heads)
            s[p + "0.layer_norm.weight"] = (d,)
            ffi = 1
            if stack == "decoder":
                for w in "qkv":
                    s[p + f"1.EncDecAttention.{w}.weight"] = (inner, d)
                s[p + "1.EncDecAttention.o.weight"] = (d, inner)
                s[p + "1.layer_norm.weight"] = (d,)
                ffi = 2
            s[p + f"{ffi}.DenseReluDense.wi.weight"] = (ff, d)
            s[p + f"{ffi}.DenseReluDense.wo.weight"] = (d, ff)
            s[p + f"{ffi}.layer_norm.weight"] = (d,)
        s[f"t5_model.{stack}.final_layer_norm.weight"] = (d,)
    v = cfg.vit_dim
    s["visual_encoder.pos_embed"] = (1, cfg.num_features, v)
    for i in range(cfg.vit_depth):
        p = f"visual_encoder.blocks.{i}."
        s[p + "norm1.weight"] = (v,); s[p + "norm1.bias"] = (v,)
        s[p + "attn.qkv.weight"] = (3 * v, v); s[p + "attn.qkv.bias"] = (3 * v,)
        s[p + "attn.proj.weight"] = (v, v); s[p + "attn.proj.bias"] = (v,)
        s[p + "norm2.weight"] = (v,); s[p + "norm2.bias"] = (v,)
        s[p + "mlp.fc1.weight"] = (cfg.vit_mlp, v); s[p + "mlp.fc1.bias"] = (cfg.vit_mlp,)
        s[p + "mlp.fc2.weight"] = (v, cfg.vit_mlp); s[p + "mlp.fc2.bias"] = (v,)
    s["visual_encoder.norm.weight"] = (v,); s["visual_encoder.norm.bias"] = (v,)
    if cfg.d_model != cfg.vit_dim:                       # vid2seq.py:54-56
        s["proj_v2t.weight"] = (cfg.d_model, cfg.vit_dim)
        s["proj_v2t.bias"] = (cfg.d_model,)
    return s


#: state-dict aliases of the tied embedding (modeling_t5.py:1514,1524,1530 + HF tie_weights)
TIED_ALIASES = ("t5_model.encoder.embed_tokens.weight", "t5_model.decoder.embed_tokens.weight",
                "t5_model.lm_head.weight")


# ----------------------------------------------------------------------------------------------
# T5 pieces
# ----------------------------------------------------------------------------------------------
def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """modeling_t5.py:263-277 (T5LayerNorm.forward): no mean, no bias, fp32 variance."""
    var = x.float().pow(2).mean(-1, keepdim=True)
    return _ra(w * (x * torch.rsqrt(var + eps)))


def relative_position_bucket(rel: torch.Tensor, bidirectional: bool, num_buckets: int = 32,
                             max_distance: int = 128) -> torch.Tensor:
    """modeling_t5.py:397-443.  ``rel`` = memory_position - query_position (int64)."""
    out = torch.zeros_like(rel)
    nb = num_buckets
    if bidirectional:
        nb //= 2
        out = out + (rel > 0).long() * nb
        n = rel.abs()
    else:
        n = (-rel).clamp(min=0)
    exact = nb // 2
    # float32 log exactly as the reference computes it (rounding at bucket edges matters)
    big = exact + (torch.log(n.float() / exact) / math.log(max_distance / exact) * (nb - exact)).long()
    big = big.clamp(max=nb - 1)
    return out + torch.where(n < exact, n, big)


def position_bias(table: torch.Tensor, q_len: int, k_len: int, bidirectional: bool,
                  num_buckets: int, max_distance: int) -> torch.Tensor:
    """modeling_t5.py:445-460 (compute_bias) -> [1, H, q_len, k_len]."""
    ctx = torch.arange(q_len, dtype=torch.long)[:, None]
    mem = torch.arange(k_len, dtype=torch.long)[None, :]
    bucket = relative_position_bucket(mem - ctx, bidirectional, num_buckets, max_distance)
    return table[bucket].permute(2, 0, 1).unsqueeze(0)


def _heads(x: torch.Tensor, h: int) -> torch.Tensor:
    b, n, _ = x.shape
    return x.view(b, n, h, -1).transpose(1, 2)


def t5_attention(P: Params, prefix: str, cfg: RefConfig, x: torch.Tensor, bias: torch.Tensor,
                 kv_src: Optional[torch.Tensor] = None,
                 past: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                 cross: bool = False):
    """modeling_t5.py:462-588.  No 1/sqrt(d) scaling (:539-541); ``bias`` already contains the
    additive mask (:559).  Returns (out, (k, v))."""
    q = _heads(_ra(x @ _rf(P[prefix + "q.weight"]).T), cfg.heads)
    if cross and past is not None:
        k, v = past                                            # :524 static cross cache
    else:
        src = kv_src if cross else x
        k = _heads(_ra(src @ _rf(P[prefix + "k.weight"]).T), cfg.heads)
        v = _heads(_ra(src @ _rf(P[prefix + "v.weight"]).T), cfg.heads)
        if past is not None:                                   # :515 growing self cache
            k = torch.cat([past[0], k], dim=2)
            v = torch.cat([past[1], v], dim=2)
    if BF16_MODE:
        o = _AttnCoreBf16.apply(q, k, v, bias, 1.0).transpose(1, 2).reshape(x.shape[0], x.shape[1], cfg.inner)
    else:
        scores = q @ k.transpose(-1, -2) + bias
        w = torch.softmax(scores.float(), dim=-1)
        o = (w @ v).transpose(1, 2).reshape(x.shape[0], x.shape[1], cfg.inner)
    return o @ _rf(P[prefix + "o.weight"]).T, (k, v)          # (bf16 mode: the caller rounds h + this, like the GEMM epilogue does)


def ext_mask(mask: torch.Tensor) -> torch.Tensor:
    """HF 4.28 get_extended_attention_mask / invert_attention_mask for a [B, K] keep-mask
    (call sites modeling_t5.py:996,1005): (1 - m) * finfo.min, shape [B,1,1,K]."""
    return (1.0 - mask[:, None, None, :].float()) * FMIN


def causal_ext_mask(mask: torch.Tensor) -> torch.Tensor:
    """HF 4.28 get_extended_attention_mask for a decoder: causal[q,k] * mask[b,k] -> additive."""
    n = mask.shape[1]
    idx = torch.arange(n)
    causal = (idx[None, :] <= idx[:, None]).float()                    # [q, k]
    keep = causal[None, None] * mask[:, None, None, :].float()        # [B,1,q,k]
    return (1.0 - keep) * FMIN


def t5_self_sublayer(P: Params, p: str, cfg: RefConfig, h: torch.Tensor, bias: torch.Tensor, past=None):
    """T5LayerSelfAttention, modeling_t5.py:598-626: h + SelfAttention(layer_norm(h)) (dropout 0).  ``p`` = "t5_model.<stack>.block.<i>.layer.".
    Returns (h', (k, v)).  One sublayer = one unit of the teacher-forced per-layer GPU parity test (tests/test_layers_gpu.py)."""
    a, kv = t5_attention(P, p + "0.SelfAttention.", cfg, rms_norm(h, P[p + "0.layer_norm.weight"], cfg.rms_eps), bias, past=past)
    return _ra(h + a), kv                                       # :618


def t5_cross_sublayer(P: Params, p: str, cfg: RefConfig, h: torch.Tensor, cbias: torch.Tensor, memory: torch.Tensor, past=None):
    """T5LayerCrossAttention, modeling_t5.py:629-667: h + EncDecAttention(layer_norm(h), memory)."""
    c, kv = t5_attention(P, p + "1.EncDecAttention.", cfg, rms_norm(h, P[p + "1.layer_norm.weight"], cfg.rms_eps), cbias,
                         kv_src=memory, past=past, cross=True)
    return _ra(h + c), kv


def t5_ff_sublayer(P: Params, p: str, j: int, cfg: RefConfig, h: torch.Tensor) -> torch.Tensor:
    """T5LayerFF with T5DenseActDense (ReLU), modeling_t5.py:304-311,338-354: h + wo(relu(wi(layer_norm(h)))).  ``j`` = 1 (encoder) / 2 (decoder)."""
    n = rms_norm(h, P[p + f"{j}.layer_norm.weight"], cfg.rms_eps)
    return _ra(h + _ra(torch.relu(n @ _rf(P[p + f"{j}.DenseReluDense.wi.weight"]).T)) @ _rf(P[p + f"{j}.DenseReluDense.wo.weight"]).T)


def t5_encoder(P: Params, cfg: RefConfig, embeds: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """modeling_t5.py:930-1138 with is_decoder=False, dropout 0 (T5Stack.forward)."""
    h = embeds
    L = h.shape[1]
    tab = P["t5_model.encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
    bias = position_bias(tab, L, L, True, cfg.buckets, cfg.max_distance) + ext_mask(mask)
    for i in range(cfg.n_enc):
        p = f"t5_model.encoder.block.{i}.layer."
        h, _ = t5_self_sublayer(P, p, cfg, h, bias)
        h = t5_ff_sublayer(P, p, 1, cfg, h)
    return rms_norm(h, P["t5_model.encoder.final_layer_norm.weight"], cfg.rms_eps)


def t5_decoder(P: Params, cfg: RefConfig, dec_ids: torch.Tensor, dec_mask: torch.Tensor,
               memory: torch.Tensor, mem_mask: torch.Tensor, past=None, use_cache: bool = False):
    """modeling_t5.py:930-1138 with is_decoder=True.  ``past`` = list of (sk, sv, ck, cv) per layer.
    When ``past`` is given, ``dec_ids`` holds only the new tokens and ``dec_mask`` covers
    past+new positions (:984)."""
    E = _rf(P["t5_model.shared.weight"])
    h = E[dec_ids]
    n_new = dec_ids.shape[1]
    n_past = past[0][0].shape[2] if past is not None else 0
    total = n_past + n_new
    tab = P["t5_model.decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
    sbias = position_bias(tab, total, total, False, cfg.buckets, cfg.max_distance)[:, :, -n_new:, :]  # :555-556
    if past is None:
        sbias = sbias + causal_ext_mask(dec_mask)
    else:
        # HF: with a cache the causal mask is built for `total` keys then sliced to the new rows
        sbias = sbias + causal_ext_mask(dec_mask)[:, :, -n_new:, :]
    cbias = ext_mask(mem_mask)                                  # zero position bias (:544-547) + mask
    present = []
    for i in range(cfg.n_dec):
        p = f"t5_model.decoder.block.{i}.layer."
        pl = past[i] if past is not None else None
        h, skv = t5_self_sublayer(P, p, cfg, h, sbias, past=(pl[0], pl[1]) if pl is not None else None)
        h, ckv = t5_cross_sublayer(P, p, cfg, h, cbias, memory, past=(pl[2], pl[3]) if pl is not None else None)
        h = t5_ff_sublayer(P, p, 2, cfg, h)
        if use_cache:
            present.append((skv[0], skv[1], ckv[0], ckv[1]))
    h = rms_norm(h, P["t5_model.decoder.final_layer_norm.weight"], cfg.rms_eps)
    return h, (present if use_cache else None)


def shift_right(labels: torch.Tensor, cfg: RefConfig) -> torch.Tensor:
    """modeling_t5.py:845-868 (_shift_right)."""
    out = torch.full_like(labels, cfg.pad_id)
    out[:, 1:] = labels[:, :-1]
    out[:, 0] = cfg.dec_start_id
    return out.masked_fill(out == -100, cfg.pad_id)


def lm_logits(P: Params, cfg: RefConfig, h: torch.Tensor) -> torch.Tensor:
    """modeling_t5.py:1709-1714: tied head => rescale by d_model**-0.5."""
    return _rb((h * cfg.d_model ** -0.5) @ _rf(P["t5_model.shared.weight"]).T)


def smoothed_ce(logits: torch.Tensor, labels: torch.Tensor, eps: float) -> torch.Tensor:
    """modeling_t5.py:1721 == F.cross_entropy(ignore_index=-100, label_smoothing=eps), spelled out:
    sum over non-ignored rows of (1-eps)*nll + eps/V * sum_c(-logp_c), divided by #non-ignored."""
    V = logits.shape[-1]
    lp = torch.log_softmax(logits.float().view(-1, V), dim=-1)
    y = labels.view(-1)
    keep = y != -100
    nll = -lp.gather(1, y.clamp(min=0)[:, None]).squeeze(1)
    smooth = -lp.sum(-1)
    per = (1.0 - eps) * nll + (eps / V) * smooth
    return (per * keep).sum() / keep.sum()


# ----------------------------------------------------------------------------------------------
# temporal ViT
# ----------------------------------------------------------------------------------------------
def vit_block(P: Params, p: str, cfg: RefConfig, x: torch.Tensor) -> torch.Tensor:
    """model/vit.py:73-76 (Block.forward) with Attention :38-55 and Mlp :16-22, dropout 0.  ``p`` = "visual_encoder.blocks.<i>."."""
    B, N, C = x.shape
    H = cfg.vit_heads
    scale = (C // H) ** -0.5
    n = _ra(F.layer_norm(x, (C,), P[p + "norm1.weight"], P[p + "norm1.bias"], cfg.ln_eps))
    qkv = _ra(n @ _rf(P[p + "attn.qkv.weight"]).T + P[p + "attn.qkv.bias"]).reshape(B, N, 3, H, C // H).permute(2, 0, 3, 1, 4)
    if BF16_MODE:
        o = _AttnCoreBf16.apply(qkv[0], qkv[1], qkv[2], torch.zeros(1, 1, 1, 1), scale).transpose(1, 2).reshape(B, N, C)
    else:
        att = torch.softmax((qkv[0] @ qkv[1].transpose(-1, -2)) * scale, dim=-1)
        o = (att @ qkv[2]).transpose(1, 2).reshape(B, N, C)
    x = _ra(x + (o @ _rf(P[p + "attn.proj.weight"]).T + P[p + "attn.proj.bias"]))
    n = _ra(F.layer_norm(x, (C,), P[p + "norm2.weight"], P[p + "norm2.bias"], cfg.ln_eps))
    hdn = _ra(F.gelu(n @ _rf(P[p + "mlp.fc1.weight"]).T + P[p + "mlp.fc1.bias"]))
    return _ra(x + (hdn @ _rf(P[p + "mlp.fc2.weight"]).T + P[p + "mlp.fc2.bias"]))


def vit_forward(P: Params, cfg: RefConfig, x: torch.Tensor) -> torch.Tensor:
    """model/vit.py:117-133 (+ Block :73-76, Attention :38-55, Mlp :16-22), dropout 0."""
    pos = P["visual_encoder.pos_embed"]
    if x.shape[1] != pos.shape[1]:                               # :119-123 nearest resize
        pos = F.interpolate(pos.transpose(1, 2), size=x.shape[1], mode="nearest").transpose(1, 2)
    x = _ra(_rf(x) + _rf(pos))
    for i in range(cfg.vit_depth):
        x = vit_block(P, f"visual_encoder.blocks.{i}.", cfg, x)
    C = x.shape[-1]
    return _ra(F.layer_norm(x, (C,), P["visual_encoder.norm.weight"], P["visual_encoder.norm.bias"], cfg.ln_eps))


# ----------------------------------------------------------------------------------------------
# Vid2Seq surface
# ----------------------------------------------------------------------------------------------
def encode(P: Params, cfg: RefConfig, video, input_ids, input_mask):
    """model/vid2seq.py:59-84 (shared by forward and generate).  Returns (memory, memory_mask, video_dict)."""
    video_dict = None
    if cfg.use_video:
        if isinstance(video, dict):
            vis, atts = video["video"], video["atts_vis"]
        else:
            vis = vit_forward(P, cfg, video)
            if "proj_v2t.weight" in P:
                vis = _ra(vis @ _rf(P["proj_v2t.weight"]).T + P["proj_v2t.bias"])
            atts = torch.ones(vis.shape[:-1], dtype=torch.long)
        video_dict = {"video": vis, "atts_vis": atts}
    if cfg.use_speech:
        text = t5_encoder(P, cfg, _rf(P["t5_model.shared.weight"])[input_ids], input_mask)
    if cfg.use_video and cfg.use_speech:
        return torch.cat([vis, text], 1), torch.cat([atts, input_mask.long()], 1), video_dict
    if cfg.use_video:
        return vis, atts, video_dict
    return text, input_mask.long(), video_dict


def vid2seq_logits(P: Params, cfg: RefConfig, video, input_ids, input_mask, output_ids, output_mask):
    """Logits of Vid2Seq.forward (vid2seq.py:58-98 -> modeling_t5.py:1587-1738)."""
    memory, mem_mask, video_dict = encode(P, cfg, video, input_ids, input_mask)
    targets = output_ids.masked_fill(output_ids == cfg.pad_id, -100)           # vid2seq.py:86-88
    dec_in = shift_right(targets, cfg)
    h, _ = t5_decoder(P, cfg, dec_in, output_mask, memory, mem_mask)
    return lm_logits(P, cfg, h), targets, video_dict


def vid2seq_forward(P: Params, cfg: RefConfig, video, input_ids, input_mask, output_ids, output_mask):
    """Vid2Seq.forward -> ({'loss': loss}, video_dict)."""
    logits, targets, video_dict = vid2seq_logits(P, cfg, video, input_ids, input_mask, output_ids, output_mask)
    return {"loss": smoothed_ce(logits, targets, cfg.label_smoothing)}, video_dict


@torch.no_grad()
def warp_scores(scores: torch.Tensor, top_p: float, temperature: float = 1.0, top_k: int = 0, min_keep: int = 1) -> torch.Tensor:
    """HF 4.28 logits warpers in the order ``_get_logits_warper`` builds them: TemperatureLogitsWarper, TopKLogitsWarper (skipped for
    top_k == 0; HF's generation default is 50), TopPLogitsWarper (skipped for top_p == 1).  ``min_keep`` = ``min_tokens_to_keep``: 1 for
    sample(), 2 for beam_sample().  Returns the warped scores, removed entries at -inf."""
    scores = scores.float() / temperature
    if top_k:
        k = min(max(top_k, min_keep), scores.shape[-1])
        scores = scores.masked_fill(scores < torch.topk(scores, k)[0][..., -1, None], -float("inf"))
    if top_p < 1.0:
        sl, si = torch.sort(scores, descending=False, dim=-1)
        remove = sl.softmax(-1).cumsum(-1) <= (1 - top_p)
        remove[..., -min_keep:] = False
        scores = scores.masked_fill(remove.scatter(-1, si, remove), -float("inf"))
    return scores


def top_p_probs(logits: torch.Tensor, top_p: float, temperature: float = 1.0, top_k: int = 0) -> torch.Tensor:
    """The distribution HF 4.28 sample() draws from: temperature, top-k, top-p warpers (min_tokens_to_keep=1), softmax."""
    return warp_scores(logits, top_p, temperature, top_k, 1).softmax(-1)


def hash32(x):
    """v2s_hash32 (vidchapters_amd/csrc/v2s_common.h) on numpy uint32 arrays / Python ints."""
    import numpy as np
    x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    x ^= x >> 16; x = (x * 0x7feb352d) & 0xFFFFFFFF; x ^= x >> 15; x = (x * 0x846ca68b) & 0xFFFFFFFF; x ^= x >> 16
    return x


def beam_sample_gumbel(seed: int, step: int, rows: int, V: int) -> torch.Tensor:
    """The Gumbel noise of v2s_beam_sample_cand for every (row, token) of one step: a counter-based hash of (seed, step, row, token)
    -> 24-bit uniform u -> -log(-log u).  Test infrastructure for the beam-sample restatement below (torch's multinomial stream, which
    HF 4.28 beam_sample draws from, cannot be reproduced on the device: the kernel's own noise is restated instead)."""
    import numpy as np
    r = hash32((np.arange(rows, dtype=np.uint64) * 0x9E3779B1 + step * 0x85EBCA6B + 0x1234567) & 0xFFFFFFFF)
    t = hash32((np.arange(V, dtype=np.uint64) * 0xC2B2AE35 + 0x27D4EB2F) & 0xFFFFFFFF)
    h = hash32((seed & 0xFFFFFFFF) ^ r[:, None] ^ t[None, :])
    u = ((h >> 8).astype(np.float64) + 0.5) / 16777216.0
    return torch.from_numpy(-np.log(-np.log(u))).float()


def beam_sample_step(logits: torch.Tensor, beam_scores: torch.Tensor, num_beams: int, top_p: float, temperature: float, top_k: int,
                     noise: torch.Tensor, ban_token: int = -1):
    """One step of HF 4.28 beam_sample() up to the scorer: log_softmax -> (MinLength ban) -> + beam score -> warpers (min_keep 2) ->
    2*num_beams draws WITHOUT replacement from the softmax over an entry's nb*V warped scores, expressed as the largest keys
    score + noise (Gumbel top-k = successive multinomial draws without replacement) -> sorted by score.  Returns (scores, tokens,
    beam indices), each [B, 2*nb]."""
    R_, V = logits.shape
    B = R_ // num_beams
    sc = torch.log_softmax(logits.float(), -1)
    if ban_token >= 0:
        sc[:, ban_token] = -float("inf")
    sc = warp_scores(sc + beam_scores[:, None], top_p, temperature, top_k, 2)
    key = (sc + noise).view(B, num_beams * V)
    flat = sc.view(B, num_beams * V)
    drawn = torch.topk(key, 2 * num_beams, dim=1)[1]
    val = torch.gather(flat, 1, drawn)
    val, order = torch.sort(val, descending=True, dim=1, stable=True)
    drawn = torch.gather(drawn, 1, order)
    return val, drawn % V, drawn // V


def repetition_penalty_(scores: torch.Tensor, seq: torch.Tensor, penalty: float) -> None:
    """transformers 4.28 RepetitionPenaltyLogitsProcessor (in place): every token that occurs in ``seq`` (decoder ids so far, start
    token included) has its score multiplied by ``penalty`` if negative, divided by it otherwise; gather-then-scatter, so a token is
    penalised once however often it occurs."""
    sc = torch.gather(scores, 1, seq)
    sc = torch.where(sc < 0, sc * penalty, sc / penalty)
    scores.scatter_(1, seq, sc)


def greedy_generate(P: Params, cfg: RefConfig, video, input_ids, input_mask, max_new_tokens: int = 256,
                    repetition_penalty: float = 1.0, min_length: int = 1):
    """vid2seq.py:100-167 with num_beams=1, do_sample=False: HF 4.28 GenerationMixin.greedy_search
    rules: start token 0, argmax of the last-position logits, rows that already
    emitted EOS emit pad, stop when every row is finished or after max_new_tokens new tokens
    (MinLength(1) is a no-op; min_length > 1 bans EOS while the decoder sequence is shorter).  Returns int64 [B, <=1+max_new_tokens]
    including the start token."""
    memory, mem_mask, _ = encode(P, cfg, video, input_ids, input_mask)
    B = memory.shape[0]
    seq = torch.full((B, 1), cfg.dec_start_id, dtype=torch.long)
    unfinished = torch.ones(B, dtype=torch.long)
    past = None
    for _ in range(max_new_tokens):
        step_in = seq if past is None else seq[:, -1:]
        dec_mask = torch.ones(B, seq.shape[1], dtype=torch.long)      # HF passes no decoder mask => ones
        h, past = t5_decoder(P, cfg, step_in, dec_mask, memory, mem_mask, past=past, use_cache=True)
        logits = lm_logits(P, cfg, h[:, -1:]).squeeze(1).float()
        if repetition_penalty != 1.0:                                 # greedy_search: processors run on the raw logits
            repetition_penalty_(logits, seq, repetition_penalty)
        if seq.shape[1] < min_length:                                 # HF 4.28 MinLengthLogitsProcessor (cur_len = decoder ids so far)
            logits[:, cfg.eos_id] = -float("inf")
        nxt = logits.argmax(-1)
        nxt = nxt * unfinished + cfg.pad_id * (1 - unfinished)
        seq = torch.cat([seq, nxt[:, None]], 1)
        unfinished = unfinished * (nxt != cfg.eos_id).long()
        if unfinished.max() == 0:
            break
    return seq


class _BeamHyps:
    """transformers==4.28.0 generation/beam_search.py:BeamHypotheses (early_stopping=False heuristic)."""

    def __init__(self, num_beams: int, length_penalty: float):
        self.num_beams, self.length_penalty = num_beams, length_penalty
        self.beams: List[Tuple[float, torch.Tensor]] = []
        self.worst_score = 1e9

    def add(self, hyp: torch.Tensor, sum_logprobs: float) -> None:
        score = sum_logprobs / (hyp.shape[-1] ** self.length_penalty)
        if len(self.beams) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp))
            if len(self.beams) > self.num_beams:
                sorted_next = sorted([(s, i) for i, (s, _) in enumerate(self.beams)])
                del self.beams[sorted_next[0][1]]
                self.worst_score = sorted_next[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs: float, cur_len: int) -> bool:
        if len(self.beams) < self.num_beams:
            return False
        return self.worst_score >= best_sum_logprobs / cur_len ** self.length_penalty


@torch.no_grad()
def beam_search_core(step_logp, B: int, nb: int, V: int, eos_id: int, pad_id: int, start_id: int, max_length: int,
                     length_penalty: float = 1.0, min_length: int = 1, repetition_penalty: float = 1.0, num_return_sequences: int = 1,
                     trace: Optional[list] = None):
    """transformers==4.28.0 GenerationMixin.beam_search + BeamSearchScorer (early_stopping=False) over an arbitrary next-token model:
    ``step_logp(seq, beam_idx)`` returns log-softmax scores [B*nb, V] for the sequences ``seq`` ([B*nb, len]); ``beam_idx`` (None at the
    first call) is the row permutation that produced ``seq`` from the previous call's rows (for reordering a cache)."""
    seq = torch.full((B * nb, 1), start_id, dtype=torch.long)
    beam_scores = torch.zeros(B, nb)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    hyps = [_BeamHyps(nb, length_penalty) for _ in range(B)]
    done = [False] * B
    bidx = None
    while True:
        logp = step_logp(seq, bidx)
        if repetition_penalty != 1.0:                                 # beam_search: processors run on the log-probabilities
            repetition_penalty_(logp, seq, repetition_penalty)
        if seq.shape[-1] < min_length:                         # MinLengthLogitsProcessor (applied to the log-probs in 4.28 beam_search)
            logp[:, eos_id] = -float("inf")
        logp = logp + beam_scores[:, None]
        top_s, top_i = torch.topk(logp.view(B, nb * V), 2 * nb + 1, dim=1, largest=True, sorted=True)
        if trace is not None:       # goldens: the 2*nb + 1 best candidate scores of every entry (gaps = how firmly the step was decided)
            trace.append({"scores": top_s.clone(), "tokens": (top_i % V).clone(), "beams": (top_i // V).clone()})
        top_s, top_i = top_s[:, :2 * nb], top_i[:, :2 * nb]
        nidx, ntok = top_i // V, top_i % V
        cur_len = seq.shape[-1]
        new_scores = torch.zeros(B, nb); new_tok = torch.zeros(B, nb, dtype=torch.long); new_idx = torch.zeros(B, nb, dtype=torch.long)
        for b in range(B):
            if done[b]:
                new_tok[b] = pad_id
                continue
            k = 0
            for rank in range(2 * nb):
                t, sc, bi = int(ntok[b, rank]), float(top_s[b, rank]), b * nb + int(nidx[b, rank])
                if t == eos_id:
                    if rank >= nb:
                        continue
                    hyps[b].add(seq[bi].clone(), sc)
                else:
                    new_scores[b, k], new_tok[b, k], new_idx[b, k] = sc, t, bi
                    k += 1
                if k == nb:
                    break
            done[b] = done[b] or hyps[b].is_done(float(top_s[b].max()), cur_len)
        beam_scores, bt, bidx = new_scores.view(-1), new_tok.view(-1), new_idx.view(-1)
        if trace is not None:       # the step's decisions: next tokens / scores / source rows of the nb beams, and which entries are done
            trace[-1].update(next_scores=new_scores.clone(), next_tokens=new_tok.clone(), next_src=new_idx.clone(), done=torch.tensor(done))
        seq = torch.cat([seq[bidx], bt[:, None]], -1)
        if all(done) or seq.shape[-1] >= max_length:
            break
    for b in range(B):                                                              # BeamSearchScorer.finalize
        if done[b]:
            continue
        for j in range(nb):
            hyps[b].add(seq[b * nb + j], float(beam_scores[b * nb + j]))
    best = []                                                                       # num_return_sequences best hypotheses per entry,
    for hb in hyps:                                                                 # best first (BeamSearchScorer.finalize pops the sorted list)
        srt = sorted(hb.beams, key=lambda x: x[0])
        best += [srt.pop()[1] for _ in range(num_return_sequences)]
    lens = [len(x) for x in best]
    out_len = min(max(lens) + 1, max_length)
    out = torch.full((len(best), out_len), pad_id, dtype=torch.long)
    for b, hyp in enumerate(best):
        out[b, :lens[b]] = hyp
        if lens[b] < out_len:
            out[b, lens[b]] = eos_id
    return out


def beam_generate(P: Params, cfg: RefConfig, video, input_ids, input_mask, num_beams: int = 4, max_new_tokens: int = 256,
                  length_penalty: float = 1.0, min_length: int = 1, repetition_penalty: float = 1.0, num_return_sequences: int = 1,
                  trace: Optional[list] = None):
    """vid2seq.py:150-162 with num_beams>1, do_sample=False, early_stopping=False, num_return_sequences=1:
    transformers==4.28.0 GenerationMixin.beam_search + BeamSearchScorer (un-vendored dependency -> restated from the
    published algorithm; *parity unpinned by reference tests*, cross-checked against the installed transformers'
    generate in oracle/make_golden.py).  Returns int64 [B, <= 1+max_new_tokens] (start token first, pad after EOS)."""
    memory, mem_mask, _ = encode(P, cfg, video, input_ids, input_mask)
    B, nb = memory.shape[0], num_beams
    V = P["t5_model.shared.weight"].shape[0]
    mem = memory.repeat_interleave(nb, 0)
    mmask = mem_mask.repeat_interleave(nb, 0)
    state = {"past": None}

    def step_logp(seq, bidx):
        past = state["past"]
        if past is not None:
            past = [tuple(x.index_select(0, bidx) for x in layer) for layer in past]      # modeling_t5.py:1771-1793
        step_in = seq if past is None else seq[:, -1:]
        h, state["past"] = t5_decoder(P, cfg, step_in, torch.ones(B * nb, seq.shape[1], dtype=torch.long), mem, mmask, past=past, use_cache=True)
        return torch.log_softmax(lm_logits(P, cfg, h[:, -1:]).squeeze(1).float(), dim=-1)

    return beam_search_core(step_logp, B, nb, V, cfg.eos_id, cfg.pad_id, cfg.dec_start_id, max_new_tokens + 1, length_penalty,
                            min_length, repetition_penalty, num_return_sequences, trace=trace)


# ----------------------------------------------------------------------------------------------
# caller-side recipe (dvc.py) restated
# ----------------------------------------------------------------------------------------------
def renorm_time_tokens(emb: torch.Tensor, num_bins: int) -> None:
    """dvc.py:118-126 (in place): rescale the last num_bins rows so that their mean row-norm equals
    the mean row-norm of the other rows."""
    frozen = emb[:-num_bins].norm(dim=1).mean()
    train = emb[-num_bins:].norm(dim=1).mean()
    emb[-num_bins:].div_(train / frozen)


def lr_at(step: int, total: int, base_lr: float, schedule: str = "", warmup_frac: float = 0.1) -> float:
    """util/misc.py:15-42 (adjust_learning_rate)."""
    warm = round(warmup_frac * total)
    if schedule == "":
        return base_lr
    if schedule not in ("linear_with_warmup", "cosine_with_warmup"):
        raise NotImplementedError(schedule)
    if step < warm:
        return base_lr * (float(step) / float(max(1, warm)))        # gamma first, like the reference: bit-identical LR
    if schedule == "linear_with_warmup":
        return base_lr * max(0.0, float(total - step) / float(max(1, total - warm)))
    if schedule == "cosine_with_warmup":
        return base_lr * (1 + math.cos(math.pi * float(step - warm) / float(max(1, total - warm)))) / 2
    raise NotImplementedError(schedule)


def train_step(P: Params, state: dict, cfg: RefConfig, batch: dict, lr: float = 3e-4,
               clip: float = 1.0, generative: float = 1.0, denoising: float = 1.0,
               betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0) -> dict:
    """dvc.py:71-126 with dropout 0: generative pass, optional denoising pass on the cached video_dict,
    clip_grad_norm_, Adam (torch.optim.Adam defaults, dvc.py:346-351), time-token renorm (twice on the
    tied tensor, like the reference).  ``P`` leaves must require grad; updated in place.
    ``state`` holds {'step': int, 'm': {...}, 'v': {...}}."""
    losses = {}
    total = 0.0
    video_dict = None
    if generative:
        out, video_dict = vid2seq_forward(P, cfg, batch["video"], batch["input_ids"], batch["input_ids"] != 0,
                                          batch["output_ids"], batch["output_ids"] != 0)
        losses["loss"] = out["loss"]
        total = generative * out["loss"]
    if denoising:
        vid = video_dict if generative else batch["video"]
        out, _ = vid2seq_forward(P, cfg, vid, batch["den_input_ids"], batch["den_input_ids"] != 0,
                                 batch["den_output_ids"], batch["den_output_ids"] != 0)
        losses["denoising_loss"] = out["loss"]
        total = total + denoising * out["loss"]
    names = [k for k, v in P.items() if v.requires_grad]
    grads = torch.autograd.grad(total, [P[k] for k in names], allow_unused=True)
    grads = [g if g is not None else torch.zeros_like(P[k]) for k, g in zip(names, grads)]
    gnorm = torch.sqrt(sum((g.float() ** 2).sum() for g in grads))
    if clip > 0:
        coef = torch.clamp(clip / (gnorm + 1e-6), max=1.0)      # torch clip_grad_norm_
        grads = [g * coef for g in grads]
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    with torch.no_grad():
        for k, g in zip(names, grads):
            if weight_decay:
                g = g + weight_decay * P[k]
            m = state.setdefault("m", {}).setdefault(k, torch.zeros_like(P[k]))
            v = state.setdefault("v", {}).setdefault(k, torch.zeros_like(P[k]))
            m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
            v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
            bc1, bc2 = 1 - betas[0] ** t, 1 - betas[1] ** t
            denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
            P[k].addcdiv_(m, denom, value=-lr / bc1)
        if cfg.num_bins:
            renorm_time_tokens(P["t5_model.shared.weight"], cfg.num_bins)   # shared
            renorm_time_tokens(P["t5_model.shared.weight"], cfg.num_bins)   # lm_head (tied => same tensor)
    return {"losses": {k: float(v) for k, v in losses.items()}, "grad_norm": float(gnorm)}


def parse_chapters(text: str, duration: float, num_bins: int) -> List[dict]:
    """dvc.py:186-212 / demo_vid2seq.py:170-197: pair consecutive <time=k> tokens into events."""
    seqs = re.split(r"(?<!<)\s+(?!>)", text)
    idxs = [j for j in range(len(seqs) - 1) if seqs[j][:6] == "<time=" and seqs[j + 1][:6] == "<time="]
    last, res = -2, []
    for j, idx in enumerate(idxs):
        if idx == last + 1:
            continue
        stop = idxs[j + 1] if j < len(idxs) - 1 else len(seqs)
        words = [seqs[k] for k in range(idx + 2, stop) if seqs[k] != "<time="]
        if not words:
            continue
        s = re.search(r"\<time\=(\d+)\>", seqs[idx]); e = re.search(r"\<time\=(\d+)\>", seqs[idx + 1])
        assert s and e
        start = float(int(s.group(1))) * float(duration) / float(num_bins - 1)
        end = float(int(e.group(1))) * float(duration) / float(num_bins - 1)
        if end <= start:
            continue
        res.append({"sentence": " ".join(words), "timestamp": [start, end]})
        last = idx
    return res
