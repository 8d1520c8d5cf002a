"""Vid2Seq for MI355X: the reference's nn.Module surface over a hand-orchestrated HIP engine.

Surface kept from the reference (model/vid2seq.py:20-167, model/__init__.py:4-19):
``Vid2Seq(t5_path, num_features, embed_dim, depth, heads, mlp_dim, vis_drop, tokenizer, enc_drop, dec_drop,
use_speech, use_video, num_bins, label_smoothing)``, ``forward(video, input_tokenized, output_tokenized) ->
({"loss": loss}, video_dict)``, ``generate(...) -> list[str]``, identical state-dict keys/shapes (tied
``shared``/``embed_tokens``/``lm_head``), ``model.t5_model.shared.weight`` reachable for dvc.py:118-126.

Underneath there is no torch arithmetic on the hot path: the module tree below only *holds* parameters
(views into a flat arena, arena.py); ``Engine`` runs the forward and a hand-written backward as an explicit
sequence of C-ABI kernel launches (lib.py) and hooks into autograd through two coarse Functions
(temporal ViT; T5 encoder+decoder+loss) so that ``loss.backward()`` works for drop-in callers.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import synth
from .tokenizer import batch_decode_spaced

T5_CONFIGS = {   # HF hub config.json values (not shipped with the reference)
    "t5-small": dict(d_model=512, d_kv=64, heads=8, d_ff=2048, n_enc=6, n_dec=6),
    "t5-base": dict(d_model=768, d_kv=64, heads=12, d_ff=3072, n_enc=12, n_dec=12),
    "t5-large": dict(d_model=1024, d_kv=64, heads=16, d_ff=4096, n_enc=24, n_dec=24),
}


@dataclass
class T5Cfg:
    vocab: int
    d_model: int = 768
    d_kv: int = 64
    heads: int = 12
    d_ff: int = 3072
    n_enc: int = 12
    n_dec: int = 12
    buckets: int = 32
    max_distance: int = 128
    eps: float = 1e-6
    pad_id: int = 0
    eos_id: int = 1
    dec_start_id: int = 0

    @property
    def inner(self) -> int:
        return self.heads * self.d_kv


def resolve_t5_config(t5_path) -> dict:
    """``t5_path`` may be a dict (explicit shapes), a directory holding an HF ``config.json``, or a name
    containing one of the known model names (the reference passes e.g. TRANSFORMERS_CACHE + "t5-base")."""
    if isinstance(t5_path, dict):
        return dict(t5_path)
    cfg_file = os.path.join(str(t5_path), "config.json")
    if os.path.isfile(cfg_file):
        c = json.load(open(cfg_file))
        if c.get("feed_forward_proj", "relu") != "relu":
            raise NotImplementedError("gated-act (v1.1) T5 checkpoints are not supported by the HIP path yet")
        return dict(d_model=c["d_model"], d_kv=c["d_kv"], heads=c["num_heads"], d_ff=c["d_ff"], n_enc=c["num_layers"],
                    n_dec=c.get("num_decoder_layers", c["num_layers"]),
                    buckets=c.get("relative_attention_num_buckets", 32),
                    max_distance=c.get("relative_attention_max_distance", 128), eps=c.get("layer_norm_epsilon", 1e-6))
    if "v1_1" in str(t5_path):
        raise NotImplementedError("gated-act (v1.1) T5 is not supported by the HIP path yet")
    for name in sorted(T5_CONFIGS, key=len, reverse=True):
        if name in str(t5_path):
            return dict(T5_CONFIGS[name])
    raise NotImplementedError(f"cannot resolve a T5 configuration from {t5_path!r}")


# ------------------------------------------------------------------------------------------------------
# parameter holders (names mirror the reference so that state_dict keys match; no compute here)
# ------------------------------------------------------------------------------------------------------
class _W(nn.Module):
    def __init__(self, *shape, bias: Optional[int] = None):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(*shape))
        if bias is not None:
            self.bias = nn.Parameter(torch.zeros(bias))


class _T5Attention(nn.Module):
    def __init__(self, c: T5Cfg, rel_bias: bool):
        super().__init__()
        self.q, self.k, self.v = _W(c.inner, c.d_model), _W(c.inner, c.d_model), _W(c.inner, c.d_model)
        self.o = _W(c.d_model, c.inner)
        if rel_bias:
            self.relative_attention_bias = _W(c.buckets, c.heads)


class _T5SelfLayer(nn.Module):
    def __init__(self, c, rel_bias):
        super().__init__()
        self.SelfAttention = _T5Attention(c, rel_bias)
        self.layer_norm = _W(c.d_model)


class _T5CrossLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.EncDecAttention = _T5Attention(c, False)
        self.layer_norm = _W(c.d_model)


class _T5FF(nn.Module):
    def __init__(self, c):
        super().__init__()
        dense = nn.Module()
        dense.wi, dense.wo = _W(c.d_ff, c.d_model), _W(c.d_model, c.d_ff)
        self.DenseReluDense = dense
        self.layer_norm = _W(c.d_model)


class _T5Block(nn.Module):
    def __init__(self, c, decoder: bool, first: bool):
        super().__init__()
        layers = [_T5SelfLayer(c, first)]
        if decoder:
            layers.append(_T5CrossLayer(c))
        layers.append(_T5FF(c))
        self.layer = nn.ModuleList(layers)


class _T5Stack(nn.Module):
    def __init__(self, c, shared, decoder: bool):
        super().__init__()
        self.embed_tokens = shared
        n = c.n_dec if decoder else c.n_enc
        self.block = nn.ModuleList([_T5Block(c, decoder, i == 0) for i in range(n)])
        self.final_layer_norm = _W(c.d_model)


class _T5(nn.Module):
    """Holder with the attribute names of the reference's T5ForConditionalGeneration (modeling_t5.py:1497-1530)."""

    def __init__(self, c: T5Cfg):
        super().__init__()
        self.model_dim = c.d_model
        self.shared = _W(c.vocab, c.d_model)
        self.encoder = _T5Stack(c, self.shared, False)
        self.decoder = _T5Stack(c, self.shared, True)
        self.lm_head = _W(c.vocab, c.d_model)
        self.lm_head.weight = self.shared.weight          # tie_word_embeddings (HF 4.28 re-ties after each resize)


class _VitBlock(nn.Module):
    def __init__(self, dim, mlp):
        super().__init__()
        self.norm1 = _W(dim, bias=dim)
        attn = nn.Module(); attn.qkv = _W(3 * dim, dim, bias=3 * dim); attn.proj = _W(dim, dim, bias=dim)
        self.attn = attn
        self.norm2 = _W(dim, bias=dim)
        mlp_m = nn.Module(); mlp_m.fc1 = _W(mlp, dim, bias=mlp); mlp_m.fc2 = _W(dim, mlp, bias=dim)
        self.mlp = mlp_m


class _ViT(nn.Module):
    def __init__(self, num_features, dim, depth, mlp):
        super().__init__()
        self.pos_embed = nn.Parameter(torch.zeros(1, num_features, dim))
        self.blocks = nn.ModuleList([_VitBlock(dim, mlp) for _ in range(depth)])
        self.norm = _W(dim, bias=dim)


def _load_weight_file(f: str) -> Dict[str, torch.Tensor]:
    if f.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(f)
    return torch.load(f, map_location="cpu")


# ------------------------------------------------------------------------------------------------------
class Vid2Seq(nn.Module):
    def __init__(self, t5_path, num_features=100, embed_dim=768, depth=12, heads=12, mlp_dim=2048, vis_drop=0.,
                 tokenizer=None, enc_drop=0., dec_drop=0.1, use_speech=True, use_video=True, num_bins=100,
                 label_smoothing=0.1, init_seed: Optional[int] = None, device=None):
        """``init_seed`` (extra to the reference signature): an int selects the deterministic synthetic weights of synth.py for EVERY
        parameter (parity tests, benchmarks: there are no checkpoints offline).  ``None`` (the default, what the reference's callers
        get) behaves like the reference: the T5 weights MUST be loadable from ``t5_path`` (``from_pretrained(local_files_only=True)``
        raises otherwise, and so do we), the ViT / projection / time-token rows get the reference's initialisation."""
        super().__init__()
        if tokenizer is None:
            raise ValueError("Vid2Seq needs a tokenizer (len(), pad_token_id, eos_token_id, batch_decode)")
        c = resolve_t5_config(t5_path)
        self.cfg = T5Cfg(vocab=len(tokenizer), **c)
        if self.cfg.d_kv != 64 or embed_dim % heads or embed_dim // heads != 64:
            raise NotImplementedError("the HIP attention kernels are built for head_dim == 64")
        self.vit_dim, self.vit_depth, self.vit_heads, self.vit_mlp = embed_dim, depth, heads, mlp_dim
        self.num_features = num_features
        self.vis_drop, self.enc_drop, self.dec_drop = float(vis_drop), float(enc_drop), float(dec_drop)
        self.label_smoothing = float(label_smoothing)
        self.num_bins = num_bins
        self.t5_model = _T5(self.cfg)
        self.visual_encoder = _ViT(num_features, embed_dim, depth, mlp_dim)
        self.t5_tokenizer = tokenizer
        self.use_speech, self.use_video = use_speech, use_video
        self.proj_v2t = None
        # reference: Linear(768, model_dim) iff model_dim != 768 (vid2seq.py:54-56); expressed on embed_dim so that
        # reduced test shapes work -- identical for every real configuration (embed_dim == 768)
        if self.cfg.d_model != embed_dim:
            self.proj_v2t = _W(self.cfg.d_model, embed_dim, bias=self.cfg.d_model)
        self._engine = None
        if device is not None:          # extra to the reference signature: build (and init) directly on the GPU
            self.to(device)
        if init_seed is not None:
            self.reset_parameters(init_seed)
            self._maybe_load_pretrained(t5_path)
        else:
            self.reference_init()
            if not self._maybe_load_pretrained(t5_path) and not isinstance(t5_path, dict):   # (a dict = explicit shapes, nothing to load)
                raise FileNotFoundError(
                    f"no T5 weights (model.safetensors / pytorch_model.bin, or their sharded index) under {t5_path!r}: the reference loads "
                    "them with from_pretrained(local_files_only=True).  Pass init_seed=<int> for deterministic synthetic weights.")

    # -------------------------------------------------------------------------------- parameters
    def reset_parameters(self, seed: int) -> None:
        """Deterministic closed-form init (synth.py) -- there are no checkpoints in this environment."""
        with torch.no_grad():
            for name, p in self.named_parameters():
                p.copy_(synth.init_tensor(name, tuple(p.shape), seed, self.cfg.d_model, self.cfg.inner, self.cfg.d_ff,
                                          device=p.device))

    def reference_init(self) -> None:
        """The initialisation the reference's constructor leaves behind for everything that does not come from the T5 checkpoint:
        ViT (vit.py:101-115: xavier_uniform linears, biases N(0, 1e-6), LayerNorm 1 / 0, trunc_normal(0.02) pos_embed), proj_v2t
        (nn.Linear default), and for the T5 holder HF's ``_init_weights`` laws (embedding N(0, 1): the law the ``num_bins`` time-token
        rows keep after ``resize_token_embeddings``, vid2seq.py:39-40; norms 1) until a checkpoint overwrites the text rows."""
        c = self.cfg
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.startswith("visual_encoder."):
                    if name.endswith("pos_embed"):
                        nn.init.trunc_normal_(p, std=0.02)
                    elif ".norm" in name or name.startswith("visual_encoder.norm."):
                        p.fill_(1.0 if name.endswith("weight") else 0.0)
                    elif name.endswith(".bias"):
                        nn.init.normal_(p, std=1e-6)      # vit.py:104-108
                    else:
                        nn.init.xavier_uniform_(p)
                elif name.startswith("proj_v2t."):
                    bound = 1.0 / math.sqrt(self.vit_dim)
                    if name.endswith("weight"):
                        nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                    else:
                        p.uniform_(-bound, bound)
                elif "layer_norm" in name:
                    p.fill_(1.0)
                elif name.endswith("shared.weight"):
                    p.normal_(0.0, 1.0)
                elif name.endswith("relative_attention_bias.weight"):
                    p.normal_(0.0, c.d_model ** -0.5)
                elif name.endswith(".q.weight"):
                    p.normal_(0.0, (c.d_model * c.d_kv) ** -0.5)
                elif name.endswith((".k.weight", ".v.weight", ".wi.weight")):
                    p.normal_(0.0, c.d_model ** -0.5)
                elif name.endswith(".o.weight"):
                    p.normal_(0.0, c.inner ** -0.5)
                elif name.endswith(".wo.weight"):
                    p.normal_(0.0, c.d_ff ** -0.5)

    def _maybe_load_pretrained(self, t5_path) -> bool:
        """Load HF T5 weights from a directory: model.safetensors / pytorch_model.bin or their sharded forms (*.index.json).
        Returns False if there is nothing to load."""
        if isinstance(t5_path, dict):
            return False
        d = str(t5_path)
        for fn in ("model.safetensors", "pytorch_model.bin"):
            f = os.path.join(d, fn)
            if os.path.isfile(f):
                self.load_t5_state_dict(_load_weight_file(f))
                return True
            idx = f + ".index.json"
            if os.path.isfile(idx):
                shards = sorted(set(json.load(open(idx))["weight_map"].values()))
                sd = {}
                for sh in shards:
                    sd.update(_load_weight_file(os.path.join(d, sh)))
                self.load_t5_state_dict(sd)
                return True
        return False

    def load_t5_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """Load an HF T5 state dict; the embedding is cut to len(tokenizer)-num_bins rows and the num_bins
        time-token rows keep their fresh init (vid2seq.py:39-40)."""
        own = self.t5_model.state_dict()
        keep = {}
        for k, v in sd.items():
            if k not in own:
                continue
            if own[k].shape != v.shape and k.endswith(("shared.weight", "embed_tokens.weight", "lm_head.weight")):
                n = min(self.cfg.vocab - self.num_bins, v.shape[0])
                t = own[k].clone(); t[:n] = v[:n]; v = t
            keep[k] = v
        self.t5_model.load_state_dict(keep, strict=False)

    # -------------------------------------------------------------------------------- engine plumbing
    def engine(self, device=None) -> "Engine":
        from .engine import Engine
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("Vid2Seq (vidchapters_amd) runs on the GPU only: move the model with .to('cuda') first; "
                               "the HIP path has no CPU fallback")
        if self._engine is None or self._engine.device != dev or not self._engine.arena.intact():
            self._engine = Engine(self, dev)
        return self._engine

    # -------------------------------------------------------------------------------- reference surface
    def forward(self, video, input_tokenized, output_tokenized):
        eng = self.engine()
        return eng.forward(video, input_tokenized, output_tokenized)

    @torch.no_grad()
    def generate(self, video, input_tokenized, use_nucleus_sampling=False, num_beams=4, max_length=256, min_length=1,
                 top_p=0.9, repetition_penalty=1.0, length_penalty=1.0, num_captions=1, temperature=1):
        eng = self.engine()
        if use_nucleus_sampling:
            # HF sample() / beam_sample(): multinomial draws from the warped softmax (temperature, top-k, top-p).  The draws use a
            # counter-based generator keyed on self.sampling_seed (torch's global RNG stream cannot be reproduced): same distribution,
            # different samples.  top_k: the call site (vid2seq.py:150-162) never passes one, so HF's generation default of 50 is in
            # force whenever do_sample is set; ``self.sampling_top_k`` overrides it (0 = no top-k filter).
            if num_captions > 1:    # HF: num_return_sequences expands every input row num_captions times and samples them independently
                video = video.repeat_interleave(num_captions, 0)
                input_tokenized = {k: v.repeat_interleave(num_captions, 0) for k, v in input_tokenized.items()}
            self.sampling_seed = (getattr(self, "sampling_seed", 0) + 1) & 0xFFFFFFFF
            top_k = int(getattr(self, "sampling_top_k", 50))
            sample = (float(top_p), float(temperature), self.sampling_seed, top_k)
            if num_beams > 1 and not (1 <= top_k <= 64):
                # v2s_beam_sample_cand keeps at most 64 warped candidates per beam row: HF's default top_k = 50 fits, "no top-k filter"
                # (0) or a wider k would be truncated to 64 -- a different distribution from HF's beam_sample, so refuse it
                raise ValueError(f"beam-sample (use_nucleus_sampling with num_beams > 1) supports 1 <= sampling_top_k <= 64 (got {top_k}); "
                                 "use num_beams <= 1 for an unrestricted nucleus")
            if num_beams > 1:       # HF 4.28 beam_sample: one best hypothesis per (expanded) input row
                toks = eng.beam_search(video, input_tokenized, num_beams=num_beams, max_new_tokens=max_length, length_penalty=length_penalty,
                                       min_length=min_length, repetition_penalty=repetition_penalty, num_return=1, sample=sample)
            else:
                toks = eng.greedy(video, input_tokenized, max_new_tokens=max_length, repetition_penalty=repetition_penalty,
                                  sample=sample, min_length=min_length)
            return batch_decode_spaced(self.t5_tokenizer, toks, skip_special_tokens=True)
        if num_captions != 1 and num_beams <= 1:
            raise ValueError("num_captions > 1 needs beam search (HF: greedy search returns one sequence)")
        if num_beams > 1:
            toks = eng.beam_search(video, input_tokenized, num_beams=num_beams, max_new_tokens=max_length,
                                   length_penalty=length_penalty, min_length=min_length, repetition_penalty=repetition_penalty,
                                   num_return=num_captions)
        else:
            toks = eng.greedy(video, input_tokenized, max_new_tokens=max_length, repetition_penalty=repetition_penalty, min_length=min_length)
        return batch_decode_spaced(self.t5_tokenizer, toks, skip_special_tokens=True)


def build_vid2seq_model(args, tokenizer) -> Vid2Seq:
    """model/__init__.py:4-19."""
    return Vid2Seq(t5_path=args.model_name, num_features=args.max_feats, embed_dim=args.embedding_dim, depth=args.depth,
                   heads=args.heads, mlp_dim=args.mlp_dim, vis_drop=args.visual_encoder_dropout,
                   enc_drop=args.text_encoder_dropout, dec_drop=args.text_decoder_dropout, tokenizer=tokenizer,
                   num_bins=args.num_bins, label_smoothing=args.label_smoothing, use_speech=args.use_speech,
                   use_video=args.use_video)
