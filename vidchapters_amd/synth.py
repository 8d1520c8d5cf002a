"""Deterministic synthetic weights and batches.

A closed-form counter-based generator -- NOT torch's RNG -- so that the build container (where the
reference is importable and goldens are made) and the GPU box (where it is not) produce bit-identical
fp32 tensors.  All arithmetic is int64/float64 torch CPU ops on explicit element indices.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Tuple

import torch

_M32 = 0xFFFFFFFF


def _mix(x: torch.Tensor) -> torch.Tensor:
    """32-bit avalanche hash on int64 lanes (both multipliers < 2**31 so products stay < 2**63)."""
    x = x & _M32
    x = x ^ (x >> 16)
    x = (x * 0x21F0AAAD) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x735A2D97) & _M32
    x = x ^ (x >> 15)
    return x


def hash_u32(n: int, seed: int, stream: int = 0, device="cpu") -> torch.Tensor:
    """n hashed 32-bit values (as int64 in [0, 2**32)) for element indices 0..n-1.  Integer-only,
    so CPU and GPU produce identical bits."""
    idx = torch.arange(n, dtype=torch.int64, device=device)
    key = int(_mix(torch.tensor([(seed * 0x9E3779B1 + stream * 0x85EBCA6B + 0x1234567) & _M32], dtype=torch.int64))[0])
    return _mix(idx ^ key) ^ _mix((idx >> 32) + key + 0x5BD1E995)


def normal(shape, seed: int, std: float = 1.0, mean: float = 0.0, device="cpu") -> torch.Tensor:
    """fp32 ~N(mean, std): Irwin-Hall sum of four 16-bit uniforms (exact integer arithmetic, one
    float64 multiply-add, one rounding to fp32) -- no transcendentals, so bit-identical on any device."""
    n = 1
    for s in shape:
        n *= int(s)
    a = hash_u32(n, seed, 1, device)
    b = hash_u32(n, seed, 2, device)
    tot = (a & 0xFFFF) + (a >> 16) + (b & 0xFFFF) + (b >> 16) - 131070          # in [-131070, 131070]
    z = tot.double() * (math.sqrt(3.0) / 65536.0)
    return (z * std + mean).float().view(*shape)


def key_seed(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF


def init_tensor(name: str, shape: Tuple[int, ...], seed: int, d_model: int, inner: int, d_ff: int,
                device="cpu") -> torch.Tensor:
    """Per-key init law (T5 Mesh-TF style scales for the T5 stack, Xavier-like for the ViT; norm
    weights/biases perturbed away from 1/0 so every code path is exercised by parity tests)."""
    s = key_seed(name, seed)
    if name.endswith("shared.weight"):
        return normal(shape, s, 0.2, device=device)
    if name.endswith("relative_attention_bias.weight"):
        return normal(shape, s, 0.5, device=device)
    if name.endswith("pos_embed"):
        return normal(shape, s, 0.02, device=device)
    if "layer_norm.weight" in name or name.endswith("final_layer_norm.weight"):
        return normal(shape, s, 0.1, 1.0, device=device)
    if ".norm1." in name or ".norm2." in name or name.startswith("visual_encoder.norm."):
        return normal(shape, s, 0.1, 1.0, device=device) if name.endswith("weight") else normal(shape, s, 0.05, device=device)
    if name.endswith(".bias"):
        return normal(shape, s, 0.02, device=device)
    if name.endswith(".q.weight"):
        return normal(shape, s, 0.5 * d_model ** -0.5, device=device)   # score std ~ 0.5*sqrt(d_kv): peaky, unscaled softmax
    if name.endswith(".k.weight") or name.endswith(".v.weight") or name.endswith(".wi.weight"):
        return normal(shape, s, d_model ** -0.5, device=device)
    if name.endswith(".o.weight"):
        return normal(shape, s, inner ** -0.5, device=device)
    if name.endswith(".wo.weight"):
        return normal(shape, s, d_ff ** -0.5, device=device)
    # ViT / projection linears: Xavier-like normal
    fan_out, fan_in = shape[0], shape[1]
    return normal(shape, s, math.sqrt(2.0 / (fan_in + fan_out)), device=device)


def init_params(shapes: Dict[str, Tuple[int, ...]], seed: int, d_model: int, inner: int, d_ff: int,
                device="cpu") -> Dict[str, torch.Tensor]:
    return {k: init_tensor(k, tuple(v), seed, d_model, inner, d_ff, device) for k, v in shapes.items()}


def token_batch(B: int, L: int, vocab: int, seed: int, lo_frac: float = 0.7, eos: int = 1) -> torch.Tensor:
    """int64 [B, L]: ids uniform in [2, vocab), per-row valid length uniform in [lo_frac*L, L],
    last valid token = EOS, tail = pad 0 (masks are ``ids != 0`` as in dvc.py:44-53)."""
    ids = 2 + (hash_u32(B * L, seed, 3) % (vocab - 2))
    ids = ids.view(B, L)
    lo = max(1, int(math.floor(lo_frac * L)))
    lens = lo + (hash_u32(B, seed, 4) % (L - lo + 1))
    pos = torch.arange(L)[None, :]
    ids = torch.where(pos < lens[:, None], ids, torch.zeros_like(ids))
    ids[torch.arange(B), lens - 1] = eos
    return ids.long()


def video_batch(B: int, T: int, D: int, seed: int) -> torch.Tensor:
    """fp32 [B, T, D] ~ N(0,1) (stand-in for pre-extracted CLIP frame features)."""
    return normal((B, T, D), key_seed("video", seed), 1.0)


def make_batch(B: int, T: int, L: int, Lo: int, vocab: int, seed: int, feat_dim: int = 768,
               denoising: bool = False) -> Dict[str, torch.Tensor]:
    """One synthetic training batch with the tensors dvc.py's loop consumes."""
    b = {
        "video": video_batch(B, T, feat_dim, seed),
        "input_ids": token_batch(B, L, vocab, seed * 7 + 1),
        "output_ids": token_batch(B, Lo, vocab, seed * 7 + 2),
    }
    if denoising:  # sizes of a T5 span-corruption pair at noise density .25 / mean span 5 (dataset/dvc_dataset.py:127-142)
        b["den_input_ids"] = token_batch(B, max(8, int(0.8 * L)), vocab, seed * 7 + 3)
        b["den_output_ids"] = token_batch(B, max(8, int(0.3 * L) + 1), vocab, seed * 7 + 4)
    return b
