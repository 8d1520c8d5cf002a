"""Input side of the Vid2Seq path: from per-video arrays to the device batch ``Trainer.step`` / ``Vid2Seq.forward`` consume.

Mirrors the reference's data formats -- dataset/dvc_dataset.py (frame subsampling :63-88, ``time_tokenize`` :90-93, sequence
assembly :113-168, ``densevideocaptioning_collate_fn`` :179-226), util/t5.py (span corruption) and the mask rule of dvc.py:44-53
(``attention_mask = ids != 0``) -- with the batch-level work moved to the GPU:

  host (numpy, what the reference also does on the CPU): frame index selection, time tokens, sequence assembly, the random noise
      mask (numpy's RNG, same call sequence as util/t5.py so that seeded runs reproduce the reference's masks);
  pinned staging + ONE async H2D copy per tensor on a side stream;
  device: fp32 -> bf16 feature cast (v2s_cast_bf16), span corruption of the whole batch (v2s_span_corrupt), masks.

Tokenisation itself (sentencepiece) stays with the caller: sequences are passed as id arrays.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as L


# ----------------------------------------------------------------------------------------------------------- host helpers
def frame_indices(n: int, max_feats: int) -> np.ndarray:
    """dataset/dvc_dataset.py:75-79: rows kept when a video has more than ``max_feats`` frames."""
    return (np.arange(max_feats, dtype=np.int64) * n) // max_feats


def subsample_or_pad(frames: np.ndarray, max_feats: int, out: Optional[np.ndarray] = None) -> np.ndarray:
    """dataset/dvc_dataset.py:75-88 -> float32 [max_feats, D] (uniform subsample, or zero-pad at the end)."""
    n, dim = frames.shape
    if out is None:
        out = np.empty((max_feats, dim), dtype=np.float32)
    if n > max_feats:
        out[:] = frames[frame_indices(n, max_feats)]
    else:
        out[:n] = frames
        out[n:] = 0
    return out


def time_tokenize(x: float, duration: float, num_bins: int, num_text_tokens: int) -> int:
    """dataset/dvc_dataset.py:90-93."""
    t = int(float((num_bins - 1) * x) / float(duration))
    if t > num_bins:
        raise ValueError(f"time {x} beyond duration {duration}")
    return t + num_text_tokens


def assemble_sequence(times: Sequence[Tuple[float, float]], texts: Sequence[Sequence[int]], duration: float, num_bins: int,
                      num_text_tokens: int, max_tokens: int, eos: int = 1) -> np.ndarray:
    """dataset/dvc_dataset.py:113-125 (speech) / :149-159 (chapters): ``[<t_start>, <t_end>, text ids...]`` per segment,
    truncated to ``max_tokens - 1`` and closed with EOS.  No segment -> ``[eos]`` (:110,147)."""
    seq: List[int] = []
    for (st, ed), tx in zip(times, texts):
        seq.append(time_tokenize(st, duration, num_bins, num_text_tokens))
        seq.append(time_tokenize(ed, duration, num_bins, num_text_tokens))
        seq.extend(int(t) for t in tx)
    return np.asarray(seq[:max_tokens - 1] + [eos], dtype=np.int64)


def _random_segmentation(num_items: int, num_segments: int) -> np.ndarray:
    first = np.arange(num_items - 1) < (num_segments - 1)
    np.random.shuffle(first)                                  # global numpy RNG, as util/t5.py:70
    return np.unique(np.cumsum(np.pad(first, [[1, 0]])), return_counts=True)[1]


def random_spans_noise_mask(length: int, noise_density: float = 0.25, mean_noise_span_length: float = 5.0) -> np.ndarray:
    """util/t5.py:35-93: boolean [length], spans alternate non-noise / noise starting with non-noise."""
    num_noise = min(max(int(np.round(length * noise_density)), 1), length - 1)
    num_spans = max(int(np.round(num_noise / mean_noise_span_length)), 1)
    noise = _random_segmentation(num_noise, num_spans)
    keep = _random_segmentation(length - num_noise, num_spans)
    starts = np.cumsum(np.reshape(np.stack([keep, noise], axis=1), [num_spans * 2]))[:-1]
    ind = np.zeros((length,), dtype=np.int8)
    ind[starts] = 1
    return (np.cumsum(ind) % 2) == 1


def corrupted_lengths(noise: np.ndarray) -> Tuple[int, int]:
    """Lengths (incl. EOS) of the corrupted input / target of a row with this noise mask."""
    if len(noise) <= 1:
        return 1, 1
    m = noise.astype(bool)
    starts_in = int(bool(m[0])) + int(np.count_nonzero(m[1:] & ~m[:-1]))
    starts_out = int(not m[0]) + int(np.count_nonzero(~m[1:] & m[:-1]))
    return int((~m).sum()) + starts_in + 1, int(m.sum()) + starts_out + 1


def pad_ids(seqs: Sequence[np.ndarray], out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """densevideocaptioning_collate_fn (:184-214): right-pad with 0 to the longest sequence -> int64 [B, max_len]."""
    n = max(len(s) for s in seqs)
    if out is None:
        out = torch.zeros(len(seqs), n, dtype=torch.int64)
    else:
        out.zero_()
    for i, s in enumerate(seqs):
        out[i, :len(s)] = torch.from_numpy(np.asarray(s, dtype=np.int64))
    return out


# ----------------------------------------------------------------------------------------------------------- device batcher
class DeviceBatcher:
    """Turns a list of samples ``{"video": float array [n, D], "input_tokens": ids, "output_tokens": ids}`` into the device batch
    ``{"video" bf16 [B,T,D], "input_ids", "output_ids"[, "den_input_ids", "den_output_ids"]}`` (masks are ``ids != 0``, built
    by the consumers exactly like dvc.py:44-53).  Staging buffers are pinned and reused; copies and the device-side work run on
    ``self.stream`` so that batch k+1 is prepared underneath step k; ``ready`` is recorded for the consumer to wait on."""

    def __init__(self, device, max_feats: int = 100, num_text_tokens: int = 32100, eos: int = 1, denoising: bool = True,
                 noise_density: float = 0.25, mean_noise_span_length: float = 5.0):
        self.device = torch.device(device)
        self.max_feats, self.ntext, self.eos = max_feats, num_text_tokens, eos
        self.denoising, self.noise_density, self.mean_span = denoising, noise_density, mean_noise_span_length
        self.stream = torch.cuda.Stream(device=self.device)
        self.ready = torch.cuda.Event()
        self._pin: Dict[str, torch.Tensor] = {}

    def _pinned(self, key: str, shape, dtype) -> torch.Tensor:
        t = self._pin.get(key)
        n = int(np.prod(shape))
        if t is None or t.numel() < n or t.dtype != dtype:
            t = self._pin[key] = torch.empty(max(n, 1), dtype=dtype).pin_memory()
        return t[:n].view(*shape)

    def __call__(self, samples: Sequence[dict], noise_masks: Optional[Sequence[np.ndarray]] = None) -> Dict[str, torch.Tensor]:
        B = len(samples)
        D = int(np.asarray(samples[0]["video"]).shape[1])
        vid = self._pinned("video", (B, self.max_feats, D), torch.float32)
        vnp = vid.numpy()
        for i, s in enumerate(samples):
            subsample_or_pad(np.asarray(s["video"]), self.max_feats, out=vnp[i])
        ins = [np.asarray(s["input_tokens"], dtype=np.int64) for s in samples]
        outs = [np.asarray(s["output_tokens"], dtype=np.int64) for s in samples]
        in_ids = pad_ids(ins, self._pinned("in", (B, max(len(x) for x in ins)), torch.int64))
        out_ids = pad_ids(outs, self._pinned("out", (B, max(len(x) for x in outs)), torch.int64))
        Lx = in_ids.shape[1]
        if self.denoising:
            if noise_masks is None:
                noise_masks = [random_spans_noise_mask(len(x), self.noise_density, self.mean_span) if len(x) > 1
                               else np.zeros(len(x), dtype=bool) for x in ins]
            lens = self._pinned("lens", (B,), torch.int32)
            noise = self._pinned("noise", (B, Lx), torch.uint8)
            noise.zero_()
            li = lo = 1
            for i, (x, m) in enumerate(zip(ins, noise_masks)):
                lens[i] = len(x)
                noise[i, :len(x)] = torch.from_numpy(np.asarray(m, dtype=np.uint8))
                a, b = corrupted_lengths(np.asarray(m)) if len(x) > 1 else (1, 1)
                li, lo = max(li, a), max(lo, b)
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            dv = vid.to(self.device, non_blocking=True)
            video = torch.empty(B, self.max_feats, D, dtype=torch.bfloat16, device=self.device)
            L.cast_bf16(dv, video, dv.numel())
            batch = {"video": video, "input_ids": in_ids.to(self.device, non_blocking=True),
                     "output_ids": out_ids.to(self.device, non_blocking=True)}
            if self.denoising:
                d_lens, d_noise = lens.to(self.device, non_blocking=True), noise.to(self.device, non_blocking=True)
                den_in = torch.empty(B, li, dtype=torch.int64, device=self.device)
                den_out = torch.empty(B, lo, dtype=torch.int64, device=self.device)
                out_lens = torch.empty(B, 2, dtype=torch.int32, device=self.device)
                L.span_corrupt(batch["input_ids"], d_lens, d_noise, Lx, self.ntext, self.eos, den_in, den_out, out_lens)
                batch.update(den_input_ids=den_in, den_output_ids=den_out, den_lens=out_lens)
            self.ready.record(self.stream)
        for t in batch.values():
            t.record_stream(cur)
        return batch
