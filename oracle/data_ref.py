"""CPU restatement (numpy) of the data formats on the input side of the Vid2Seq path -- TEST INFRASTRUCTURE ONLY
(imported by tests/ and oracle/make_golden.py; the product never imports it).

Follows dataset/dvc_dataset.py (frame subsampling :63-88, time tokens :90-93, sequence assembly :113-168, collate :179-226) and
util/t5.py (T5 span corruption :3-93).  Pinned by tests/golden/data_pipeline.npz, which oracle/make_golden.py writes by running
the reference's own functions on seeded inputs.  All of it is integer / byte work: parity is bit-exact.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np


def get_video(frames: np.ndarray, max_feats: int) -> np.ndarray:
    """dataset/dvc_dataset.py:75-88: uniform subsample (index (j*n)//max_feats) or zero-pad to max_feats rows."""
    n, dim = frames.shape
    frames = frames.astype(np.float32)
    if n > max_feats:
        return np.stack([frames[(j * n) // max_feats] for j in range(max_feats)])
    if n < max_feats:
        return np.concatenate([frames, np.zeros((max_feats - n, dim), np.float32)], 0)
    return frames


def time_tokenize(x: float, duration: float, num_bins: int, num_text_tokens: int) -> int:
    """dataset/dvc_dataset.py:90-93."""
    t = int(float((num_bins - 1) * x) / float(duration))
    assert t <= num_bins
    return t + num_text_tokens


def random_spans_noise_mask(length: int, noise_density: float, mean_noise_span_length: float) -> np.ndarray:
    """util/t5.py:35-93 (draws from numpy's global RNG with the same call sequence: two np.random.shuffle calls)."""
    num_noise_tokens = int(np.round(length * noise_density))
    num_noise_tokens = min(max(num_noise_tokens, 1), length - 1)
    num_noise_spans = max(int(np.round(num_noise_tokens / mean_noise_span_length)), 1)
    num_nonnoise_tokens = length - num_noise_tokens

    def segmentation(num_items, num_segments):
        first = np.arange(num_items - 1) < (num_segments - 1)
        np.random.shuffle(first)
        seg_id = np.cumsum(np.pad(first, [[1, 0]]))
        return np.unique(seg_id, return_counts=True)[1]

    noise = segmentation(num_noise_tokens, num_noise_spans)
    nonnoise = segmentation(num_nonnoise_tokens, num_noise_spans)
    inter = np.reshape(np.stack([nonnoise, noise], axis=1), [num_noise_spans * 2])
    starts = np.cumsum(inter)[:-1]
    ind = np.zeros((length,), dtype=np.int8)
    ind[starts] = True
    return np.equal(np.cumsum(ind) % 2, 1)[:length]


def create_sentinel_ids(mask: np.ndarray, num_text_tokens: int) -> np.ndarray:
    """util/t5.py:3-16 with len(tokenizer) - num_bins = num_text_tokens; mask int8 [1, L]."""
    start = mask - np.roll(mask, 1, axis=-1) * mask
    start[:, 0] = mask[:, 0]
    sent = np.where(start != 0, np.cumsum(start, axis=-1), start)
    sent = np.where(sent != 0, (num_text_tokens - sent), 0)
    sent -= mask - start
    return sent


def filter_input_ids(input_ids: np.ndarray, sentinel_ids: np.ndarray, eos: int) -> np.ndarray:
    """util/t5.py:19-32."""
    bs = input_ids.shape[0]
    full = np.where(sentinel_ids != 0, sentinel_ids, input_ids)
    ids = full[full >= 0].reshape((bs, -1))
    return np.concatenate([ids, np.full((bs, 1), eos, dtype=np.int32)], axis=-1)


def span_corrupt(input_tokens: np.ndarray, noise: np.ndarray, num_text_tokens: int, eos: int) -> Tuple[np.ndarray, np.ndarray]:
    """dataset/dvc_dataset.py:127-145 for one sequence given its noise mask -> (denoising_input, denoising_output)."""
    if len(input_tokens) <= 1:
        return np.array([0], dtype=np.int64), np.array([eos], dtype=np.int64)
    m = np.asarray([noise])
    in_sent = create_sentinel_ids(m.astype(np.int8), num_text_tokens)
    lab_sent = create_sentinel_ids((~m).astype(np.int8), num_text_tokens)
    den_out = filter_input_ids(input_tokens[None], lab_sent, eos)[0]
    den_in = filter_input_ids(input_tokens[None], in_sent, eos)[0]
    return den_in.astype(np.int64), den_out.astype(np.int64)


def assemble(times: Sequence[Tuple[float, float]], texts: Sequence[Sequence[int]], duration: float, num_bins: int,
             num_text_tokens: int, max_tokens: int, eos: int) -> np.ndarray:
    """dataset/dvc_dataset.py:113-125 / :149-159: [t_start, t_end, text...] per segment, truncated to max_tokens-1, + EOS."""
    out: List[int] = []
    for (st, ed), tx in zip(times, texts):
        out += [time_tokenize(st, duration, num_bins, num_text_tokens), time_tokenize(ed, duration, num_bins, num_text_tokens)]
        out += list(tx)
    return np.array(out[:max_tokens - 1] + [eos], dtype=np.int64)


def collate(seqs: Sequence[np.ndarray]) -> np.ndarray:
    """dataset/dvc_dataset.py:184-214: right-pad with 0 to the longest sequence of the batch."""
    n = max(len(s) for s in seqs)
    out = np.zeros((len(seqs), n), dtype=np.int64)
    for i, s in enumerate(seqs):
        out[i, :len(s)] = s
    return out
