"""Host side of beam search: hypothesis bookkeeping between two device steps.

Mirrors what the reference gets from ``transformers==4.28.0`` ``BeamSearchScorer``/``BeamHypotheses`` when
``Vid2Seq.generate`` (model/vid2seq.py:150-162) runs with ``num_beams>1, do_sample=False, early_stopping=False,
num_return_sequences=1``.  The device produces, for every live beam, its 2*num_beams best continuations
(``v2s_topk_logprob``); everything here is small integer/float bookkeeping on numpy arrays.

With ``do_sample=True`` (HF 4.28 ``beam_sample``) the device sends sampled candidates instead (``v2s_beam_sample_cand``: warped
scores + Gumbel keys): an entry's 2*num_beams continuations are the candidates with the largest keys (= HF's multinomial draw
without replacement from the softmax over the entry's warped scores), then ordered by score like HF's ``torch.sort`` of the draws;
all beams start with score 0 (beam_sample does not mask beams 1.. with -1e9: the sampling tells them apart).
"""
from __future__ import annotations

import numpy as np


class _Heap:
    """Keeps the num_beams best finished hypotheses of one batch entry, scored by sum_logprobs / len**length_penalty."""
    __slots__ = ("cap", "lp", "items", "worst")

    def __init__(self, cap: int, lp: float):
        self.cap, self.lp, self.items, self.worst = cap, lp, [], 1e9

    def push(self, tokens: np.ndarray, sum_logprobs: float) -> None:
        score = sum_logprobs / (len(tokens) ** self.lp)
        if len(self.items) < self.cap or score > self.worst:
            self.items.append((score, tokens))
            if len(self.items) > self.cap:
                order = sorted(range(len(self.items)), key=lambda i: (self.items[i][0], i))
                del self.items[order[0]]
                self.worst = min(s for s, _ in self.items)
            else:
                self.worst = min(score, self.worst)

    def full_and_unbeatable(self, best_sum_logprobs: float, cur_len: int) -> bool:
        return len(self.items) >= self.cap and self.worst >= best_sum_logprobs / cur_len ** self.lp

    def best(self, n: int = 1):
        """The n best hypotheses, best first; among equal scores the one added later wins (a stable ascending sort popped from
        the end, like BeamSearchScorer.finalize)."""
        order = sorted(range(len(self.items)), key=lambda i: self.items[i][0])
        return [self.items[order.pop()][1] for _ in range(n)]


class BeamScorer:
    def __init__(self, batch: int, num_beams: int, length_penalty: float, eos_id: int, pad_id: int, start_id: int, max_length: int,
                 sample: bool = False):
        self.B, self.nb, self.eos, self.pad, self.max_length = batch, num_beams, eos_id, pad_id, max_length
        self.heaps = [_Heap(num_beams, length_penalty) for _ in range(batch)]
        self.done = np.zeros(batch, dtype=bool)
        self.seqs = np.full((batch * num_beams, max_length), pad_id, dtype=np.int64)
        self.seqs[:, 0] = start_id
        self.cur_len = 1
        self.scores = np.zeros((batch, num_beams), dtype=np.float32)
        if not sample:
            self.scores[:, 1:] = -1e9

    def advance(self, cand_val: np.ndarray, cand_tok: np.ndarray, cand_key: np.ndarray = None):
        """cand_val/cand_tok: [B*nb, K] per-beam sorted candidates.  Returns (tokens [B*nb] int64, source rows [B*nb] int32,
        finished) and updates ``self.scores`` / ``self.seqs``.  ``cand_key`` (beam-sample): the 2*nb candidates of an entry are the
        ones with the largest keys, ordered by value."""
        B, nb = self.B, self.nb
        K = cand_val.shape[1]
        val = cand_val.reshape(B, nb * K)
        tok = cand_tok.reshape(B, nb * K)
        if cand_key is None:
            order = np.argsort(-val, axis=1, kind="stable")[:, :2 * nb]
        else:
            drawn = np.argsort(-cand_key.reshape(B, nb * K), axis=1, kind="stable")[:, :2 * nb]
            by_val = np.argsort(-np.take_along_axis(val, drawn, 1), axis=1, kind="stable")
            order = np.take_along_axis(drawn, by_val, 1)
        tok_sel = np.take_along_axis(tok, order, 1)
        val_sel = np.take_along_axis(val, order, 1)
        src_sel = (np.arange(B, dtype=np.int64)[:, None] * nb + order // K).astype(np.int32)
        new_tok = np.full((B, nb), self.pad, dtype=np.int64)
        new_src = np.zeros((B, nb), dtype=np.int32)
        new_sc = np.zeros((B, nb), dtype=np.float32)
        # entries with no EOS among their 2*nb best and fewer than nb finished hypotheses (nothing to push, cannot be done) take
        # the nb best as they are -- vectorised; the others walk the ranks like BeamSearchScorer.process
        heap_len = np.fromiter((len(h.items) for h in self.heaps), dtype=np.int64, count=B)
        slow = ~self.done & ((tok_sel == self.eos).any(1) | (heap_len >= nb))
        fast = ~self.done & ~slow
        new_tok[fast], new_src[fast], new_sc[fast] = tok_sel[fast, :nb], src_sel[fast, :nb], val_sel[fast, :nb]
        for b in np.nonzero(slow)[0]:
            k = 0
            for rank in range(2 * nb):
                t, sc, src = int(tok_sel[b, rank]), float(val_sel[b, rank]), int(src_sel[b, rank])
                if t == self.eos:
                    if rank < nb:
                        self.heaps[b].push(self.seqs[src, :self.cur_len].copy(), sc)
                    continue
                new_tok[b, k], new_src[b, k], new_sc[b, k] = t, src, sc
                k += 1
                if k == nb:
                    break
            if self.heaps[b].full_and_unbeatable(float(val_sel[b, 0]), self.cur_len):
                self.done[b] = True
        self.scores = new_sc
        src = new_src.reshape(-1)
        self.seqs = self.seqs[src]
        if self.cur_len < self.max_length:
            self.seqs[:, self.cur_len] = new_tok.reshape(-1)
        self.cur_len += 1
        return new_tok.reshape(-1), src, bool(self.done.all()) or self.cur_len >= self.max_length

    def finalize(self, num_return: int = 1) -> np.ndarray:
        for b in range(self.B):
            if self.done[b]:
                continue
            for j in range(self.nb):
                self.heaps[b].push(self.seqs[b * self.nb + j, :self.cur_len].copy(), float(self.scores[b, j]))
        best = [hyp for h in self.heaps for hyp in h.best(num_return)]
        out_len = min(max(len(x) for x in best) + 1, self.max_length)
        out = np.full((len(best), out_len), self.pad, dtype=np.int64)
        for b, hyp in enumerate(best):
            out[b, :len(hyp)] = hyp
            if len(hyp) < out_len:
                out[b, len(hyp)] = self.eos
        return out
