"""Dense-captioning evaluation without Java: the host-side counterpart of the reference's `dvc_eval` package.

Mirrors the callers' surface -- `eval_dvc(...)` and `eval_soda(...)` as used at dvc.py:232-233, `COCOEvalCap(results).evaluate()` as
used at vc.py:169-170 -- as ONE data-parallel job over the whole prediction file instead of per-video, per-n-gram Python dict loops:

  * every distinct sentence is tokenised once; its n-grams (orders 1-4) are ranked with `np.unique` on flat integer arrays and stored as
    one CSR count matrix per order (`_ngram_csr`);
  * all (prediction, ground truth) pairs of all videos and annotation files are enumerated as flat index arrays; one vectorised IoU gives
    the localisation precision / recall / F1 at every tIoU and start-distance threshold (dvc_eval/eval_dvc.py:99-212, :305-333) and the
    tIoU-matched caption items (eval_dvc.py:214-302);
  * CIDEr-D (dvc_eval/pycocoevalcap/cider/cider_scorer.py) of ALL items of ALL (video, tIoU) groups is one sparse computation: the
    reference's per-video document frequencies become counts over (group, n-gram) keys, the clipped cosine an intersection of
    (item, n-gram) keys (`_cider_batch`);
  * SODA_c (dvc_eval/SODA/soda.py:61-191, dataset.py, eval_soda.py): the (gold, prediction) score matrices of all videos come out of the
    same batch; the order-preserving matching is a row-wise dynamic program (`np.maximum.accumulate` per row).

What is NOT reproduced, because the reference itself shells out to Java for it and the jars are neither in the reference tree
(`.MISSING_LARGE_BLOBS`) nor in this image: full METEOR (its exact + stem stages are restated in meteor_lite.py and reported as `METEOR-lite` /
`soda_c_meteor_lite`; WordNet synonyms, paraphrase tables and the function-word discount are not), and the Stanford PTB tokenizer.  `ptb_tokenize` below is an approximation of the
latter (lower-casing, punctuation split off and dropped like pycocoevalcap's PUNCTUATIONS list); pass `tokenize=` to use another one.
SODA_c is therefore computed with the reference's own alternative scorer choice `Cider` (soda.py:224) and reported under the key
`soda_c_cider` (or with METEOR-lite under `soda_c_meteor_lite`); no key is called METEOR or soda_c unless the caller brings a scorer.  BLEU-1..4 and ROUGE-L are computed, but pycocoevalcap's bleu/ and rouge/ are not
vendored in the reference either: they are restated from the published package and their parity is UNPINNED (hand-checked cases and
the plain-Python restatement in the oracle only).  Parity with the reference's modules (run on
pre-tokenised text): tests/golden/eval_metrics.json, tests/test_oracle_cpu.py.
"""
from __future__ import annotations

import json
import re
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

# ------------------------------------------------------------------------------------------------------------ text
_PUNCT = {"''", "'", "``", "`", "-lrb-", "-rrb-", "-lcb-", "-rcb-", ".", "?", "!", ",", ":", "-", "--", "...", ";"}
_TOKEN = re.compile(r"\.\.\.|--|``|''|n't\b|'(?:s|re|ve|m|ll|d)\b|[A-Za-z0-9]+(?:[.,'/-][A-Za-z0-9]+)*|\S")
_BRACKETS = {"(": "-lrb-", ")": "-rrb-", "{": "-lcb-", "}": "-rcb-", "[": "-lsb-", "]": "-rsb-", '"': "''"}


def remove_nonascii(text: str) -> str:
    return text if text.isascii() else "".join(c if ord(c) < 128 else " " for c in text)


def ptb_tokenize(text: str) -> str:
    """Approximation of pycocoevalcap's PTBTokenizer (Stanford PTBTokenizer -lowerCase, then its PUNCTUATIONS list removed)."""
    text = re.sub(r"(\w)n't\b", r"\1 n't", text.lower())
    text = re.sub(r"(\w)('(?:s|re|ve|m|ll|d))\b", r"\1 \2", text)
    out = []
    for t in _TOKEN.findall(text):
        t = _BRACKETS.get(t, t)
        if t not in _PUNCT:
            out.append(t)
    return " ".join(out)


# ------------------------------------------------------------------------------------------------------------ CIDEr-D
def _ngram_csr(sentences: Sequence[str], n: int = 4):
    """sentences: tokenised strings.  Returns (mats, bigrams, words): mats[k-1] = CSR count matrix [len(sentences) + 1, V_k] of the k-grams
    (the extra last row is empty: the 'garbage' reference of eval_dvc.py:253-259), bigrams[s] = number of 2-grams of sentence s, which is
    what the reference uses as the sentence length (cider_scorer.py:127-128).  No per-n-gram Python work: word ids -> k-gram keys
    rank(k-1 gram) * V + next word, re-ranked by np.unique at every order so that the keys stay below 2^62."""
    import scipy.sparse as sp
    vocab: Dict[str, int] = {}
    words = [s.split() for s in sentences]
    lens = np.fromiter((len(w) for w in words), np.int64, len(words))
    flat = np.fromiter((vocab.setdefault(w, len(vocab)) for ws in words for w in ws), np.int64, int(lens.sum()))
    S, V = len(words), max(len(vocab), 1)
    sid = np.repeat(np.arange(S), lens)
    pos = np.arange(len(flat)) - np.repeat(np.cumsum(lens) - lens, lens)
    room = lens[sid] - pos                           # words left in the sentence from this position on
    mats, rank = [], flat
    for k in range(1, n + 1):
        idx = np.nonzero(room >= k)[0]
        comb = flat[idx] if k == 1 else rank[idx] * V + flat[idx + k - 1]
        uniq, inv = np.unique(comb, return_inverse=True)
        rank = np.zeros(len(flat), np.int64)
        rank[idx] = inv
        mats.append(sp.coo_matrix((np.ones(len(idx)), (sid[idx], inv)), shape=(S + 1, max(len(uniq), 1))).tocsr())   # duplicates are summed
    words = (flat, np.cumsum(lens) - lens, lens)                    # word ids, sentence starts and lengths (BLEU lengths, LCS of ROUGE-L)
    return mats, np.concatenate((np.maximum(lens - 1, 0), [0])).astype(np.float64), words


def _cider_batch(mats, bigrams: np.ndarray, h_rows: np.ndarray, r_rows: np.ndarray, item_group: np.ndarray, doc_rows: np.ndarray,
                 doc_id: np.ndarray, doc_group: np.ndarray, n_docs: np.ndarray, sigma: float = 6.0) -> np.ndarray:
    """CIDEr-D similarity (x10, mean over orders) of N items (hypothesis row h_rows[i] vs reference row r_rows[i] of the count matrices)
    that belong to groups: the tf-idf weights of an item use the document frequencies of ITS group, counted over that group's documents
    (rows doc_rows; rows sharing a doc_id form one document: the references of one item in cider_scorer.py:93-104) of which there are
    n_docs[group].  cider_scorer.py:106-187, all items at once."""
    N = len(h_rows)
    penalty = np.e ** (-((bigrams[h_rows] - bigrams[r_rows]) ** 2) / (2 * sigma ** 2))
    ref_len = np.log(np.maximum(n_docs, 1).astype(np.float64))
    total = np.zeros(N)
    for S in mats:
        NC = S.shape[1]
        H, R, D = S[h_rows].tocoo(), S[r_rows].tocoo(), S[doc_rows].tocoo()
        dkey = np.unique(doc_id[D.row].astype(np.int64) * NC + D.col)                   # one entry per (document, n-gram)
        gkey, df = _group_df(dkey, NC, doc_id, doc_group)

        def weights(M):
            g = item_group[M.row]
            key = g * NC + M.col
            at = np.minimum(np.searchsorted(gkey, key), max(len(gkey) - 1, 0))
            d = np.where(gkey[at] == key, df[at], 0) if len(gkey) else np.zeros(len(key))
            return M.data * (ref_len[g] - np.log(np.maximum(1.0, d)))

        hw, rw = weights(H), weights(R)
        nh = np.sqrt(np.bincount(H.row, hw ** 2, N).astype(np.float64))
        nr = np.sqrt(np.bincount(R.row, rw ** 2, N).astype(np.float64))
        _, ih, ir = np.intersect1d(H.row.astype(np.int64) * NC + H.col, R.row.astype(np.int64) * NC + R.col, assume_unique=True, return_indices=True)
        val = np.bincount(H.row[ih], np.minimum(hw[ih], rw[ir]) * rw[ir], N).astype(np.float64)
        den = nh * nr
        total += np.divide(val, den, out=val.copy(), where=den != 0) * penalty
    return total / len(mats) * 10.0


def _group_df(dkey: np.ndarray, NC: int, doc_id: np.ndarray, doc_group: np.ndarray):
    """dkey = doc * NC + ngram (one entry per document and n-gram) -> sorted keys group * NC + ngram and their document counts."""
    group_of_doc = np.zeros(int(doc_id.max()) + 1 if len(doc_id) else 1, np.int64)
    group_of_doc[doc_id] = doc_group
    return np.unique(group_of_doc[dkey // NC] * NC + dkey % NC, return_counts=True)


def _bleu_batch(mats, words, h_rows: np.ndarray, r_rows: np.ndarray, item_group: np.ndarray, n_groups: int) -> np.ndarray:
    """Corpus-level BLEU-1..n of every group over its (hypothesis, single reference) items, [n_groups, n]: clipped n-gram matches and
    guesses summed per group, brevity penalty on the summed lengths -- pycocoevalcap's BleuScorer.compute_score(option='closest') as
    called through Bleu(4).compute_score at eval_dvc.py:286 / eval_vc.py:60 (the scorer itself is not vendored in the reference:
    restated from the published package, parity unpinned).  A garbage reference (row = last) is one word long and matches nothing."""
    n = len(mats)
    lens = np.concatenate((words[2], [1])).astype(np.float64)
    N = len(h_rows)
    correct = np.zeros((n_groups, n)); guess = np.zeros((n_groups, n))
    for k, S in enumerate(mats):
        NC = S.shape[1]
        H, R = S[h_rows].tocoo(), S[r_rows].tocoo()
        _, ih, ir = np.intersect1d(H.row.astype(np.int64) * NC + H.col, R.row.astype(np.int64) * NC + R.col, assume_unique=True, return_indices=True)
        correct[:, k] = np.bincount(item_group[H.row[ih]], np.minimum(H.data[ih], R.data[ir]), n_groups)
        guess[:, k] = np.bincount(item_group, np.maximum(0.0, lens[h_rows] - k), n_groups)
    testlen = np.bincount(item_group, lens[h_rows], n_groups)
    reflen = np.bincount(item_group, lens[r_rows], n_groups)
    bleus = np.cumprod((correct + 1e-15) / (guess + 1e-9), axis=1) ** (1.0 / np.arange(1, n + 1))
    ratio = (testlen + 1e-15) / (reflen + 1e-9)
    return bleus * np.where(ratio < 1, np.exp(1 - 1 / ratio), 1.0)[:, None]


def _rouge_batch(words, h_rows: np.ndarray, r_rows: np.ndarray, beta: float = 1.2) -> np.ndarray:
    """ROUGE-L F-score (beta = 1.2) of N (hypothesis, single reference) items: LCS by a dynamic program that advances all items
    together (one vector step per hypothesis position; the dependence along the reference is a running maximum).  pycocoevalcap's
    Rouge.calc_score (not vendored in the reference: restated from the published package, parity unpinned), incl. its
    `split(" ")`: an empty sentence is one empty token."""
    flat, starts, lens = words
    S = len(lens)

    def padded(rows, fill):
        ln = np.where(rows < S, lens[np.minimum(rows, S - 1)], 1)             # garbage row: one word that matches nothing
        L = max(int(ln.max()) if len(ln) else 1, 1)
        out = np.full((len(rows), L), fill, np.int64)
        j = np.arange(L)[None, :]
        ok = (j < ln[:, None]) & (rows < S)[:, None]
        src = np.minimum(starts[np.minimum(rows, S - 1)][:, None] + j, max(len(flat) - 1, 0))
        if len(flat):
            out[ok] = flat[src[ok]]
        empty = (rows < S) & (ln == 0)
        out[empty, 0] = -3                                                      # '' == '' (split(" ") of an empty string)
        out[rows >= S, 0] = -4
        return out, np.maximum(ln, 1).astype(np.float64)

    Hs, lh = padded(h_rows, -1)
    Rs, lr = padded(r_rows, -2)
    dp = np.zeros((len(h_rows), Rs.shape[1] + 1))
    for i in range(Hs.shape[1]):
        eq = Hs[:, i:i + 1] == Rs
        dp[:, 1:] = np.maximum.accumulate(np.maximum(dp[:, 1:], np.where(eq, dp[:, :-1] + 1, 0.0)), axis=1)
    lcs = dp[:, -1]
    p, r = lcs / lh, lcs / lr
    den = r + beta ** 2 * p
    return np.where((p != 0) & (r != 0), (1 + beta ** 2) * p * r / np.where(den != 0, den, 1.0), 0.0)


class Cider:
    """pycocoevalcap-compatible scorer object (dvc_eval/pycocoevalcap/cider/cider.py:12-53): compute_score(gts, res) with
    gts[id] = list of tokenised reference strings, res[id] = [tokenised hypothesis]; one corpus, any number of references per item."""

    def __init__(self, n: int = 4, sigma: float = 6.0):
        self._n, self._sigma = n, sigma

    def method(self) -> str:
        return "CIDEr"

    def compute_score(self, gts: Dict, res: Dict) -> Tuple[float, np.ndarray]:
        assert gts.keys() == res.keys()
        ids = list(gts.keys())
        for i in ids:
            assert type(res[i]) is list and len(res[i]) == 1
            assert type(gts[i]) is list and len(gts[i]) > 0
        H = len(ids)
        flat = [r for i in ids for r in gts[i]]
        owner = np.repeat(np.arange(H), [len(gts[i]) for i in ids])
        mats, bigrams, _ = _ngram_csr([res[i][0] for i in ids] + flat, self._n)
        r_rows = H + np.arange(len(flat))
        zeros = np.zeros(len(flat), np.int64)
        pair = _cider_batch(mats, bigrams, owner, r_rows, zeros, r_rows, owner, zeros, np.array([H]), self._sigma)
        scores = np.bincount(owner, pair, H) / np.bincount(owner, minlength=H)
        return float(scores.mean()), scores


def _single_ref_corpus(gts: Dict, res: Dict):
    ids = list(gts.keys())
    assert gts.keys() == res.keys()
    for i in ids:
        assert type(res[i]) is list and len(res[i]) == 1
        assert type(gts[i]) is list and len(gts[i]) == 1, "the batched BLEU / ROUGE-L scorers take one reference per item (all call sites of the reference do)"
    H = len(ids)
    mats, _, words = _ngram_csr([res[i][0] for i in ids] + [gts[i][0] for i in ids])
    return H, mats, words


class Bleu:
    """pycocoevalcap-compatible Bleu(n).compute_score(gts, res) -> ([BLEU-1..n], per-item lists); one reference per item."""

    def __init__(self, n: int = 4):
        self._n = n

    def method(self) -> str:
        return "Bleu"

    def compute_score(self, gts: Dict, res: Dict):
        H, mats, words = _single_ref_corpus(gts, res)
        h, r = np.arange(H), H + np.arange(H)
        corpus = _bleu_batch(mats[:self._n], words, h, r, np.zeros(H, np.int64), 1)[0]
        per_item = _bleu_batch(mats[:self._n], words, h, r, h, H)
        return [float(x) for x in corpus], [per_item[:, k].tolist() for k in range(self._n)]


class Rouge:
    """pycocoevalcap-compatible Rouge().compute_score(gts, res) -> (mean ROUGE-L, per-item array); one reference per item."""

    def method(self) -> str:
        return "Rouge"

    def compute_score(self, gts: Dict, res: Dict):
        H, _, words = _single_ref_corpus(gts, res)
        sc = _rouge_batch(words, np.arange(H), H + np.arange(H))
        return float(sc.mean()), sc


# ------------------------------------------------------------------------------------------------------------ localisation
def _iou(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Element-wise temporal IoU of interval arrays [..., 2]; eval_dvc.py:99-105 incl. the 1e-8 in the denominator."""
    inter = np.maximum(0.0, np.minimum(a[..., 1], b[..., 1]) - np.maximum(a[..., 0], b[..., 0]))
    union = np.minimum(np.maximum(a[..., 1], b[..., 1]) - np.minimum(a[..., 0], b[..., 0]), (a[..., 1] - a[..., 0]) + (b[..., 1] - b[..., 0]))
    return inter / (union + 1e-8)


def iou_matrix(a, b) -> np.ndarray:
    """[len(a), len(b)] temporal IoU of intervals (start, end)."""
    a, b = np.asarray(a, np.float64).reshape(-1, 2), np.asarray(b, np.float64).reshape(-1, 2)
    return _iou(a[:, None, :], b[None, :, :])


def _load(x):
    return x if isinstance(x, dict) else json.load(open(x))


class _Sentences:
    """Interns tokenised sentences: raw string -> row of the n-gram matrices (each distinct raw string is tokenised once)."""

    def __init__(self, tokenize):
        self.tokenize, self.raw, self.rows = tokenize, {}, []

    def __call__(self, text: str) -> int:
        r = self.raw.get(text)
        if r is None:
            r = self.raw[text] = len(self.rows)
            self.rows.append(self.tokenize(remove_nonascii(text)))
        return r


def _blocks(n_a: np.ndarray, n_b: np.ndarray):
    """For blocks with n_a[i] x n_b[i] pairs: (block, a, b) index of every pair, a and b counted from the start of their block."""
    cnt = n_a * n_b
    blk = np.repeat(np.arange(len(cnt)), cnt)
    within = np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    nb = np.maximum(n_b[blk], 1)
    return blk, within // nb, within % nb


def eval_dvc(submission, references, tious=[0.3, 0.5, 0.7, 0.9], distances=[1, 3, 5, 10, 30, 60], max_proposals_per_video=1000,
             verbose=False, no_lang_eval=False, tokenize: Optional[Callable[[str], str]] = None, meteor_lite: bool = False) -> Dict[str, float]:
    """dvc_eval/eval_dvc.py:305-333.  `submission`: {"results": {vid: [{"sentence", "timestamp": [s, e]}]}} or a json path;
    `references`: annotation dicts ({vid: {"timestamps", "sentences"}}) or json paths.  Returns CIDEr (mean over `tious` of the
    per-video CIDEr of the tIoU-matched pairs) and Recall / Precision / F1 @tIoU, their means over the first four thresholds, and
    @<d>s for the start-distance thresholds.  Raises ZeroDivisionError like the reference if no video has predictions."""
    if len(tious) == 0:
        raise IOError("Please input a valid tIoU.")
    if not references:
        raise IOError("Please input a valid ground truth file.")
    if not submission:
        raise IOError("Please input a valid prediction file.")
    sents = _Sentences(tokenize or ptb_tokenize)
    gts = [_load(r) for r in references]
    pred = {v: r[:max_proposals_per_video] for v, r in _load(submission)["results"].items()}
    vids = sorted(v for v in set().union(*[set(g) for g in gts]) if v in pred)
    if verbose:
        print("available video number", len(vids))
    if not vids:                 # eval_dvc.py:178: sum(precision) / len(precision) over an empty list
        raise ZeroDivisionError("division by zero")
    NV = len(vids)
    # flat predictions, and one block per (video, annotation file that has the video) with its flat ground truths
    p_cnt = np.array([len(pred[v]) for v in vids])
    p_off = np.cumsum(p_cnt) - p_cnt
    p_ts = np.array([p["timestamp"] for v in vids for p in pred[v]], np.float64).reshape(-1, 2)
    p_vid = np.repeat(np.arange(NV), p_cnt)
    p_row = np.array([sents(p["sentence"]) for v in vids for p in pred[v]], np.int64) if not no_lang_eval else None
    b_vid, g_cnt, g_ts, g_row = [], [], [], []
    for g in gts:
        for vi, v in enumerate(vids):
            if v in g:
                b_vid.append(vi); g_cnt.append(len(g[v]["timestamps"])); g_ts += list(g[v]["timestamps"])
                if not no_lang_eval:
                    g_row += [sents(s) for s in g[v]["sentences"]]
    b_vid, g_cnt = np.array(b_vid), np.array(g_cnt)
    g_off = np.cumsum(g_cnt) - g_cnt
    g_ts = np.array(g_ts, np.float64).reshape(-1, 2)
    blk, pa, gb = _blocks(p_cnt[b_vid], g_cnt)
    pi, gi = p_off[b_vid][blk] + pa, g_off[blk] + gb                      # global prediction / ground-truth index of every pair
    iou = _iou(p_ts[pi], g_ts[gi])
    dist = np.abs(p_ts[pi, 0] - g_ts[gi, 0])
    pslot = (np.cumsum(p_cnt[b_vid]) - p_cnt[b_vid])[blk] + pa             # (block, prediction) slot: coverage is per annotation file
    n_pslot, NB = int(p_cnt[b_vid].sum()), len(b_vid)
    slot_blk = np.repeat(np.arange(NB), p_cnt[b_vid])
    g_blk = np.repeat(np.arange(NB), g_cnt)

    P, R = [], []
    for thr, by_dist in [(t, False) for t in tious] + [(d, True) for d in distances]:
        hit = dist < thr if by_dist else iou > thr
        pc = np.bincount(slot_blk[np.unique(pslot[hit])], minlength=NB) / np.maximum(p_cnt[b_vid], 1)
        rc = np.bincount(g_blk[np.unique(gi[hit])], minlength=NB) / g_cnt
        pv, rv = np.zeros(NV), np.zeros(NV)
        np.maximum.at(pv, b_vid, pc); np.maximum.at(rv, b_vid, rc)       # best annotation file per video (eval_dvc.py:175-177)
        P.append(pv.mean()); R.append(rv.mean())
    P, R = np.array(P), np.array(R)
    F = np.where(R + P > 0, 2 * R * P / np.where(R + P > 0, R + P, 1.0), 0.0)

    out: Dict[str, float] = {}
    if not no_lang_eval:
        # items of group (tIoU t, video v): every pair with IoU >= t, plus (prediction, garbage) for predictions without any such pair
        garbage = len(sents.rows)                                         # the empty extra row of the count matrices
        g_rows = np.asarray(g_row, np.int64)
        h, r, grp = [], [], []
        for ti, t in enumerate(tious):
            m = iou >= t
            lone = np.ones(len(p_ts), bool)
            lone[pi[m]] = False
            lone = np.nonzero(lone)[0]
            h += [p_row[pi[m]], p_row[lone]]
            r += [g_rows[gi[m]], np.full(len(lone), garbage)]
            grp += [ti * NV + p_vid[pi[m]], ti * NV + p_vid[lone]]
        h, r, grp = np.concatenate(h), np.concatenate(r), np.concatenate(grp)
        n_items = np.bincount(grp, minlength=len(tious) * NV)
        mats, bigrams, words = _ngram_csr(sents.rows)
        sc = _cider_batch(mats, bigrams, h, r, grp, r, np.arange(len(r)), grp, n_items)
        per_group = np.bincount(grp, sc, len(tious) * NV) / np.maximum(n_items, 1)          # videos without predictions score 0
        out["CIDEr"] = float(per_group.reshape(len(tious), NV).mean(1).mean())
        # BLEU-1..4 (corpus-level per video) and ROUGE-L (mean per video), averaged like CIDEr (eval_dvc.py:283-301); unpinned restatements
        bl = _bleu_batch(mats, words, h, r, grp, len(tious) * NV) * (n_items > 0)[:, None]
        for k in range(4):
            out[f"Bleu_{k + 1}"] = float(bl[:, k].reshape(len(tious), NV).mean(1).mean())
        rg = np.bincount(grp, _rouge_batch(words, h, r), len(tious) * NV) / np.maximum(n_items, 1)
        out["Rouge-L"] = float(rg.reshape(len(tious), NV).mean(1).mean())
        if meteor_lite:
            # METEOR (eval_dvc.py:67) restated without the jar for its exact + stem stages only (meteor_lite.py: labelled, unpinned): per
            # (tIoU, video) group the score of the group's SUMMED alignment statistics, like Meteor.compute_score's first return value
            # OPT-IN: a pure-Python beam alignment, ~0.7 ms per distinct (hypothesis, reference) pair, inside an evaluation that is
            # otherwise one vectorised job; the pair cache lives at module level and is reused across calls
            from . import meteor_lite as ML
            ml = ML.shared_scorer()
            pairs, inv = np.unique(np.stack([h, r], 1), axis=0, return_inverse=True)
            st = np.array([ml.pair_stats(sents.rows[a], sents.rows[b] if b != garbage else "#garbage#") for a, b in pairs], np.float64).reshape(-1, 5)
            agg = np.stack([np.bincount(grp, st[inv.reshape(-1), k], len(tious) * NV) for k in range(5)], 1)
            mg = np.array([ML._score(x) for x in agg]) * (n_items > 0)
            out["METEOR-lite"] = float(mg.reshape(len(tious), NV).mean(1).mean())
    for i, x in enumerate(tious):
        out[f"Recall@{x}"], out[f"Precision@{x}"], out[f"F1@{x}"] = float(R[i]), float(P[i]), float(F[i])
    out["Recall"], out["Precision"], out["F1"] = float(R[:4].mean()), float(P[:4].mean()), float(F[:4].mean())
    for i, x in enumerate(distances):
        j = len(tious) + i
        out[f"Recall@{x}s"], out[f"Precision@{x}s"], out[f"F1@{x}s"] = float(R[j]), float(P[j]), float(F[j])
    return out


# ------------------------------------------------------------------------------------------------------------ SODA
def dp_assignment(scores: np.ndarray) -> float:
    """Value of the best order-preserving one-to-one matching of rows to columns (soda.py:156-191):
    dp[i, j] = max(dp[i-1, j], dp[i, j-1], dp[i-1, j-1] + s[i, j]).  The dependence on dp[i, j-1] is a running maximum, so each row
    is one vector expression."""
    s = np.asarray(scores, np.float64)
    dp = np.maximum.accumulate(np.maximum(s[0], -1.0))
    for i in range(1, s.shape[0]):
        cand = np.maximum(dp, np.concatenate(([-1.0], dp[:-1] + s[i, 1:])))
        cand[0] = max(cand[0], s[i, 0])
        dp = np.maximum.accumulate(cand)
    return float(dp[-1])


def soda_c(submission, reference, tokenize: Optional[Callable[[str], str]] = None, scorer=None) -> Tuple[float, float, float]:
    """Mean (precision, recall, F1) of SODA_c against ONE annotation file (eval_soda.py:5-33, soda.py:74-129): per video, predictions
    and ground truths sorted by start time, F-measure of the optimum of sum(IoU x caption score) over order-preserving matchings.
    scorer=None: CIDEr-D with the prediction set of the video as the reference corpus (what soda.py:66-72 computes for `Cider`), all
    videos in one batch; otherwise an object with pycocoevalcap's compute_score(gts, res), called exactly like soda.py:66-72 calls it."""
    sents = _Sentences(tokenize or ptb_tokenize)
    sub, ref = _load(submission)["results"], _load(reference)
    vids = sorted(set(ref) & set(sub))
    NV = len(vids)
    prs = [sorted(sub[v], key=lambda x: x["timestamp"][0]) for v in vids]
    orders = [sorted(range(len(ref[v]["timestamps"])), key=lambda i: ref[v]["timestamps"][i][0]) for v in vids]
    p_cnt, g_cnt = np.array([len(p) for p in prs]), np.array([len(o) for o in orders])
    p_off, g_off = np.cumsum(p_cnt) - p_cnt, np.cumsum(g_cnt) - g_cnt
    p_ts = np.array([p["timestamp"] for pr in prs for p in pr], np.float64).reshape(-1, 2)
    g_ts = np.array([ref[v]["timestamps"][i] for v, o in zip(vids, orders) for i in o], np.float64).reshape(-1, 2)
    p_row = np.array([sents(p["sentence"]) for pr in prs for p in pr], np.int64)
    g_row = np.array([sents(ref[v]["sentences"][i]) for v, o in zip(vids, orders) for i in o], np.int64)
    blk, ga, pb = _blocks(g_cnt, p_cnt)                                   # every (gold, prediction) combination of every video
    gi, pi = g_off[blk] + ga, p_off[blk] + pb
    weight = _iou(p_ts[pi], g_ts[gi])
    if scorer is None:
        mats, bigrams, _ = _ngram_csr(sents.rows)
        p_vid = np.repeat(np.arange(NV), p_cnt)
        weight = weight * _cider_batch(mats, bigrams, g_row[gi], p_row[pi], blk, p_row, np.arange(len(p_row)), p_vid, p_cnt)
    pair_off = np.cumsum(g_cnt * p_cnt) - g_cnt * p_cnt
    P, R, F = np.zeros(NV), np.zeros(NV), np.zeros(NV)
    for v in range(NV):
        if p_cnt[v] == 0:                                                 # soda.py:88-93: no predictions -> zeros
            continue
        m = weight[pair_off[v]: pair_off[v] + g_cnt[v] * p_cnt[v]].reshape(g_cnt[v], p_cnt[v])
        if scorer is not None:
            ptok = [sents.rows[i] for i in p_row[p_off[v]: p_off[v] + p_cnt[v]]]
            res = {i: [p] for i, p in enumerate(ptok)}
            m = m * np.array([scorer.compute_score(res, {i: [sents.rows[g]] for i in range(len(ptok))})[1]
                              for g in g_row[g_off[v]: g_off[v] + g_cnt[v]]])
        best = dp_assignment(m)
        p, r = best / p_cnt[v], best / g_cnt[v]
        P[v], R[v], F[v] = p, r, (2 * p * r / (p + r) if p + r > 0 else 0.0)
    return float(P.mean()), float(R.mean()), float(F.mean())


def eval_soda(p, ref_list, verbose=False, tokenize: Optional[Callable[[str], str]] = None, scorer=None) -> Dict[str, float]:
    """dvc_eval/eval_soda.py:35-43: mean over the annotation files of the SODA_c F-measure.  The key is `soda_c_cider` with the
    built-in CIDEr scorer, `soda_c_meteor_lite` with scorer="meteor_lite" (meteor_lite.py: METEOR's exact + stem stages, no jar) and
    `soda_c` when the caller supplies a scorer object (e.g. their own METEOR wrapper)."""
    key = "soda_c_cider" if scorer is None else "soda_c"
    if isinstance(scorer, str):
        if scorer != "meteor_lite":
            raise ValueError(f"unknown scorer {scorer!r}")
        from .meteor_lite import MeteorLite
        scorer, key = MeteorLite(), "soda_c_meteor_lite"          # the README's SODA_c is IoU x METEOR: this is its exact + stem restatement, labelled
    f = float(np.mean([soda_c(p, ref, tokenize, scorer)[2] for ref in ref_list]))
    return {key: f}


# ------------------------------------------------------------------------------------------------------------ vc.py
class COCOEvalCap:
    """dvc_eval/eval_vc.py:7-79 as used by vc.py:169-170: results = {id: {"sentence": prediction, "gt": ground truth}}; one corpus-level
    CIDEr over all items.  (The reference stores the PREDICTION as `gts` and the ground truth as `res`, eval_vc.py:16-23; kept.)"""

    def __init__(self, results: Dict, tokenize: Optional[Callable[[str], str]] = None):
        self.tokenize = tokenize or ptb_tokenize
        self.gts = {k: [self.tokenize(r["sentence"])] for k, r in results.items()}
        self.res = {k: [self.tokenize(r["gt"])] for k, r in results.items()}
        self.eval: Dict[str, float] = {}
        self.imgToEval: Dict = {}
        self.evalImgs: List = []

    def evaluate(self) -> Dict[str, float]:
        out: Dict[str, float] = {}

        def put(name, score, scores):
            out[name] = self.eval[name] = score
            # eval_vc.py:67-72 pairs the scores with sorted(ids) although they were computed in dict order; kept
            for k, s in zip(sorted(self.gts.keys()), scores):
                self.imgToEval.setdefault(k, {"image_id": k})[name] = float(s)

        bl, bls = Bleu(4).compute_score(self.gts, self.res)
        for k in range(4):
            put(f"Bleu_{k + 1}", bl[k], bls[k])
        put("ROUGE_L", *Rouge().compute_score(self.gts, self.res))
        put("CIDEr", *Cider().compute_score(self.gts, self.res))
        self.evalImgs = [self.imgToEval[k] for k in sorted(self.imgToEval.keys())]
        return out
