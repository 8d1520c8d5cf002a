"""CPU restatement of the reference's dense-captioning evaluation: TEST INFRASTRUCTURE ONLY.

Only tests/, oracle/make_golden.py and tests/tools/eval_bench.py may import this module; the product implementation is
vidchapters_amd/evalmetrics.py.  Plain-Python loops over dicts, written to follow the reference's arithmetic step by step:

  * temporal IoU                      dvc_eval/eval_dvc.py:99-105, dvc_eval/SODA/utils.py:3-15
  * CIDEr-D scorer                    dvc_eval/pycocoevalcap/cider/cider_scorer.py:10-25 (n-gram counts), :93-104 (document frequency),
                                      :106-187 (tf-idf vectors, clipped cosine, Gaussian length penalty, x10), cider.py:25-50
  * localisation precision / recall   dvc_eval/eval_dvc.py:146-212 (tIoU and start-distance variants)
  * tIoU-matched caption score        dvc_eval/eval_dvc.py:214-302, aggregation :305-333
  * SODA_c (IoU x caption score, DP)  dvc_eval/SODA/soda.py:61-72 (matrices), :74-129 (per-video P/R/F), :148-150, :156-191 (DP);
                                      data preparation dvc_eval/SODA/dataset.py:27-85; driver dvc_eval/eval_soda.py:5-43

Pinned by tests/golden/eval_metrics.json, produced by running the reference's own modules (oracle/make_golden.py:case_eval) on seeded
synthetic predictions.  NOT pinned: tokenisation, METEOR, and the BLEU / ROUGE-L restatements (pycocoevalcap.bleu / .rouge are imported
by the reference but not vendored in it).  The reference tokenises with the Stanford PTBTokenizer jar and scores SODA
with the METEOR jar (pycocoevalcap, Java); neither jar is in the reference tree (`.MISSING_LARGE_BLOBS`) nor in this image, so the
goldens are taken with a whitespace tokenizer stub and with the reference's own `Cider` choice of SODA scorer (soda.py:224).
"""
from __future__ import annotations

import math
from collections import defaultdict
from typing import Callable, Dict, List, Sequence, Tuple


def iou(a: Sequence[float], b: Sequence[float]) -> float:
    """eval_dvc.py:99-105 (SODA/utils.py:3-15 is the same expression)."""
    inter = max(0, min(a[1], b[1]) - max(a[0], b[0]))
    union = min(max(a[1], b[1]) - min(a[0], b[0]), (a[1] - a[0]) + (b[1] - b[0]))
    return float(inter) / (union + 1e-8)


def remove_nonascii(text: str) -> str:
    return "".join(c if ord(c) < 128 else " " for c in text)


# ----------------------------------------------------------------------------------------------------------- CIDEr-D
def _ngrams(sentence: str, n: int = 4) -> Dict[tuple, int]:
    """cider_scorer.py:10-25."""
    words = sentence.split()
    counts: Dict[tuple, int] = defaultdict(int)
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            counts[tuple(words[i:i + k])] += 1
    return counts


def cider(hyps: List[str], refs: List[List[str]], n: int = 4, sigma: float = 6.0) -> Tuple[float, List[float]]:
    """CIDEr-D of hypothesis i against its reference list refs[i]; the document frequencies come from `refs` (one document per i).
    Returns (mean, per-item scores).  cider_scorer.py:93-192."""
    ctest = [_ngrams(h, n) for h in hyps]
    crefs = [[_ngrams(r, n) for r in rs] for rs in refs]
    df: Dict[tuple, float] = defaultdict(float)
    for rs in crefs:
        for ng in set(g for r in rs for g in r):
            df[ng] += 1
    ref_len = math.log(float(len(crefs)))

    def vec(cnts):
        v = [defaultdict(float) for _ in range(n)]
        norm = [0.0] * n
        length = 0
        for ng, tf in cnts.items():
            d = math.log(max(1.0, df[ng]))
            k = len(ng) - 1
            v[k][ng] = float(tf) * (ref_len - d)
            norm[k] += v[k][ng] ** 2
            if k == 1:                       # the reference counts BIGRAMS as the sentence length (cider_scorer.py:127-128)
                length += tf
        return v, [math.sqrt(x) for x in norm], length

    scores = []
    for test, rs in zip(ctest, crefs):
        vh, nh, lh = vec(test)
        total = [0.0] * n
        for r in rs:
            vr, nr, lr = vec(r)
            delta = float(lh - lr)
            for k in range(n):
                val = 0.0
                for ng in vh[k]:
                    val += min(vh[k][ng], vr[k][ng]) * vr[k][ng]
                if nh[k] != 0 and nr[k] != 0:
                    val /= nh[k] * nr[k]
                total[k] += val * math.e ** (-(delta ** 2) / (2 * sigma ** 2))
        scores.append(sum(total) / n / len(rs) * 10.0)
    return sum(scores) / len(scores), scores


# ------------------------------------------------------------------------------------------- BLEU / ROUGE-L (UNPINNED)
# The reference imports pycocoevalcap.bleu / .rouge (eval_dvc.py:21-22, eval_vc.py:2-4) but vendors neither; the two functions below restate
# the published pycocoevalcap package (BleuScorer.compute_score(option='closest'), Rouge.calc_score with beta = 1.2).  No golden pins them.
def bleu(hyps: List[str], refs: List[List[str]], n: int = 4) -> Tuple[List[float], List[List[float]]]:
    small, tiny = 1e-9, 1e-15
    tot_guess, tot_correct = [0] * n, [0] * n
    tot_test = tot_ref = 0
    per_item = [[] for _ in range(n)]
    for h, rs in zip(hyps, refs):
        hc, hl = _ngrams(h, n), len(h.split())
        maxc: Dict[tuple, int] = {}
        for r in rs:
            for g, c in _ngrams(r, n).items():
                maxc[g] = max(maxc.get(g, 0), c)
        reflen = min((abs(len(r.split()) - hl), len(r.split())) for r in rs)[1]
        guess = [max(0, hl - k) for k in range(n)]
        correct = [0] * n
        for g, c in hc.items():
            correct[len(g) - 1] += min(maxc.get(g, 0), c)
        tot_test += hl; tot_ref += reflen
        b = 1.0
        for k in range(n):
            tot_guess[k] += guess[k]; tot_correct[k] += correct[k]
            b *= (float(correct[k]) + tiny) / (float(guess[k]) + small)
            per_item[k].append(b ** (1.0 / (k + 1)))
        ratio = (hl + tiny) / (reflen + small)
        if ratio < 1:
            for k in range(n):
                per_item[k][-1] *= math.exp(1 - 1 / ratio)
    out, b = [], 1.0
    for k in range(n):
        b *= float(tot_correct[k] + tiny) / (tot_guess[k] + small)
        out.append(b ** (1.0 / (k + 1)))
    ratio = (tot_test + tiny) / (tot_ref + small)
    if ratio < 1:
        out = [x * math.exp(1 - 1 / ratio) for x in out]
    return out, per_item


def rouge_l(hyps: List[str], refs: List[List[str]], beta: float = 1.2) -> Tuple[float, List[float]]:
    def lcs(a, b):
        dp = [[0] * (len(b) + 1) for _ in range(len(a) + 1)]
        for i in range(1, len(a) + 1):
            for j in range(1, len(b) + 1):
                dp[i][j] = dp[i - 1][j - 1] + 1 if a[i - 1] == b[j - 1] else max(dp[i - 1][j], dp[i][j - 1])
        return dp[len(a)][len(b)]
    scores = []
    for h, rs in zip(hyps, refs):
        tc = h.split(" ")
        prec = [lcs(r.split(" "), tc) / float(len(tc)) for r in rs]
        rec = [lcs(r.split(" "), tc) / float(len(r.split(" "))) for r in rs]
        p, r = max(prec), max(rec)
        scores.append((1 + beta ** 2) * p * r / float(r + beta ** 2 * p) if p != 0 and r != 0 else 0.0)
    return sum(scores) / len(scores), scores


# ------------------------------------------------------------------------------------------------- eval_dvc restated
def _gt_video_ids(gts: List[dict]) -> List[str]:
    ids = set()
    for g in gts:
        ids |= set(g.keys())
    return sorted(ids)           # the reference iterates a set (order irrelevant: everything is a mean over videos)


def detection(pred: Dict[str, list], gts: List[dict], thr: float, by_distance: bool) -> Tuple[float, float]:
    """eval_dvc.py:146-178 (tIoU > thr) and :180-212 (|start - start_gt| < thr): mean precision, mean recall."""
    precision, recall = [], []
    for vid in _gt_video_ids(gts):
        if vid not in pred:
            continue
        best_p = best_r = 0.0
        for g in gts:
            if vid not in g:
                continue
            ref_cov, pred_cov = set(), set()
            for pi, p in enumerate(pred[vid]):
                for ri, rt in enumerate(g[vid]["timestamps"]):
                    hit = abs(p["timestamp"][0] - rt[0]) < thr if by_distance else iou(p["timestamp"], rt) > thr
                    if hit:
                        ref_cov.add(ri); pred_cov.add(pi)
            best_p = max(best_p, float(len(pred_cov)) / max(len(pred[vid]), 1))
            best_r = max(best_r, float(len(ref_cov)) / len(g[vid]["timestamps"]))
        precision.append(best_p); recall.append(best_r)
    return sum(precision) / len(precision), sum(recall) / len(recall)


def caption_score_at_tiou(pred: Dict[str, list], gts: List[dict], tiou: float, tokenize: Callable[[str], str], scorer=None):
    """eval_dvc.py:214-302 with the CIDEr scorer: every (prediction, ground-truth) pair with IoU >= tiou is one item; a prediction
    without any match is scored against a garbage reference (a random 10-20 letter word in the reference; here a unique token);
    CIDEr is computed PER VIDEO over its items (document frequencies from that video's items only) and averaged over videos."""
    per_video = []
    for vid in _gt_video_ids(gts):
        if vid not in pred:
            continue
        hyps, refs = [], []
        for pi, p in enumerate(pred[vid]):
            added = False
            for g in gts:
                if vid not in g:
                    continue
                for ci, ct in enumerate(g[vid]["timestamps"]):
                    if iou(p["timestamp"], ct) >= tiou:
                        hyps.append(tokenize(remove_nonascii(p["sentence"])))
                        refs.append([tokenize(remove_nonascii(g[vid]["sentences"][ci]))])
                        added = True
            if not added:
                hyps.append(tokenize(remove_nonascii(p["sentence"])))
                refs.append([f"zzgarbage{pi}qq"])
        if scorer is None:
            per_video.append(cider(hyps, refs)[0] if hyps else 0.0)
        else:                                   # unpinned scorers (BLEU: list of 4, ROUGE-L: float), eval_dvc.py:283-301
            per_video.append(scorer(hyps, refs)[0] if hyps else ([0.0] * 4 if scorer is bleu else 0.0))
    if scorer is bleu:
        return [sum(v[k] for v in per_video) / len(per_video) for k in range(4)]
    return sum(per_video) / len(per_video)


def eval_dvc(submission: dict, references: List[dict], tokenize: Callable[[str], str], tious=(0.3, 0.5, 0.7, 0.9),
             distances=(1, 3, 5, 10, 30, 60), max_proposals_per_video: int = 1000) -> Dict[str, float]:
    """eval_dvc.py:305-333 with CIDEr as the only language scorer."""
    pred = {v: r[:max_proposals_per_video] for v, r in submission["results"].items()}
    out: Dict[str, float] = {}
    cid = [caption_score_at_tiou(pred, references, t, tokenize) for t in tious]
    out["CIDEr"] = sum(cid) / len(cid)
    bl = [caption_score_at_tiou(pred, references, t, tokenize, bleu) for t in tious]
    for k in range(4):
        out[f"Bleu_{k + 1}"] = sum(b[k] for b in bl) / len(bl)
    rg = [caption_score_at_tiou(pred, references, t, tokenize, rouge_l) for t in tious]
    out["Rouge-L"] = sum(rg) / len(rg)
    P, Rc, F = [], [], []
    for t, by_distance in [(t, False) for t in tious] + [(d, True) for d in distances]:
        p, r = detection(pred, references, t, by_distance)
        P.append(p); Rc.append(r); F.append(2 * r * p / (r + p) if r + p else 0.0)
    for i, t in enumerate(tious):
        out[f"Recall@{t}"], out[f"Precision@{t}"], out[f"F1@{t}"] = Rc[i], P[i], F[i]
    out["Recall"], out["Precision"], out["F1"] = sum(Rc[:4]) / 4, sum(P[:4]) / 4, sum(F[:4]) / 4
    for i, d in enumerate(distances):
        j = len(tious) + i
        out[f"Recall@{d}s"], out[f"Precision@{d}s"], out[f"F1@{d}s"] = Rc[j], P[j], F[j]
    return out


# ----------------------------------------------------------------------------------------------------------- SODA_c
def dp_assignment(scores: List[List[float]]) -> float:
    """soda.py:156-191: best order-preserving one-to-one matching, dp[i][j] = max(dp[i-1][j], dp[i][j-1], dp[i-1][j-1] + s[i][j]);
    the first row / column hold the best single entry so far."""
    M, N = len(scores), len(scores[0])
    dp = [[0.0] * N for _ in range(M)]
    for i in range(M):
        for j in range(N):
            if i == 0 and j == 0:
                dp[i][j] = max(-1, -1, scores[0][0])
            elif i == 0:
                dp[i][j] = max(-1, dp[0][j - 1], scores[0][j])
            elif j == 0:
                dp[i][j] = max(dp[i - 1][0], -1, scores[i][0])
            else:
                dp[i][j] = max(dp[i - 1][j], dp[i][j - 1], dp[i - 1][j - 1] + scores[i][j])
    return dp[M - 1][N - 1]


def soda_c(submission: dict, reference: dict, tokenize: Callable[[str], str]) -> Tuple[float, float, float]:
    """eval_soda.py:5-33 + soda.py:74-129 for ONE reference file, scorer `Cider`: predictions and ground truths sorted by start time
    (dataset.py:62, :80), score matrix[g][p] = CIDEr with the gold sentence as the hypothesis and the predictions as the reference
    corpus (soda.py:66-72: the arguments of compute_score are swapped there), F-measure of the DP optimum of IoU x score per video,
    means over the videos that have predictions."""
    P, Rc, F = [], [], []
    for vid in sorted(set(reference) & set(submission["results"])):
        pr = sorted(submission["results"][vid], key=lambda x: x["timestamp"][0])
        ts, ss = zip(*sorted(zip(reference[vid]["timestamps"], reference[vid]["sentences"]), key=lambda x: x[0][0]))
        if not pr:
            P.append(0.0); Rc.append(0.0); F.append(0.0)
            continue
        ptok = [tokenize(remove_nonascii(p["sentence"])) for p in pr]
        gtok = [tokenize(remove_nonascii(s)) for s in ss]
        mat = []
        for g, gt_t in zip(gtok, ts):
            sc = cider([g] * len(ptok), [[p] for p in ptok])[1]
            mat.append([iou(p["timestamp"], gt_t) * s for p, s in zip(pr, sc)])
        best = dp_assignment(mat)
        p, r = best / len(pr), best / len(gtok)
        P.append(p); Rc.append(r); F.append(2 * p * r / (p + r) if p + r > 0 else 0.0)
    return sum(P) / len(P), sum(Rc) / len(Rc), sum(F) / len(F)


def eval_soda(submission: dict, references: List[dict], tokenize: Callable[[str], str]) -> Dict[str, float]:
    """eval_soda.py:35-43: mean over the reference files of the F-measure."""
    f = [soda_c(submission, ref, tokenize)[2] for ref in references]
    return {"soda_c": sum(f) / len(f)}
