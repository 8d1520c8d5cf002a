"""METEOR-lite: the exact + stem stages of METEOR 1.5 without Java.

The reference scores captions with METEOR through `pycocoevalcap.meteor.Meteor`, a pipe to `meteor-1.5.jar` (dvc_eval/eval_dvc.py:20,67,
dvc_eval/SODA/soda.py:16-72, dvc_eval/pycocoevalcap/meteor/meteor.py:13-82); neither the jar nor a JVM exists in the reference tree or in this
image.  This module restates the PUBLISHED algorithm (Denkowski & Lavie, "Meteor Universal", WMT 2014; METEOR 1.5 `-l en -norm` defaults) for
the two matching stages that need no external resource:

  stage 1  exact   (module weight 1.0): identical lower-cased tokens
  stage 2  stem    (module weight 0.6): identical Porter stems (Porter 1980, restated below; the jar uses Snowball English, which differs
                    on a few suffix classes)
  NOT here: stage 3 synonyms (WordNet, weight 0.8), stage 4 paraphrase tables (weight 0.6), and the function-word list that the
            delta parameter discounts -- every word counts as a content word.

Scoring (MeteorScorer of the jar, parameters alpha = 0.85, beta = 0.2, gamma = 0.6): P / R = weighted matches over hypothesis / reference
length, Fmean = P R / (alpha P + (1 - alpha) R), fragmentation = chunks / matched words (0 when the two sentences align completely in one
chunk), score = Fmean (1 - gamma frag^beta); several references: the best one; the corpus score is computed from the SUMMED statistics of
the segments' best references, the per-segment scores from their own.

Because stages 3-4 only ADD matches, METEOR-lite <= METEOR on every pair.  Parity with the jar is UNPINNED (no jar to run): the tests pin the
implementation to hand-computed alignments.  Everything that reports it says "lite": `MeteorLite.method()`, the result keys
`METEOR-lite` / `soda_c_meteor_lite`.  Same call convention as pycocoevalcap scorers: compute_score(gts, res) -> (score, [scores])."""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Sequence, Tuple

ALPHA, BETA, GAMMA = 0.85, 0.2, 0.6
W_EXACT, W_STEM = 1.0, 0.6
BEAM = 40

# ------------------------------------------------------------------------------------------------------------ Porter (1980) stemmer
_V = "aeiou"


def _cons(w: str, i: int) -> bool:
    c = w[i]
    if c in _V:
        return False
    if c == "y":
        return i == 0 or not _cons(w, i - 1)
    return True


def _m(stem: str) -> int:
    """number of VC sequences"""
    n, prev_v = 0, False
    for i in range(len(stem)):
        v = not _cons(stem, i)
        if prev_v and not v:
            n += 1
        prev_v = v
    return n


def _has_vowel(stem: str) -> bool:
    return any(not _cons(stem, i) for i in range(len(stem)))


def _double_c(w: str) -> bool:
    return len(w) >= 2 and w[-1] == w[-2] and _cons(w, len(w) - 1)


def _cvc(w: str) -> bool:
    return len(w) >= 3 and _cons(w, len(w) - 3) and not _cons(w, len(w) - 2) and _cons(w, len(w) - 1) and w[-1] not in "wxy"


_STEP2 = (("ational", "ate"), ("tional", "tion"), ("enci", "ence"), ("anci", "ance"), ("izer", "ize"), ("abli", "able"), ("alli", "al"),
          ("entli", "ent"), ("eli", "e"), ("ousli", "ous"), ("ization", "ize"), ("ation", "ate"), ("ator", "ate"), ("alism", "al"),
          ("iveness", "ive"), ("fulness", "ful"), ("ousness", "ous"), ("aliti", "al"), ("iviti", "ive"), ("biliti", "ble"))
_STEP3 = (("icate", "ic"), ("ative", ""), ("alize", "al"), ("iciti", "ic"), ("ical", "ic"), ("ful", ""), ("ness", ""))
_STEP4 = ("al", "ance", "ence", "er", "ic", "able", "ible", "ant", "ement", "ment", "ent", "ion", "ou", "ism", "ate", "iti", "ous", "ive", "ize")


def porter_stem(w: str) -> str:
    if len(w) <= 2 or not w.isalpha():
        return w
    # 1a
    if w.endswith("sses"):
        w = w[:-2]
    elif w.endswith("ies"):
        w = w[:-2]
    elif w.endswith("ss"):
        pass
    elif w.endswith("s"):
        w = w[:-1]
    # 1b
    second = False
    if w.endswith("eed"):
        if _m(w[:-3]) > 0:
            w = w[:-1]
    elif w.endswith("ed") and _has_vowel(w[:-2]):
        w, second = w[:-2], True
    elif w.endswith("ing") and _has_vowel(w[:-3]):
        w, second = w[:-3], True
    if second:
        if w.endswith(("at", "bl", "iz")):
            w += "e"
        elif _double_c(w) and w[-1] not in "lsz":
            w = w[:-1]
        elif _m(w) == 1 and _cvc(w):
            w += "e"
    # 1c
    if w.endswith("y") and _has_vowel(w[:-1]):
        w = w[:-1] + "i"
    # 2, 3
    for table in (_STEP2, _STEP3):
        for suf, rep in table:
            if w.endswith(suf):
                if _m(w[:-len(suf)]) > 0:
                    w = w[:-len(suf)] + rep
                break
    # 4
    for suf in sorted(_STEP4, key=len, reverse=True):
        if w.endswith(suf):
            stem = w[:-len(suf)]
            if _m(stem) > 1 and (suf != "ion" or stem.endswith(("s", "t"))):
                w = stem
            break
    # 5
    if w.endswith("e"):
        stem = w[:-1]
        if _m(stem) > 1 or (_m(stem) == 1 and not _cvc(stem)):
            w = stem
    if _m(w) > 1 and _double_c(w) and w.endswith("l"):
        w = w[:-1]
    return w


# ------------------------------------------------------------------------------------------------------------ alignment
def _candidates(hyp: Sequence[str], ref: Sequence[str]) -> List[List[Tuple[int, float]]]:
    """per hypothesis position: the reference positions it may align to, with the weight of the FIRST stage that matches the pair"""
    hs, rs = [porter_stem(w) for w in hyp], [porter_stem(w) for w in ref]
    out = []
    for i, w in enumerate(hyp):
        c = []
        for j, v in enumerate(ref):
            if w == v:
                c.append((j, W_EXACT))
            elif hs[i] == rs[j]:
                c.append((j, W_STEM))
        out.append(c)
    return out


def align(hyp: Sequence[str], ref: Sequence[str]) -> Tuple[int, float, int]:
    """(matched words, their summed module weights, chunks) of the best one-to-one alignment: most matched words, then most weight (an
    exact match before a stem match), then fewest chunks -- the aligner's criteria, by a beam over the hypothesis positions (width 40
    like the jar's).  A chunk is a run of matches contiguous and in the same order in both sentences."""
    cands = _candidates(hyp, ref)
    # states: (used reference positions (bitmask), reference position of the previous hypothesis word's match or -2) -> (matches, weight, -chunks)
    beam = {(0, -2): (0, 0.0, 0)}
    for c in cands:
        nxt: Dict[Tuple[int, int], Tuple[int, float, int]] = {}

        def put(key, val):
            if key not in nxt or val > nxt[key]:
                nxt[key] = val
        for (used, last), (n, wsum, negch) in beam.items():
            put((used, -2), (n, wsum, negch))                              # leave this word unmatched
            for j, wt in c:
                if not (used >> j) & 1:
                    put((used | (1 << j), j), (n + 1, wsum + wt, negch - (0 if j == last + 1 and last >= 0 else 1)))
        if len(nxt) > BEAM:
            nxt = dict(sorted(nxt.items(), key=lambda kv: kv[1], reverse=True)[:BEAM])
        beam = nxt
    n, wsum, negch = max(beam.values())
    return n, wsum, -negch


def _stats(hyp: Sequence[str], ref: Sequence[str]) -> Tuple[float, ...]:
    n, wsum, ch = align(hyp, ref)
    return (float(len(hyp)), float(len(ref)), float(n), wsum, float(ch))


def _score(st: Sequence[float]) -> float:
    lh, lr, n, wsum, ch = st
    if n == 0 or lh == 0 or lr == 0:
        return 0.0
    p, r = wsum / lh, wsum / lr
    fmean = p * r / (ALPHA * p + (1.0 - ALPHA) * r)
    frag = 0.0 if (n == lh and n == lr and ch == 1) else ch / n
    return fmean * (1.0 - GAMMA * frag ** BETA)


_TOK = re.compile(r"\s+")


class MeteorLite:
    """pycocoevalcap-style scorer: gts = {id: [reference, ...]}, res = {id: [hypothesis]} (pre-tokenised strings)."""

    def __init__(self):
        self._cache: Dict[Tuple[str, str], Tuple[float, ...]] = {}

    def method(self) -> str:
        return "METEOR-lite"

    def pair_stats(self, hyp: str, ref: str) -> Tuple[float, ...]:
        key = (hyp, ref)
        if key not in self._cache:
            h, r = [t for t in _TOK.split(hyp.lower()) if t], [t for t in _TOK.split(ref.lower()) if t]
            self._cache[key] = _stats(h, r)
        return self._cache[key]

    def best(self, hyp: str, refs: Sequence[str]) -> Tuple[float, Tuple[float, ...]]:
        cand = [self.pair_stats(hyp, r) for r in refs] or [(0.0,) * 5]
        st = max(cand, key=_score)
        return _score(st), st

    def compute_score(self, gts: Dict, res: Dict) -> Tuple[float, List[float]]:
        assert gts.keys() == res.keys()
        scores, agg = [], [0.0] * 5
        for k in gts.keys():                                            # dict order, like pycocoevalcap (soda.py:66-72 relies on it)
            assert len(res[k]) == 1
            s, st = self.best(res[k][0], gts[k])
            scores.append(s)
            agg = [a + b for a, b in zip(agg, st)]
        return _score(agg), scores


_SHARED: Optional["MeteorLite"] = None


def shared_scorer() -> "MeteorLite":
    """One module-level scorer: its (hypothesis, reference) pair cache is kept across eval_dvc / eval_soda calls instead of being rebuilt by each."""
    global _SHARED
    if _SHARED is None:
        _SHARED = MeteorLite()
    return _SHARED
