"""METEOR-lite (vidchapters_amd/meteor_lite.py): the exact + stem stages of METEOR 1.5 restated without the jar (SURVEY.md §8f N4;
reference: dvc_eval/eval_dvc.py:20,67, dvc_eval/SODA/soda.py:16-72, dvc_eval/pycocoevalcap/meteor/meteor.py).  There is no jar to run, so
parity with the reference's scorer is UNPINNED; these tests pin the restatement to the published stemmer vectors and to hand-computed
alignments / scores, and check how evalmetrics reports it (always under a "lite" key)."""
import math

import numpy as np
import pytest

from vidchapters_amd import evalmetrics as M
from vidchapters_amd import meteor_lite as ML

# examples of M. F. Porter, "An algorithm for suffix stripping" (1980), one or more per rule
PORTER = {"caresses": "caress", "ponies": "poni", "ties": "ti", "caress": "caress", "cats": "cat", "feed": "feed", "agreed": "agre",
          "plastered": "plaster", "bled": "bled", "motoring": "motor", "sing": "sing", "conflated": "conflat", "troubled": "troubl",
          "sized": "size", "hopping": "hop", "tanned": "tan", "falling": "fall", "hissing": "hiss", "fizzed": "fizz", "failing": "fail",
          "filing": "file", "happy": "happi", "sky": "sky", "relational": "relat", "conditional": "condit", "rational": "ration",
          "valenci": "valenc", "digitizer": "digit", "conformabli": "conform", "radicalli": "radic", "differentli": "differ", "vileli": "vile",
          "analogousli": "analog", "vietnamization": "vietnam", "predication": "predic", "operator": "oper", "feudalism": "feudal",
          "decisiveness": "decis", "hopefulness": "hope", "callousness": "callous", "formaliti": "formal", "sensitiviti": "sensit",
          "sensibiliti": "sensibl", "triplicate": "triplic", "formative": "form", "formalize": "formal", "electriciti": "electr",
          "electrical": "electr", "hopeful": "hope", "goodness": "good", "revival": "reviv", "allowance": "allow", "inference": "infer",
          "airliner": "airlin", "gyroscopic": "gyroscop", "adjustable": "adjust", "defensible": "defens", "irritant": "irrit",
          "replacement": "replac", "adjustment": "adjust", "dependent": "depend", "adoption": "adopt", "homologou": "homolog",
          "communism": "commun", "activate": "activ", "angulariti": "angular", "homologous": "homolog", "effective": "effect",
          "bowdlerize": "bowdler", "probate": "probat", "rate": "rate", "cease": "ceas", "controll": "control", "roll": "roll"}


def test_porter_stemmer_published_vectors():
    bad = {w: (ML.porter_stem(w), s) for w, s in PORTER.items() if ML.porter_stem(w) != s}
    assert not bad, bad
    assert ML.porter_stem("cooking") == ML.porter_stem("cooks") == ML.porter_stem("cooked") == "cook"
    assert ML.porter_stem("a") == "a" and ML.porter_stem("3-4") == "3-4"


def _by_hand(lh, lr, wsum, n, ch):
    p, r = wsum / lh, wsum / lr
    f = p * r / (0.85 * p + 0.15 * r)
    frag = 0.0 if (n == lh == lr and ch == 1) else ch / n
    return f * (1.0 - 0.6 * frag ** 0.2)


def test_alignment_and_score_hand_computed():
    # exact a, man | stem cooking ~ cooks, exact food: 4 words, weight 3.6, two chunks ("is" is unmatched and splits them)
    assert ML.align("a man is cooking food".split(), "a man cooks food".split()) == (4, pytest.approx(3.6), 2)
    # a permutation: every word matched exactly, three chunks (the cat | sat | on the mat)
    assert ML.align("the cat sat on the mat".split(), "on the mat sat the cat".split()) == (6, pytest.approx(6.0), 3)
    # a repeated word must take the occurrence that keeps the chunk whole: 3 matches, ONE chunk
    assert ML.align("the dog the".split(), "a the dog the".split()) == (3, pytest.approx(3.0), 1)
    # more occurrences in the hypothesis than in the reference: only as many matches as the reference has
    assert ML.align("go go go".split(), "go".split())[0] == 1
    m = ML.MeteorLite()
    s, per = m.compute_score({0: ["a man cooks food"]}, {0: ["a man is cooking food"]})
    assert per[0] == pytest.approx(_by_hand(5, 4, 3.6, 4, 2)) == pytest.approx(0.414364, abs=1e-5) and s == pytest.approx(per[0])
    # identical sentences: fragmentation 0 by the scorer's special case -> 1.0; nothing in common -> 0.0
    assert m.compute_score({0: ["x y z"]}, {0: ["x y z"]}) == (pytest.approx(1.0), [pytest.approx(1.0)])
    assert m.compute_score({0: ["x y z"]}, {0: ["a b"]}) == (0.0, [0.0])
    # several references: the best one; the corpus score comes from the SUMMED statistics, not from the mean of the segments
    s, per = m.compute_score({0: ["a b", "x y z w"], 1: ["a man cooks food"]}, {0: ["x y z"], 1: ["a man is cooking food"]})
    assert per[0] == pytest.approx(_by_hand(3, 4, 3.0, 3, 1)) and per[1] == pytest.approx(_by_hand(5, 4, 3.6, 4, 2))
    assert s == pytest.approx(_by_hand(3 + 5, 4 + 4, 3.0 + 3.6, 3 + 4, 1 + 2)) and abs(s - np.mean(per)) > 1e-3
    assert m.method() == "METEOR-lite"


def test_reported_only_under_lite_keys():
    rng = np.random.RandomState(1)
    vocab = ["a", "man", "woman", "is", "cooking", "cooks", "food", "in", "the", "kitchen", "talks", "talking", "to", "camera", "dog", "runs", "running"]
    sent = lambda: " ".join(vocab[i] for i in rng.randint(0, len(vocab), rng.randint(3, 9)))
    ref = {f"v{v}": {"timestamps": [[10.0 * i, 10.0 * i + 8] for i in range(4)], "sentences": [sent() for _ in range(4)]} for v in range(5)}
    perfect = {"results": {v: [{"sentence": s, "timestamp": list(t)} for t, s in zip(r["timestamps"], r["sentences"])] for v, r in ref.items()}}
    noisy = {"results": {v: [{"sentence": sent(), "timestamp": [t[0] + 1.0, t[1] + 1.0]} for t in r["timestamps"]] for v, r in ref.items()}}
    tok = lambda s: " ".join(s.lower().split())
    out_p, out_n = M.eval_dvc(perfect, [ref], tokenize=tok, meteor_lite=True), M.eval_dvc(noisy, [ref], tokenize=tok, meteor_lite=True)
    # a perfect submission: per (tIoU, video) group four identical pairs, one chunk each -- the jar's aggregate fragmentation is chunks / matches
    # of the SUMS, so a group of several segments scores below 1 even when every segment is perfect
    want = np.mean([1.0 - 0.6 * (4.0 / sum(len(s.split()) for s in r["sentences"])) ** 0.2 for r in ref.values()])
    assert "METEOR" not in out_p and out_p["METEOR-lite"] == pytest.approx(want) and 0.0 <= out_n["METEOR-lite"] < 0.8 * want
    assert "METEOR-lite" not in M.eval_dvc(perfect, [ref], tokenize=tok)          # opt-in
    sp, sn = M.eval_soda(perfect, [ref], tokenize=tok, scorer="meteor_lite"), M.eval_soda(noisy, [ref], tokenize=tok, scorer="meteor_lite")
    assert list(sp) == ["soda_c_meteor_lite"] and sp["soda_c_meteor_lite"] == pytest.approx(1.0) and 0.0 <= sn["soda_c_meteor_lite"] < sp["soda_c_meteor_lite"]
    # the string is a shorthand for passing the scorer object, which goes through the reference's call convention (soda.py:66-72)
    assert M.soda_c(noisy, ref, tok, scorer=ML.MeteorLite())[2] == pytest.approx(sn["soda_c_meteor_lite"])
    with pytest.raises(ValueError):
        M.eval_soda(perfect, [ref], tokenize=tok, scorer="meteor")
    assert not math.isnan(out_n["METEOR-lite"])
