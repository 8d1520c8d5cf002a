"""Throughput of the evaluation post-processing (eval_dvc + eval_soda) on a synthetic test-set-sized job: numpy implementation
(vidchapters_amd/evalmetrics.py) next to the plain-Python oracle port of the reference (oracle/eval_ref.py).  CPU only.
usage: python tests/tools/eval_bench.py [n_videos] [oracle_videos]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from vidchapters_amd import evalmetrics as M
from oracle import eval_ref as E

n_videos = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n_oracle = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.RandomState(0)
vocab = [f"w{i}" for i in range(3000)]
sent = lambda: " ".join(vocab[i] for i in rng.zipf(1.3, rng.randint(2, 10)) % 3000)
ref, sub = {}, {"results": {}}
for v in range(n_videos):
    dur = float(rng.randint(120, 1800)); ng = int(rng.randint(3, 15))
    cuts = np.sort(rng.uniform(0, dur, ng + 1))
    ref[f"v{v}"] = {"timestamps": [[float(cuts[i]), float(cuts[i + 1])] for i in range(ng)], "sentences": [sent() for _ in range(ng)]}
    npred = int(rng.randint(2, 15)); pc = np.sort(rng.uniform(0, dur, npred + 1))
    sub["results"][f"v{v}"] = [{"sentence": sent(), "timestamp": [float(pc[i]), float(pc[i + 1])]} for i in range(npred)]
tok = lambda s: " ".join(s.split())
t0 = time.time(); a = M.eval_dvc(sub, [ref], tokenize=tok); b = M.eval_soda(sub, [ref], tokenize=tok); t1 = time.time()
print(f"numpy  : {n_videos} videos in {t1 - t0:6.2f} s = {n_videos / (t1 - t0):8.1f} videos/s   CIDEr {a['CIDEr']:.4f} F1 {a['F1']:.4f} soda_c_cider {b['soda_c_cider']:.4f}")
keys = list(ref)[:n_oracle]
sref = {k: ref[k] for k in keys}; ssub = {"results": {k: sub["results"][k] for k in keys}}
t0 = time.time(); c = E.eval_dvc(ssub, [sref], tok); d = E.eval_soda(ssub, [sref], tok); t1 = time.time()
a2 = M.eval_dvc(ssub, [sref], tokenize=tok); b2 = M.eval_soda(ssub, [sref], tokenize=tok)
assert abs(a2["CIDEr"] - c["CIDEr"]) < 1e-9 and abs(b2["soda_c_cider"] - d["soda_c"]) < 1e-9 and abs(a2["F1"] - c["F1"]) < 1e-12
print(f"oracle : {n_oracle} videos in {t1 - t0:6.2f} s = {n_oracle / (t1 - t0):8.1f} videos/s   (same results on that subset)")
