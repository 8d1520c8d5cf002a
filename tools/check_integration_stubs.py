"""Runs the ctypes stubs printed in INTEGRATION.md section 2 as they stand (the python blocks are exec'd from the file) and checks the
beam-step stub against vidchapters_amd.lib.beam_advance on the same random candidates.  usage: python tools/check_integration_stubs.py"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vidchapters_amd import lib as L

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
text = open(os.path.join(root, "INTEGRATION.md")).read()
sec = text[text.index("## 2. The C-ABI"):text.index("## 2b.")]
ns = {}
os.chdir(root)
for block in re.findall(r"```python\n(.*?)```", sec, flags=re.S):
    exec(block, ns)

B, nb, K, maxlen, eos, pad, lp = 3, 4, 8, 12, 1, 0, 0.8
R = B * nb
rng = np.random.default_rng(0)
st_a, st_b = ns["beam_state"](B, nb, maxlen + 1, lp), L.BeamState(B, nb, maxlen + 1, "cuda", lp)
mk = lambda: dict(hist=torch.zeros(R, maxlen + 1, dtype=torch.long, device="cuda"), row_map=torch.zeros(R, maxlen, dtype=torch.int32, device="cuda"),
                  nxt=torch.zeros(R, dtype=torch.long, device="cuda"), sc=torch.zeros(R, device="cuda"), src=torch.zeros(R, dtype=torch.int32, device="cuda"))
a, b = mk(), mk()
pos = torch.zeros(1, dtype=torch.int32, device="cuda")
for t in range(maxlen):
    val = torch.from_numpy((-np.sort(rng.random((R, K)).astype(np.float32), axis=1) * 3 - t)).cuda().contiguous()
    tok = torch.from_numpy(np.stack([rng.permutation(np.arange(1 if t > 1 else 2, 30))[:K] for _ in range(R)]).astype(np.int32)).cuda()
    for x in (a, b):
        x["row_map"][:, t] = torch.arange(R, dtype=torch.int32, device="cuda")
    ns["beam_step"](val, tok, K, st_a, B, nb, eos, pad, pos, a["hist"], a["row_map"], a["nxt"], a["sc"], a["src"])
    L.beam_advance(val, tok, K, st_b, eos, pad, pos, b["hist"], b["row_map"], b["nxt"], b["sc"], b["src"])
    L.counter_add(pos, 1)
    for k in a:
        assert torch.equal(a[k], b[k]), (t, k)
for k in ("hyp_tok", "hyp_len", "hyp_score", "hyp_order", "heap_n", "heap_worst", "done", "ndone"):
    assert torch.equal(st_a[k], getattr(st_b, k)), k
print("INTEGRATION.md stubs: exec ok; beam-step stub == lib.beam_advance over", maxlen, "steps; finished hypotheses:", int(st_b.heap_n.sum()))
