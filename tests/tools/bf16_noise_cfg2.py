"""How much gradient-direction noise does bf16 compute add at the cfg-2 shape?  The fp32 CPU oracle vs the SAME oracle under torch's CPU
bf16 autocast (matmuls in bf16, fp32 accumulation -- the precision class of the HIP engine) on the inputs of
tests/golden/full_cfg2_scalars.npz: cosine of every sampled gradient tensor.  Justifies the thresholds of
tests/test_configs_gpu.py::test_cfg2_shape_vs_reference_golden (the engine must not be worse than the reference math run in bf16).
usage: python tests/tools/bf16_noise_cfg2.py   (several minutes on 8 cores)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from oracle import vid2seq_ref as R
from oracle.make_golden import grad_sample, wants_slice
from vidchapters_amd import synth

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "full_cfg2_scalars.npz"))
seed, B, T, L, Lo = (int(g[k]) for k in ("seed", "B", "T", "L", "Lo"))
cfg = R.RefConfig()
torch.set_num_threads(os.cpu_count())
P = synth.init_params(R.param_shapes(cfg), seed, cfg.d_model, cfg.inner, cfg.d_ff)
for v in P.values():
    v.requires_grad_(True)
b = synth.make_batch(B, T, L, Lo, cfg.vocab, seed, 768)
with torch.autocast("cpu", dtype=torch.bfloat16):
    out, _ = R.vid2seq_forward(P, cfg, b["video"], b["input_ids"], b["input_ids"] != 0, b["output_ids"], b["output_ids"] != 0)
print(f"loss bf16-autocast {out['loss'].item():.6f}  fp32 reference {float(g['loss']):.6f}")
names = list(P)
grads = dict(zip(names, torch.autograd.grad(out["loss"], [P[k] for k in names])))
rows = []
for key in g.files:
    if key.startswith("gs:"):
        name = key[3:]
        want = torch.from_numpy(g[key]).double().flatten()
        got = grad_sample(name, grads[name].float()).double().flatten()
        rows.append((float(want @ got / (want.norm() * got.norm() + 1e-30)), name))
rows.sort()
print("lowest cosines (bf16 autocast of the reference math vs its fp32 gradients):")
for c, n in rows[:20]:
    print(f"  {c:.4f}  {n}")
one_d = [c for c, n in rows if g["gs:" + n].ndim <= 1 or n.endswith("relative_attention_bias.weight")]
two_d = [c for c, n in rows if not (g["gs:" + n].ndim <= 1 or n.endswith("relative_attention_bias.weight"))]
print(f"min 2-D {min(two_d):.4f}  min 1-D {min(one_d):.4f}  median 2-D {sorted(two_d)[len(two_d) // 2]:.4f}")
