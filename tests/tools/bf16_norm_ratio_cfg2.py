"""Is the one-sided gradient-NORM deficit of the bf16 HIP path (total norm -0.5 % at the cfg-2 shape, every tensor 0.996-0.999 of the
fp32 reference) a property of bf16 compute or of a kernel?  The fp32 CPU oracle under torch's CPU bf16 autocast (bf16 matmul operands,
fp32 accumulation -- the precision class of the engine) on the inputs of tests/golden/full_cfg2_scalars.npz: total gradient norm and the
norm of every tensor the golden records, as ratios to the fp32 reference's.  Companion of bf16_noise_cfg2.py (directions).
usage: python tests/tools/bf16_norm_ratio_cfg2.py   (several minutes on 8 cores)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from oracle import vid2seq_ref as R
from oracle.make_golden import grad_sample
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
names = list(P)
grads = dict(zip(names, torch.autograd.grad(out["loss"], [P[k] for k in names])))
tot = float(torch.sqrt(sum((v.double() ** 2).sum() for v in grads.values())))
print(f"total grad norm: bf16 autocast {tot:.6f}  fp32 reference {float(g['grad_norm']):.6f}  ratio {tot / float(g['grad_norm']):.5f}")
keys = [str(k) for k in g["grad_norm_keys"]]
vals = g["grad_norm_vals"]
ratios = []
for k, v in zip(keys, vals):
    if k in grads and v > 0:
        ratios.append((float(grads[k].double().norm()) / float(v), k))
ratios.sort()
rs = np.array([r for r, _ in ratios])
print(f"{len(rs)} tensors: norm ratio min {rs.min():.4f} median {np.median(rs):.4f} max {rs.max():.4f}; below 1: {(rs < 1).mean() * 100:.0f} %")
for r, k in ratios[:8] + ratios[-4:]:
    print(f"  {r:.4f}  {k}")
# projection of the bf16 gradient on the fp32 one (sampled entries): a = <got, want> / <want, want>
proj = []
for key in g.files:
    if key.startswith("gs:"):
        name = key[3:]
        want = torch.from_numpy(g[key]).double().flatten()
        got = grad_sample(name, grads[name].float()).double().flatten()
        proj.append(float(want @ got / (want @ want + 1e-300)))
proj = np.array(proj)
print(f"projection coefficient on the fp32 gradient (sampled): min {proj.min():.4f} median {np.median(proj):.4f} max {proj.max():.4f}")
