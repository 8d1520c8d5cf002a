"""How far does the bf16-mode oracle move against ITSELF when only the fp32 summation order (BLAS thread count) or a 1e-6 input jitter changes?\nThe self-noise floor that bounds what tests/test_configs_gpu.py::test_cfg2_shape_vs_bf16_mode_oracle can demand (profiles/r05_bf16mode_parity.txt).\nusage: python tests/tools/bf16_mode_self_noise.py   (build container or GPU host; ~1 minute on 8 cores)"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import vid2seq_ref as R
from oracle.make_golden import oracle_params, grad_sample, wants_slice
from vidchapters_amd import synth
cfg = R.RefConfig(); seed = 2024
batch = synth.make_batch(2, 100, 1000, 256, cfg.vocab, seed, cfg.vit_dim)
def run(nthreads, jitter):
    torch.set_num_threads(nthreads)
    P = oracle_params(cfg, seed, grad=True)
    if jitter:      # perturb the VIDEO input by one part in 1e6 (far below bf16 resolution except at rounding boundaries)
        g = torch.Generator().manual_seed(1)
        v = batch["video"] * (1 + 1e-6 * torch.randn(batch["video"].shape, generator=g))
    else:
        v = batch["video"]
    with R.bf16_mode():
        out, _ = R.vid2seq_forward(P, cfg, v, batch["input_ids"], batch["input_ids"] != 0, batch["output_ids"], batch["output_ids"] != 0)
        out["loss"].backward()
    return float(out["loss"]), {k: grad_sample(k, P[k].grad) for k in P if wants_slice(k, cfg.n_enc)}
a = run(8, False); b = run(3, False); c = run(8, True)
def cmp(x, y, tag):
    cs = sorted((float(x[1][k].double().flatten() @ y[1][k].double().flatten() / (x[1][k].double().norm() * y[1][k].double().norm() + 1e-30)), k) for k in x[1])
    print(tag, "loss", x[0], y[0], "worst", cs[:3], "mean", sum(c for c, _ in cs) / len(cs))
cmp(a, b, "threads 8 vs 3 (summation order):")
cmp(a, c, "video jitter 1e-6:")
