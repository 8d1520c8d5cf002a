import sys; sys.path.insert(0,'/root/repo')
import torch
from oracle import vid2seq_ref as R
from vidchapters_amd import synth
cfg = R.RefConfig.small()
def run(autocast):
    P = synth.init_params(R.param_shapes(cfg), 7, cfg.d_model, cfg.inner, cfg.d_ff)
    for v in P.values(): v.requires_grad_(True)
    b = synth.make_batch(3, 10, 24, 12, cfg.vocab, 7, cfg.vit_dim)
    b["input_ids"][0, 1:] = 0; b["input_ids"][0, 0] = 1
    b["output_ids"][1, 1:] = 0; b["output_ids"][1, 0] = 1
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        out,_ = R.vid2seq_forward(P, cfg, b["video"], b["input_ids"], b["input_ids"]!=0, b["output_ids"], b["output_ids"]!=0)
    g = torch.autograd.grad(out["loss"], list(P.values()))
    return out["loss"].item(), dict(zip(P.keys(), g))
l0,g0 = run(False); l1,g1 = run(True)
print("loss", l0, l1)
cs = {}
for k in g0:
    a,b = g0[k].double().flatten(), g1[k].double().flatten()
    cs[k] = float(a@b/(a.norm()*b.norm()))
for k,v in sorted(cs.items(), key=lambda kv: kv[1])[:12]: print(f"{v:.4f} {k}")
