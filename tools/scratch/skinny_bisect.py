import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vidchapters_amd import SyntheticTokenizer, Vid2Seq, synth
from vidchapters_amd import lib as L
DEV = torch.device("cuda")
model = Vid2Seq("t5-base", tokenizer=SyntheticTokenizer(32100, 100), vis_drop=0.0, enc_drop=0.0, dec_drop=0.0, init_seed=11, device=DEV).train()
eng = model.engine()
b = synth.make_batch(2, 100, 1000, 256, 32200, 4242, 768)
video, ids, out = b["video"].to(DEV).to(torch.bfloat16), b["input_ids"].to(DEV), b["output_ids"].to(DEV)
names = list(model.state_dict().keys())
def run(sl):
    eng.prepare(); eng.arena.grad.zero_()
    vt, tp = {}, {}
    vis = eng.vit_forward(video[sl], vt).view(-1, 100, eng.d)
    loss = eng.t5_loss_forward(vis, ids[sl], ids[sl] != 0, out[sl], out[sl] != 0, tp)
    dvis = eng.t5_loss_backward(tp, torch.ones(1, device=DEV))
    eng.vit_backward(vt, dvis); eng.join_wgrads(); torch.cuda.synchronize()
    return float(loss.item()), eng.arena.grad.clone(), dvis.clone()
res = {}
for mode in (0, 1):
    L.set_option("gemm_skinny", mode)
    res[mode] = run(slice(0, 1))
L.set_option("gemm_skinny", 1)
print("loss", res[0][0], res[1][0])
g0, g1 = res[0][1], res[1][1]
print("dvis rel diff", float((res[0][2].float() - res[1][2].float()).norm() / res[0][2].float().norm()))
a = eng.arena
worst = []
for k in a.names if hasattr(a, "names") else []:
    x0, x1 = a.g(k), None
seen = set()
for k in names:
    try:
        off, n = a.offsets[k], 1
    except Exception:
        continue
for k, off in sorted(a.offsets.items(), key=lambda kv: kv[1]):
    n = 1
    for s_ in a.shapes[k]: n *= s_
    x0, x1 = g0[off:off + n].double(), g1[off:off + n].double()
    c = float((x0 * x1).sum() / (x0.norm() * x1.norm() + 1e-300))
    worst.append((c, k))
worst.sort()
for c, k in worst[:25]: print(f"{c:.6f} {k}")
