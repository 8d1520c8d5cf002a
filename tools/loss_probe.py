"""Six optimizer steps of the bench workload (cfg-2, B = 32, dropout 0.1, dense) per value of the attn_order option: (loss, gradient norm) per step.
The probe that showed the dK/dV early-exit race of round 6 (gradient norm 4e12 in step 0, loss -inf in step 2 while every parity test passed);
V2S_LIB=<other build> compares builds.  usage: python tools/loss_probe.py 0 1 2"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from vidchapters_amd import SyntheticTokenizer, Vid2Seq, synth
from vidchapters_amd import lib as L
from vidchapters_amd.train import Trainer
dev = torch.device("cuda", 0)
for order in [int(x) for x in sys.argv[1:]]:
    L.set_option("attn_order", order)
    tok = SyntheticTokenizer(32100, 100)
    model = Vid2Seq("t5-base", num_features=100, tokenizer=tok, vis_drop=0.1, enc_drop=0.1, dec_drop=0.1, init_seed=1234, device=dev).train()
    model.engine().pack = False
    tr = Trainer(model, lr=3e-4, clip_max_norm=1.0, generative=1.0, denoising=0.0)
    batch = {k: v.to(dev) for k, v in synth.make_batch(32, 100, 1000, 256, len(tok), 1234, 768).items()}
    batch["video"] = batch["video"].to(torch.bfloat16)
    out = []
    for i in range(6):
        l = tr.step(batch)
        out.append((round(float(l["loss"]), 4), round(float(tr.grad_norm()), 4)))
    print("attn_order", order, out, flush=True)
    del model, tr
