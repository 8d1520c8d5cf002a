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
