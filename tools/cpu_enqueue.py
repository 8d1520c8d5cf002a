"""How long does the host take to ENQUEUE one training step (no device sync inside)?  If this approaches the device time of a step,
the step is launch-bound and kernel work stops mattering."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import SyntheticTokenizer, Vid2Seq, synth
from vidchapters_amd.train import Trainer
dev = torch.device("cuda")
tok = SyntheticTokenizer(32100, 100)
model = Vid2Seq("t5-base", tokenizer=tok, vis_drop=0.1, enc_drop=0.1, dec_drop=0.1, init_seed=1234, device=dev).train()
tr = Trainer(model, denoising=0.0)
batch = {k: v.to(dev) for k, v in synth.make_batch(32, 100, 1000, 256, len(tok), 1234, 768).items()}
batch["video"] = batch["video"].to(torch.bfloat16)
batch["input_lens"] = (batch["input_ids"] != 0).sum(1).tolist()
for pack in (False, True):
    model.engine().pack = pack
    for _ in range(3): tr.step(batch)
    torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        tr.step(batch)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
    print(f"pack={pack}: host enqueue {min(enq):.1f}-{max(enq):.1f} ms per step, step wall {min(tot):.1f}-{max(tot):.1f} ms")
# the same step replayed from a hipGraph (Trainer.step_graph)
b2 = {k: v for k, v in batch.items() if torch.is_tensor(v)}
for _ in range(3): tr.step_graph(b2)
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(8):
    t0 = time.perf_counter()
    tr.step_graph(b2)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print(f"graph replay: host enqueue {min(enq):.2f}-{max(enq):.2f} ms per step, step wall {min(tot):.1f}-{max(tot):.1f} ms")
