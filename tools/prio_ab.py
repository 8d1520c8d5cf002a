"""Stream priorities in the train step: the main chain on a high-priority stream, the weight-gradient stream at normal priority
(does the hardware let the main chain's kernels go first where both are ready?).  usage: python tools/prio_ab.py"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import SyntheticTokenizer, Vid2Seq, synth
from vidchapters_amd.train import Trainer
dev = torch.device("cuda", 0)
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
tok = SyntheticTokenizer(32100, 100)
model = Vid2Seq("t5-base", num_features=100, tokenizer=tok, vis_drop=0.1, enc_drop=0.1, dec_drop=0.1, init_seed=1234, device=dev).train()
eng = model.engine(); eng.pack = False
tr = Trainer(model, lr=3e-4, clip_max_norm=1.0, generative=1.0, denoising=0.0)
batch = {k: v.to(dev) for k, v in synth.make_batch(32, 100, 1000, 256, len(tok), 1234, 768).items()}
batch["video"] = batch["video"].to(torch.bfloat16)
batch["input_lens"] = (batch["input_ids"] != 0).sum(1).tolist(); batch["output_lens"] = (batch["output_ids"] != 0).sum(1).tolist()
hi = torch.cuda.Stream(device=dev, priority=-1)
side0 = (eng.wstream, eng.vstream, eng.kstream)
sideh = tuple(torch.cuda.Stream(device=dev, priority=-1) for _ in range(3))
def setup(name):
    if name == "all normal": eng.wstream, eng.vstream, eng.kstream = side0; return None
    if name == "main high": eng.wstream, eng.vstream, eng.kstream = side0; return hi
    if name == "main + vit + kv high, wgrad normal": eng.wstream = side0[0]; eng.vstream, eng.kstream = sideh[1], sideh[2]; return hi
    if name == "wgrad high": eng.wstream = sideh[0]; eng.vstream, eng.kstream = side0[1], side0[2]; return None
names = ["all normal", "main high", "main + vit + kv high, wgrad normal", "wgrad high"]
def step(st):
    if st is None:
        tr.step(batch)
    else:
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            tr.step(batch)
        torch.cuda.current_stream().wait_stream(st)
for n in names:
    st = setup(n); step(st); step(st)
torch.cuda.synchronize()
times = {n: [] for n in names}
for i in range(8):
    for n in names:
        st = setup(n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); step(st); step(st); e1.record(); torch.cuda.synchronize()
        times[n].append(e0.elapsed_time(e1) / 2)
for n in names:
    print(f"{n:40s} median {statistics.median(times[n]):7.2f} ms  min {min(times[n]):7.2f}")
