import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import SyntheticTokenizer, Vid2Seq, synth
dev = torch.device("cuda")
tok = SyntheticTokenizer(32100, 100)
model = Vid2Seq("t5-base", tokenizer=tok, init_seed=1234, device=dev).eval()
b = synth.make_batch(64, 100, 1000, 8, len(tok), 4321, 768)
ids = b["input_ids"].to(dev)
eng = model.engine()
toks = eng.greedy(b["video"].to(dev).to(torch.bfloat16), {"input_ids": ids, "attention_mask": ids != 0}, max_new_tokens=64, stop_at_eos=False,
                  use_graph=False)
torch.cuda.synchronize()
print(toks.shape)
