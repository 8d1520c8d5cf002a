import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import SyntheticTokenizer, Vid2Seq, synth
dev = torch.device("cuda")
tok = SyntheticTokenizer(32100, 100)
model = Vid2Seq("t5-base", tokenizer=tok, init_seed=1234, device=dev).eval()
b = synth.make_batch(int(os.environ.get("BEAM_B", "64")), 100, 1000, 8, len(tok), 4321, 768)
ids = b["input_ids"].to(dev)
eng = model.engine()
toks = eng.beam_search(b["video"].to(dev).to(torch.bfloat16), {"input_ids": ids, "attention_mask": ids != 0}, num_beams=4, max_new_tokens=24,
                       use_graph=False)
torch.cuda.synchronize()
print(toks.shape)
