"""What a greedy generate() call spends before / around its replayed decode steps (B = 64, t5-base, 100 frames + 1000 ASR tokens): the encoder,
the first (eager) step, the graph capture + instantiation, the replay loop.  usage: python tools/greedy_prologue_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import SyntheticTokenizer, Vid2Seq, synth

dev = torch.device("cuda")
tok = SyntheticTokenizer(32100, 100)
model = Vid2Seq("t5-base", tokenizer=tok, init_seed=1234, device=dev).eval()
b = synth.make_batch(64, 100, 1000, 8, len(tok), 4321, 768)
ids = b["input_ids"].to(dev); vid = b["video"].to(dev).to(torch.bfloat16)
eng = model.engine()
inp = {"input_ids": ids, "attention_mask": ids != 0}


def timed(f, n=5):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


with torch.no_grad():
    eng.greedy(vid, inp, max_new_tokens=8, stop_at_eos=False)
    t_enc = timed(lambda: eng.encode(vid, inp))
    t1 = timed(lambda: eng.greedy(vid, inp, max_new_tokens=1, stop_at_eos=False))                   # encode + set-up + one eager step
    t2g = timed(lambda: eng.greedy(vid, inp, max_new_tokens=2, stop_at_eos=False))                  # + capture + one replay
    t2e = timed(lambda: eng.greedy(vid, inp, max_new_tokens=2, stop_at_eos=False, use_graph=False)) # + one more eager step
    t256 = timed(lambda: eng.greedy(vid, inp, max_new_tokens=256, stop_at_eos=False), 3)
    t128 = timed(lambda: eng.greedy(vid, inp, max_new_tokens=128, stop_at_eos=False), 3)
print(f"encode {t_enc:.2f} ms; greedy(1 token) {t1:.2f} ms (set-up + first eager step = {t1 - t_enc:.2f}); greedy(2) with graph {t2g:.2f}, eager {t2e:.2f} "
      f"(capture + instantiate ~ {t2g - t2e:.2f} ms); greedy(128) {t128:.2f}, greedy(256) {t256:.2f} ms -> {(t256 - t128) / 128:.4f} ms per replayed step, "
      f"everything else {t256 - 255 * (t256 - t128) / 128:.2f} ms")
