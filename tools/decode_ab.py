"""Interleaved A/B of library options on the cached greedy / beam decode loop (B = 64 videos, t5-base, 100 frames + 1000 ASR tokens).
Usage: python tools/decode_ab.py [opt=value,opt=value ...]   e.g.  python tools/decode_ab.py gemm_skinny=2 gemm_skinny=1 gemm_skinny=3
(engine attributes with the eng: prefix: eng:decode_mem_attn=0 eng:decode_mem_attn=2 (3 with BEAMS); env B, BEAMS, STEPS, ROUNDS)
Each variant runs `rounds` times, interleaved; prints the per-variant median ms per decode step (whole greedy() call / steps, i.e. it
includes the encoder prologue -- so the encoder-only time is measured too and subtracted)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vidchapters_amd import SyntheticTokenizer, Vid2Seq, synth
from vidchapters_amd import lib as L


def main():
    variants = sys.argv[1:] or ["gemm_skinny=1"]
    steps = int(os.environ.get("STEPS", "64"))
    rounds = int(os.environ.get("ROUNDS", "5"))
    beams = int(os.environ.get("BEAMS", "1"))
    dev = torch.device("cuda")
    tok = SyntheticTokenizer(32100, 100)
    model = Vid2Seq("t5-base", tokenizer=tok, init_seed=1234, device=dev).eval()
    b = synth.make_batch(int(os.environ.get("B", "64")), 100, 1000, 8, len(tok), 4321, 768)
    ids = b["input_ids"].to(dev)
    vid = b["video"].to(dev).to(torch.bfloat16)
    eng = model.engine()
    inp = {"input_ids": ids, "attention_mask": ids != 0}

    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if beams > 1:
            out = eng.beam_search(vid, inp, num_beams=beams, max_new_tokens=n, min_length=n + 1)
        else:
            out = eng.greedy(vid, inp, max_new_tokens=n, stop_at_eos=False)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3, out

    res = {v: [] for v in variants}
    toks = {}
    for r in range(rounds + 1):
        for v in variants:
            saved = []
            for kv in v.split(","):
                k, x = kv.split("=")
                if k == "lib":                        # another build of the library (same ABI): "lib=tools/libvid2seq_hip_nt.so"
                    L.LIB_PATH = os.path.abspath(x); L._LIB = None; L.lib()
                    continue
                if k.startswith("eng:"):              # engine attribute, e.g. "eng:decode_mem_attn=0"
                    saved.append((k, getattr(eng, k[4:])))
                    setattr(eng, k[4:], int(x))
                    continue
                saved.append((k, L.get_option(k)))
                L.set_option(k, int(x))
            t_short, _ = run(2)
            t_long, out = run(steps + 2)
            for k, x in saved:
                if k.startswith("eng:"):
                    setattr(eng, k[4:], x)
                else:
                    L.set_option(k, x)
            if r:
                res[v].append((t_long - t_short) / steps)
            toks[v] = out
    ref = toks[variants[0]]
    for v in variants:
        xs = sorted(res[v])
        same = bool((toks[v] == ref).all()) if toks[v].shape == ref.shape else False
        print(f"{v:40s} median {xs[len(xs) // 2]:7.4f} ms/step   min {xs[0]:7.4f}  max {xs[-1]:7.4f}   tokens == first variant: {same}")


if __name__ == "__main__":
    main()
