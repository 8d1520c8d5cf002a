"""Chapter/event parsing of generated text (dvc.py:186-212, demo_vid2seq.py:170-197): pure Python, no torch."""
from __future__ import annotations

import re
from typing import List

_SPLIT = re.compile(r"(?<!<)\s+(?!>)")
_TIME = re.compile(r"\<time\=(\d+)\>")


def parse_chapters(text: str, duration: float, num_bins: int) -> List[dict]:
    """Pair consecutive ``<time=k>`` tokens into {sentence, timestamp:[start,end]} events;
    t = k * duration / (num_bins - 1); events with end <= start or no text are dropped; three time tokens in a row
    are not split into two events."""
    seqs = _SPLIT.split(text)
    starts = [j for j in range(len(seqs) - 1) if seqs[j][:6] == "<time=" and seqs[j + 1][:6] == "<time="]
    out: List[dict] = []
    last = -2
    for n, idx in enumerate(starts):
        if idx == last + 1:
            continue
        stop = starts[n + 1] if n < len(starts) - 1 else len(seqs)
        words = [seqs[k] for k in range(idx + 2, stop) if seqs[k] != "<time="]
        if not words:
            continue
        ms, me = _TIME.search(seqs[idx]), _TIME.search(seqs[idx + 1])
        assert ms, seqs[idx]
        assert me, seqs[idx + 1]
        start = float(int(ms.group(1))) * float(duration) / float(num_bins - 1)
        end = float(int(me.group(1))) * float(duration) / float(num_bins - 1)
        if end <= start:
            continue
        out.append({"sentence": " ".join(words), "timestamp": [start, end]})
        last = idx
    return out
