"""Tokenizer factory (model/vid2seq.py:10-18) + a synthetic stand-in for environments without spiece.model."""
from __future__ import annotations

import os
from typing import List


def _get_tokenizer(tokenizer_path, num_bins=0):
    """T5 sentencepiece tokenizer with ``num_bins`` extra ``<time=i>`` tokens (ids 32100..32100+num_bins-1)."""
    if "t5" in tokenizer_path:
        from transformers import T5Tokenizer
        tokenizer = T5Tokenizer.from_pretrained(tokenizer_path, local_files_only=True)
        if num_bins:
            tokenizer.add_tokens(["<time=" + str(i) + ">" for i in range(num_bins)])
        return tokenizer
    raise NotImplementedError(tokenizer_path)


def _clean_up_tokenization(text: str) -> str:
    """PreTrainedTokenizerBase.clean_up_tokenization of transformers 4.28 (applied by decode by default)."""
    for a, b in ((" .", "."), (" ?", "?"), (" !", "!"), (" ,", ","), (" ' ", "'"), (" n't", "n't"), (" 'm", "'m"), (" 's", "'s"),
                 (" 've", "'ve"), (" 're", "'re")):
        text = text.replace(a, b)
    return text


def batch_decode_spaced(tokenizer, seqs, skip_special_tokens: bool = True) -> List[str]:
    """``tokenizer.batch_decode`` with the semantics of the reference's pinned transformers==4.28.0 (requirements.txt:9), whatever version
    is installed: PreTrainedTokenizer._decode there joins runs of ordinary pieces and every ADDED token (the ``<time=k>`` tokens of
    ``_get_tokenizer``) with single spaces, which is what the chapter parser of dvc.py:186-212 splits on.  Newer releases decode added
    tokens without the spaces ("<time=5><time=7> how to mix eggs<time=7>...": no chapter would ever be parsed).  Tokenizers without the
    HF added-token interface (SyntheticTokenizer) are decoded as they are."""
    added = getattr(tokenizer, "added_tokens_encoder", None)
    if added is None or not hasattr(tokenizer, "convert_ids_to_tokens"):
        return tokenizer.batch_decode(seqs, skip_special_tokens=skip_special_tokens)
    special = set(tokenizer.all_special_tokens)
    out = []
    for s in seqs:
        ids = s.tolist() if hasattr(s, "tolist") else list(s)
        subs: List[str] = []
        run: List[str] = []
        for t in tokenizer.convert_ids_to_tokens(ids, skip_special_tokens=skip_special_tokens):
            if t in added and t not in special:
                if run:
                    subs.append(tokenizer.convert_tokens_to_string(run))
                    run = []
                subs.append(t)
            else:
                run.append(t)
        if run:
            subs.append(tokenizer.convert_tokens_to_string(run))
        out.append(_clean_up_tokenization(" ".join(subs)))
    return out


class SyntheticTokenizer:
    """Duck-type of the tokenizer surface Vid2Seq uses (len, pad/eos ids, batch_decode) for synthetic-data runs:
    ``base_vocab`` sentencepiece-like ids followed by ``num_bins`` time tokens.  Token i decodes to ``w<i>``."""
    pad_token_id = 0
    eos_token_id = 1

    def __init__(self, base_vocab: int = 32100, num_bins: int = 100):
        self.base_vocab, self.num_bins = base_vocab, num_bins

    def __len__(self) -> int:
        return self.base_vocab + self.num_bins

    def decode(self, ids, skip_special_tokens: bool = True) -> str:
        words = []
        for t in ids:
            t = int(t)
            if skip_special_tokens and t in (self.pad_token_id, self.eos_token_id):
                continue
            words.append(f"<time={t - self.base_vocab}>" if t >= self.base_vocab else f"w{t}")
        return " ".join(words)

    def batch_decode(self, seqs, skip_special_tokens: bool = True) -> List[str]:
        return [self.decode(s.tolist() if hasattr(s, "tolist") else s, skip_special_tokens) for s in seqs]
