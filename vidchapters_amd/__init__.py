"""vidchapters_amd: MI355X-native (gfx950) engine for the Vid2Seq hot path of antoyang/VidChapters.

Drop-in surface (reference model/__init__.py): ``_get_tokenizer``, ``build_vid2seq_model``, ``Vid2Seq``.
"""
from .modeling import Vid2Seq, build_vid2seq_model  # noqa: F401
from .tokenizer import _get_tokenizer, SyntheticTokenizer  # noqa: F401
from .parse import parse_chapters  # noqa: F401
