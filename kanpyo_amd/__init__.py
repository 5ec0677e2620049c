"""kanpyo_amd -- MI355X-native drop-in for kanpyo::Tokenizer::tokenize().

Host-side mirror of the reference's public API for that one path
(src/tokenizer.rs, src/token.rs) over the C ABI of libkanpyo_gpu.so
(include/kanpyo_gpu.h).  The HIP extension is mandatory: there is no CPU
fallback, importing the tokenizer without the built library raises.
"""
from .token import Token, TokenClass  # noqa: F401
from .dict import Dict  # noqa: F401
from .tokenizer import Tokenizer, TOKEN_DTYPE  # noqa: F401

__all__ = ["Token", "TokenClass", "Dict", "Tokenizer", "TOKEN_DTYPE"]
