"""Token / TokenClass (reference src/token.rs:3-55)."""
from __future__ import annotations

import enum
from dataclasses import dataclass


class TokenClass(enum.IntEnum):
    """src/token.rs:3-8.  Integer values are the C ABI's KGPU_CLASS_*."""

    Dummy = 0
    Known = 1
    Unknown = 2


@dataclass(frozen=True)
class Token:
    """src/token.rs:10-18; equality compares all six fields (src/token.rs:44-53)."""

    id: int
    class_: TokenClass
    position: int  # byte position
    start: int  # char position
    end: int  # char position
    surface: str

    def length(self) -> int:  # src/token.rs:39-41
        return self.end - self.start
