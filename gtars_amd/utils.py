"""``gtars.utils`` mirror: .gtok token files (gtars-python/src/utils/mod.rs:72-100 ->
gtars-io/src/gtok.rs:125-210): ``"GTOK"`` + width flag (0x01 = u16, 0x02 = u32) + little-endian tokens."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np

from ._lib import check, lib, ptr, take_u32


def write_tokens_to_gtok(filename: str, tokens: Sequence[int]) -> None:
    t = np.ascontiguousarray(tokens, dtype=np.uint32)
    check(lib.gtars_gtok_write(str(filename).encode(), ptr(t), len(t)))


def read_tokens_from_gtok(filename: str) -> List[int]:
    p, n = C.c_void_p(), C.c_uint64()
    check(lib.gtars_gtok_read(str(filename).encode(), C.byref(p), C.byref(n)))
    return [int(v) for v in take_u32(p, n.value)]


def read_tokens_from_gtok_as_strings(filename: str) -> List[str]:
    return [str(v) for v in read_tokens_from_gtok(filename)]
