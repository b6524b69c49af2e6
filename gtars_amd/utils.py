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


def read_fragments(path: str):
    """Fragment file (``chr start end barcode count``, optionally .gz) -> SoA columns.

    The multi-threaded in-place parser of the host layer (parse_fragment_line,
    gtars-tokenizers/src/utils/fragments.rs:12-40).  Returns a dict with ``chrom`` / ``barcode``
    dictionary codes (u32, first-seen order), ``start`` / ``end`` (u32) and the two name lists.
    """
    import ctypes as C

    import numpy as np

    from . import _lib

    h = C.c_void_p()
    if _lib.lib.gtars_fragments_read(str(path).encode(), C.byref(h)) != 0:
        raise RuntimeError(_lib.last_error())
    try:
        n = int(_lib.lib.gtars_fragments_len(h))

        def col(fn):
            p = fn(h)
            if not n:
                return np.zeros(0, dtype=np.uint32)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n,)).copy()

        return {
            "chrom": col(_lib.lib.gtars_fragments_chrom_ids),
            "start": col(_lib.lib.gtars_fragments_starts),
            "end": col(_lib.lib.gtars_fragments_ends),
            "barcode": col(_lib.lib.gtars_fragments_barcode_ids),
            "chrom_names": [_lib.lib.gtars_fragments_chrom_name(h, i).decode()
                            for i in range(_lib.lib.gtars_fragments_n_chrom(h))],
            "barcode_names": [_lib.lib.gtars_fragments_barcode_name(h, i).decode()
                              for i in range(_lib.lib.gtars_fragments_n_barcodes(h))],
        }
    finally:
        _lib.lib.gtars_fragments_free(h)


def read_file(path: str) -> bytes:
    """The bytes of ``path``, gunzipped iff its extension is ``gz`` -- the reader every file front end of the library goes
    through (``get_dynamic_reader``, gtars-core/src/utils.rs:115-126: flate2's MultiGzDecoder behind the extension): concatenated
    members one after the other, every member's CRC-32 and length checked, a ".gz" without the gzip magic as it is."""
    p, n = C.c_void_p(), C.c_uint64()
    check(lib.gtars_read_file(str(path).encode(), C.byref(p), C.byref(n)))
    try:
        return C.string_at(p, n.value)
    finally:
        lib.gtars_free(p)
