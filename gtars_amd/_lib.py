"""ctypes binding of libgtars_amd.so (the C ABI in include/gtars_amd.h).

There is no fallback: if the shared library is missing this module raises at
import time, and every compute call fails with GTARS_ERR_NO_DEVICE when no
MI355X is visible.  The CPU oracle under ``oracle/`` is never imported here.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GTARS_AMD_LIB") or os.path.join(_HERE, "libgtars_amd.so")

GTARS_OK = 0
ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_CAPACITY, ERR_IO, ERR_PARSE, ERR_EMPTY, ERR_CONFIG, ERR_INTERNAL = range(1, 10)

KIND_BITS = 0
KIND_AILIST = 1
UNKNOWN_CHROM = 0xFFFFFFFF


class GtarsError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"[gtars_amd status {status}] {message}")
        self.status = status
        self.message = message


class NoDeviceError(GtarsError):
    pass


class CapacityError(GtarsError):
    @property
    def needed(self) -> int:
        """the element count the library asked for ("... need N")"""
        import re

        m = re.search(r"need (\d+)", self.message)
        return int(m.group(1)) if m else 0


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950).  gtars_amd has no CPU fallback."
    )

# One process must hold ONE HIP runtime.  PyTorch wheels bundle their own libamdhip64.so.7; if this library
# pulled in /opt/rocm's copy first, torch would later fail with "No HIP GPUs are available".  Importing torch
# first makes the dynamic linker resolve our NEEDED libamdhip64.so.7 to the copy torch already mapped.
if os.environ.get("GTARS_AMD_NO_TORCH") != "1":
    try:
        import torch  # noqa: F401
    except ImportError:  # torch is plumbing (device buffers, torch.distributed), not a hard dependency
        pass

lib = C.CDLL(LIB_PATH)

vp, u32, u64, i32, i64 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_int64
pp = C.POINTER(C.c_void_p)
pu64 = C.POINTER(C.c_uint64)

_SIG = {
    "gtars_last_error": (C.c_char_p, []),
    "gtars_version": (C.c_char_p, []),
    "gtars_device_count": (C.c_int, []),
    "gtars_free": (None, [vp]),
    "gtars_index_build": (C.c_int, [vp, vp, vp, vp, u64, u32, C.c_int, pp]),
    "gtars_index_free": (None, [vp]),
    "gtars_index_len": (u64, [vp]),
    "gtars_index_n_chrom": (u32, [vp]),
    "gtars_index_kind": (C.c_int, [vp]),
    "gtars_index_device": (C.c_int, [vp]),
    "gtars_index_chrom_len": (u64, [vp, u32]),
    "gtars_index_stored": (C.c_int, [vp, u32, vp, vp, vp]),
    "gtars_index_insert": (C.c_int, [vp, u32, u32, u32, u32, vp]),
    "gtars_index_seek": (C.c_int, [vp, u32, u32, u32, vp, vp, u64, vp]),
    "gtars_index_max_len": (u32, [vp, u32]),
    "gtars_index_n_sublists": (u64, [vp, u32]),
    "gtars_index_sublist_offsets": (C.c_int, [vp, u32, vp]),
    "gtars_tokenize_device": (C.c_int, [vp, vp, vp, vp, u64, vp, vp, u64, pu64, vp]),
    "gtars_fill_device": (C.c_int, [vp, vp, vp, vp, u64, vp, vp, vp]),
    "gtars_fill_device_n": (C.c_int, [vp, vp, vp, vp, u64, vp, vp, u64, vp]),
    "gtars_tokenize_device_ex": (C.c_int, [vp, vp, vp, vp, u64, vp, vp, u64, pu64, vp, C.c_int]),
    "gtars_histogram_u32_device": (C.c_int, [vp, u64, u32, vp, vp]),
    "gtars_histogram_rows_device": (C.c_int, [vp, vp, vp, u64, u32, u32, u32, vp, vp]),
    "gtars_tokenize": (C.c_int, [vp, vp, vp, vp, u64, vp, pp, pu64]),
    "gtars_tokenize_into": (C.c_int, [vp, vp, vp, vp, u64, vp, vp, u64, pu64]),
    "gtars_count_overlaps_device": (C.c_int, [vp, vp, vp, vp, u64, C.c_int, i32, vp, vp]),
    "gtars_count_overlaps": (C.c_int, [vp, vp, vp, vp, u64, C.c_int, i32, vp]),
    "gtars_bits_count_device": (C.c_int, [vp, vp, vp, vp, u64, vp, vp]),
    "gtars_bits_count": (C.c_int, [vp, vp, vp, vp, u64, vp]),
    "gtars_any_overlaps": (C.c_int, [vp, vp, vp, vp, u64, C.c_int, i32, vp]),
    "gtars_find_overlaps": (C.c_int, [vp, vp, vp, vp, u64, C.c_int, i32, vp, pp, pp, pp, pu64]),
    "gtars_find_overlap_indices": (C.c_int, [vp, vp, vp, vp, u64, C.c_int, i32, vp, pp, pu64]),
    "gtars_subset_by_overlaps": (C.c_int, [vp, vp, vp, vp, u64, C.c_int, i32, pp, pp, pp, pu64]),
    "gtars_subset_source_indices": (C.c_int, [vp, vp, vp, vp, u64, C.c_int, i32, pp, pu64]),
    "gtars_mark_overlapped_device": (C.c_int, [vp, vp, vp, vp, u64, C.c_int, i32, vp, vp]),
    "gtars_igd_build": (C.c_int, [vp, vp, vp, vp, vp, u64, u32, u32, pp]),
    "gtars_igd_free": (None, [vp]),
    "gtars_igd_len": (u64, [vp]),
    "gtars_igd_n_files": (u32, [vp]),
    "gtars_igd_device": (C.c_int, [vp]),
    "gtars_igd_total_records": (u64, [vp, i32]),
    "gtars_igd_export": (C.c_int, [vp, vp, vp, vp, vp, vp]),
    "gtars_igd_count_device": (C.c_int, [vp, vp, vp, vp, u64, i32, C.c_int, vp, vp]),
    "gtars_igd_count": (C.c_int, [vp, vp, vp, vp, u64, i32, C.c_int, vp]),
    "gtars_igd_count_sets_device": (C.c_int, [vp, vp, vp, vp, vp, C.c_uint32, i32, C.c_int, vp, vp]),
    "gtars_igd_count_sets": (C.c_int, [vp, vp, vp, vp, vp, C.c_uint32, i32, C.c_int, vp]),
    "gtars_igd_count_per_query": (C.c_int, [vp, vp, vp, vp, u64, i32, vp]),
    "gtars_igd_find_pairs": (C.c_int, [vp, vp, vp, vp, u64, i32, pp, pp, pu64]),
    "gtars_lola_contingency_device": (C.c_int, [vp, vp, u64, i64, i64, vp, vp, vp, vp, vp]),
    "gtars_prof_enable": (None, [C.c_int]),
    "gtars_prof_reset": (None, []),
    "gtars_prof_read": (C.c_int, [vp, vp, vp, C.c_int]),
}

pvp = C.POINTER(C.c_void_p)


class FragmentTokens(C.Structure):
    _fields_ = [
        ("n_barcodes", C.c_uint64),
        ("barcodes", C.POINTER(C.c_char_p)),
        ("offsets", C.POINTER(C.c_uint64)),
        ("ids", C.POINTER(C.c_uint32)),
    ]


cstr = C.c_char_p
# include/gtars_amd_host.h
_HOST_SIG = {
    "gtars_regionset_from_bed": (C.c_int, [cstr, pp]),
    "gtars_regionset_from_arrays": (C.c_int, [vp, vp, vp, vp, u64, pp]),
    "gtars_regionset_free": (None, [vp]),
    "gtars_regionset_dense_ids": (C.c_int, [vp, pp, vp]),
    "gtars_regionset_len": (u64, [vp]),
    "gtars_regionset_header": (cstr, [vp]),
    "gtars_regionset_n_chrom": (u32, [vp]),
    "gtars_regionset_chrom_name": (cstr, [vp, u32]),
    "gtars_regionset_chrom_ids": (vp, [vp]),
    "gtars_regionset_starts": (vp, [vp]),
    "gtars_regionset_ends": (vp, [vp]),
    "gtars_regionset_rest": (cstr, [vp, u64]),
    "gtars_regionset_count_overlaps": (C.c_int, [vp, vp, C.c_int, C.c_int, i32, vp]),
    "gtars_regionset_any_overlaps": (C.c_int, [vp, vp, C.c_int, C.c_int, i32, vp]),
    "gtars_regionset_find_overlaps": (C.c_int, [vp, vp, C.c_int, C.c_int, i32, vp, pp, pu64]),
    "gtars_tokenizer_from_auto": (C.c_int, [cstr, pp]),
    "gtars_tokenizer_from_config": (C.c_int, [cstr, pp]),
    "gtars_tokenizer_from_bed": (C.c_int, [cstr, pp]),
    "gtars_tokenizer_free": (None, [vp]),
    "gtars_tokenizer_vocab_size": (u64, [vp]),
    "gtars_tokenizer_kind": (C.c_int, [vp]),
    "gtars_tokenizer_id_to_token": (cstr, [vp, u32]),
    "gtars_tokenizer_token_to_id": (i64, [vp, cstr]),
    "gtars_tokenizer_vocab_token": (cstr, [vp, u64, C.POINTER(C.c_uint32)]),
    "gtars_tokenizer_special_token": (cstr, [vp, C.c_int]),
    "gtars_tokenizer_region_name": (cstr, [vp, cstr]),
    "gtars_tokenizer_region_score": (C.c_double, [vp, cstr]),
    "gtars_tokenizer_chrom_id": (i64, [vp, cstr]),
    "gtars_tokenizer_n_chrom": (u32, [vp]),
    "gtars_tokenizer_chrom_name": (cstr, [vp, u32]),
    "gtars_tokenizer_index": (vp, [vp]),
    "gtars_tokenizer_encode_regionset": (C.c_int, [vp, vp, pp, pu64]),
    "gtars_tokenizer_encode_arrays": (C.c_int, [vp, vp, vp, vp, u64, pp, pu64]),
    "gtars_tokenizer_encode_ids": (C.c_int, [vp, vp, vp, vp, u64, vp, pp, pu64]),
    "gtars_tokenizer_tokenize_fragment_file": (C.c_int, [vp, cstr, C.POINTER(C.POINTER(FragmentTokens))]),
    "gtars_fragment_tokens_free": (None, [C.POINTER(FragmentTokens)]),
    "gtars_fragment_tokens_barcodes_joined": (C.c_int, [C.POINTER(FragmentTokens), pp, pu64]),
    "gtars_barcode_map_from_file": (C.c_int, [cstr, pp]),
    "gtars_barcode_map_free": (None, [vp]),
    "gtars_barcode_map_len": (u64, [vp]),
    "gtars_barcode_map_n_clusters": (u32, [vp]),
    "gtars_barcode_map_cluster_label": (C.c_char_p, [vp, u32]),
    "gtars_barcode_map_lookup": (C.c_char_p, [vp, cstr]),
    "gtars_fragsplit": (C.c_int, [cstr, vp, cstr, pu64, pu64]),
    "gtars_fragsplit_tokenize": (C.c_int, [vp, cstr, vp, C.POINTER(C.POINTER(C.POINTER(FragmentTokens))), pu64]),
    "gtars_fragsplit_tokenize_files": (C.c_int, [vp, vp, u64, vp, C.POINTER(C.POINTER(C.POINTER(FragmentTokens))), pu64]),
    "gtars_host_threads": (u32, [u32]),
    "gtars_read_file": (C.c_int, [cstr, pp, pu64]),
    "gtars_fragsplit_last_stages": (None, [vp]),
    "gtars_gtok_write": (C.c_int, [cstr, vp, u64]),
    "gtars_gtok_read": (C.c_int, [cstr, pp, pu64]),
    "gtars_fragments_read": (C.c_int, [cstr, pp]),
    "gtars_fragments_read_strict": (C.c_int, [cstr, pp]),
    "gtars_bed3_lines_read": (C.c_int, [cstr, pp]),
    "gtars_format_hit_lines": (C.c_int, [vp, vp, vp, vp, u64, pp, pu64]),
    "gtars_fragments_free": (None, [vp]),
    "gtars_fragments_len": (u64, [vp]),
    "gtars_fragments_n_chrom": (u32, [vp]),
    "gtars_fragments_n_barcodes": (u32, [vp]),
    "gtars_fragments_chrom_name": (cstr, [vp, u32]),
    "gtars_fragments_barcode_name": (cstr, [vp, u32]),
    "gtars_fragments_chrom_ids": (vp, [vp]),
    "gtars_fragments_starts": (vp, [vp]),
    "gtars_fragments_ends": (vp, [vp]),
    "gtars_fragments_barcode_ids": (vp, [vp]),
    "gtars_igddb_from_bed_files": (C.c_int, [vp, u64, pp]),
    "gtars_igddb_from_bed_dir": (C.c_int, [cstr, pp]),
    "gtars_igddb_free": (None, [vp]),
    "gtars_igddb_n_files": (u32, [vp]),
    "gtars_igddb_n_contigs": (u32, [vp]),
    "gtars_igddb_file_name": (cstr, [vp, u32]),
    "gtars_igddb_file_num_regions": (u32, [vp, u32]),
    "gtars_igddb_file_avg_width": (C.c_double, [vp, u32]),
    "gtars_igddb_chrom_id": (i64, [vp, cstr]),
    "gtars_igddb_chrom_name": (cstr, [vp, u32]),
    "gtars_igddb_engine": (vp, [vp]),
    "gtars_igddb_count_regionset": (C.c_int, [vp, vp, i32, C.c_int, vp]),
    "gtars_igddb_from_arrays": (C.c_int, [vp, u32, vp, vp, vp, vp, vp, u64, vp, vp, vp, u32, pp]),
    "gtars_igddb_save": (C.c_int, [vp, cstr, i32]),
    "gtars_igddb_load": (C.c_int, [cstr, pp, C.POINTER(C.c_int32)]),
    "gtars_lola_stats": (C.c_int, [vp, vp, vp, vp, u64, u64, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "gtars_lola_rank": (C.c_int, [vp, vp, vp, u64, vp, vp, vp, vp, vp]),
    "gtars_lola_fdr": (C.c_int, [vp, vp, u64, vp]),
    "gtars_lola_fisher_pvalue": (C.c_double, [u64, u64, u64, u64, C.c_int]),
    "gtars_lola_odds_ratio": (C.c_double, [u64, u64, u64, u64]),
}

# include/gtars_amd_debug.h: test / A-B / diagnostics hooks, not part of the drop-in boundary
_DEBUG_SIG = {
    "gtars_debug_occupy_device": (C.c_int, [vp, u32, u32, u32]),
    "gtars_debug_reload_env": (None, []),
    "gtars_debug_set_handle_device": (C.c_int, [vp, C.c_int, C.c_int]),
    "gtars_debug_inflate_streams": (C.c_int, [vp, vp, vp, vp, vp, vp, u32, vp, vp, vp, vp]),
}

# every symbol the headers declare must resolve -- fail loudly otherwise (GTARS_AMD_LIB_OLDER=1, A/B tooling only: an older
# build loaded through GTARS_AMD_LIB may lack the newest entry points; calling one of those then fails at the call)
_older = bool(os.environ.get("GTARS_AMD_LIB")) and os.environ.get("GTARS_AMD_LIB_OLDER") == "1"
for _table in (_SIG, _HOST_SIG, _DEBUG_SIG):
    for _name, (_res, _args) in _table.items():
        if _older and not hasattr(lib, _name):
            continue
        _fn = getattr(lib, _name)
        _fn.restype = _res
        _fn.argtypes = _args

EXPORTED_SYMBOLS = tuple(_SIG)
EXPORTED_HOST_SYMBOLS = tuple(_HOST_SIG)
EXPORTED_DEBUG_SYMBOLS = tuple(_DEBUG_SIG)


def cstr_array(strings):
    """list[str] -> (char*[] ctypes array, keepalive)."""
    enc = [s.encode("utf-8") if s is not None else None for s in strings]
    arr = (C.c_char_p * max(len(enc), 1))(*enc)
    return arr, enc


def dec(b):
    return None if b is None else b.decode("utf-8", "replace")


def last_error() -> str:
    return (lib.gtars_last_error() or b"").decode("utf-8", "replace")


def check(status: int):
    if status == GTARS_OK:
        return
    msg = last_error()
    if status == ERR_NO_DEVICE:
        raise NoDeviceError(status, msg)
    if status == ERR_CAPACITY:
        raise CapacityError(status, msg)
    if status == ERR_IO:
        raise FileNotFoundError(msg)
    if status in (ERR_PARSE, ERR_EMPTY, ERR_CONFIG, ERR_INVALID_ARG):
        raise ValueError(msg)
    raise GtarsError(status, msg)


def device_count() -> int:
    return int(lib.gtars_device_count())


def as_u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


def ptr(a: np.ndarray):
    return C.c_void_p(a.ctypes.data) if a.size else C.c_void_p(0)


def take_u32(p: C.c_void_p, n: int) -> np.ndarray:
    """Copy a library-allocated u32 array into numpy and free it."""
    try:
        if n == 0 or not p.value:
            return np.zeros(0, dtype=np.uint32)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n,)).copy()
    finally:
        if p.value:
            lib.gtars_free(p)


def prof_read():
    cap = 64
    names = (C.c_char_p * cap)()
    ms = (C.c_double * cap)()
    launches = (C.c_uint64 * cap)()
    n = lib.gtars_prof_read(C.cast(names, vp), C.cast(ms, vp), C.cast(launches, vp), cap)
    return {names[i].decode(): {"total_ms": ms[i], "launches": int(launches[i])} for i in range(min(n, cap))}


def reload_env() -> None:
    """The library reads its GTARS_* switches ONCE, into a snapshot taken at first use; a process that changes one afterwards
    (tests, A/B harnesses) calls this to make the library take a new snapshot.  No library call may be in flight."""
    lib.gtars_debug_reload_env()
