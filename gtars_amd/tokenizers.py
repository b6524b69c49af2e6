"""``gtars.tokenizers`` mirror: ``Tokenizer``, ``BatchEncoding``, ``tokenize_fragment_file``.

Same names, argument meaning and error behaviour as the reference's pyo3
module (gtars-python/src/tokenizers/py_tokenizers/mod.rs:14-300, encoding.rs,
utils.rs:9-18; stubs py_src/gtars/tokenizers/__init__.pyi).  Universe / config
parsing and the vocabulary live in the C++ host layer, the overlap search runs
in the HIP kernels; this file only adapts Python objects.

Additive fast path (not in the reference): ``encode_arrays`` / ``encode_ids``
take numpy columns instead of per-object attribute access.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from ._lib import UNKNOWN_CHROM, check, cstr_array, dec, lib, ptr, take_u32
from .models import Region, RegionSet

_SPECIAL = ("unk", "pad", "mask", "cls", "eos", "bos", "sep")


class BatchEncoding:
    """encoding.rs:3-41: mapping with exactly ``input_ids`` and ``attention_mask``."""

    def __init__(self, input_ids: List[int], attention_mask: List[int]):
        self.input_ids = input_ids
        self.attention_mask = attention_mask

    def __getitem__(self, key: str):
        if key == "input_ids":
            return self.input_ids
        if key == "attention_mask":
            return self.attention_mask
        raise KeyError(f"Invalid key: {key}")

    def keys(self):
        return ["input_ids", "attention_mask"]

    def __repr__(self):
        return f"BatchEncoding(input_ids={self.input_ids!r}, attention_mask={self.attention_mask!r})"


def _columns_from_py_any(regions) -> Tuple[Optional[RegionSet], Optional[tuple]]:
    """extract_regions_from_py_any (gtars-python/src/utils/mod.rs:10-70).

    A ``str`` is a path: parsed AND sorted by (chr, start).  Anything else is iterated and
    ``.chr/.start/.end`` are read from every element."""
    if isinstance(regions, str):
        if not os.path.exists(regions):
            raise FileNotFoundError(f"The file {regions} does not exist.")
        try:
            return RegionSet(regions), None
        except RuntimeError as e:
            raise ValueError(str(e)) from None
    if isinstance(regions, RegionSet):
        return regions, None
    chrs, starts, ends = [], [], []
    for x in regions:
        try:
            c, s, e = x.chr, x.start, x.end
        except AttributeError as err:
            raise RuntimeError(f"Region object missing or invalid attribute: {err}") from None
        if not isinstance(c, str):
            raise RuntimeError("Region object missing or invalid 'chr' attribute (expected str)")
        if not (isinstance(s, (int, np.integer)) and 0 <= int(s) <= 0xFFFFFFFF):
            raise RuntimeError("Region object missing or invalid 'start' attribute (expected u32)")
        if not (isinstance(e, (int, np.integer)) and 0 <= int(e) <= 0xFFFFFFFF):
            raise RuntimeError("Region object missing or invalid 'end' attribute (expected u32)")
        chrs.append(c)
        starts.append(int(s))
        ends.append(int(e))
    return None, (chrs, np.asarray(starts, dtype=np.uint32), np.asarray(ends, dtype=np.uint32))


class Tokenizer:
    """gtars.tokenizers.Tokenizer (subclassable, like the pyo3 class)."""

    def __new__(cls, *args, **kwargs):
        self = super().__new__(cls)
        self._h = None
        if args or "path" in kwargs:
            path = args[0] if args else kwargs["path"]
            self._load(lib.gtars_tokenizer_from_auto, path)
        return self

    def __init__(self, *args, **kwargs):
        pass

    # -- construction ---------------------------------------------------------------
    def _load(self, fn, path):
        h = C.c_void_p()
        st = fn(str(path).encode(), C.byref(h))
        if st != 0:
            # anyhow errors surface as RuntimeError in the reference binding
            msg = _lib.last_error()
            if st == _lib.ERR_IO:
                raise FileNotFoundError(msg)
            raise RuntimeError(msg)
        self._h = h
        self._special = [dec(lib.gtars_tokenizer_special_token(h, k)) for k in range(7)]
        self._special_ids = [int(lib.gtars_tokenizer_token_to_id(h, t.encode())) for t in self._special]

    @classmethod
    def _make(cls, fn, path) -> "Tokenizer":
        self = cls.__new__(cls)
        self._load(fn, path)
        return self

    @classmethod
    def from_config(cls, cfg: str) -> "Tokenizer":
        return cls._make(lib.gtars_tokenizer_from_config, cfg)

    @classmethod
    def from_bed(cls, path: str) -> "Tokenizer":
        return cls._make(lib.gtars_tokenizer_from_bed, path)

    @classmethod
    def from_pretrained(cls, path: str) -> "Tokenizer":
        """Local directory holding ``universe.bed.gz`` (tokenizer.rs:104-126); no hub download here."""
        if os.path.isdir(path):
            return cls._make(lib.gtars_tokenizer_from_auto, os.path.join(path, "universe.bed.gz"))
        raise RuntimeError(f"from_pretrained: {path} is not a local directory and there is no network access")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib.gtars_tokenizer_free(self._h)
                self._h = None
        except Exception:
            pass

    # -- core ---------------------------------------------------------------------------
    def _encode_regions(self, regions) -> np.ndarray:
        """Tokenizer::encode (tokenizer.rs:165-171): ids in reference order, [unk] if nothing overlapped."""
        rs, cols = _columns_from_py_any(regions)
        p, n = C.c_void_p(), C.c_uint64()
        if rs is not None:
            check(lib.gtars_tokenizer_encode_regionset(self._h, rs._h, C.byref(p), C.byref(n)))
        else:
            chrs, s, e = cols
            arr, _keep = cstr_array(chrs)
            check(lib.gtars_tokenizer_encode_arrays(self._h, C.cast(arr, C.c_void_p), ptr(s), ptr(e), len(chrs),
                                                    C.byref(p), C.byref(n)))
        return take_u32(p, n.value)

    def tokenize(self, regions) -> List[str]:
        return [self._id_to_token(int(i)) for i in self._encode_regions(regions)]

    def __call__(self, regions) -> BatchEncoding:
        ids = [int(i) for i in self._encode_regions(regions)]
        pad = self.pad_token_id
        return BatchEncoding(ids, [0 if i == pad else 1 for i in ids])

    # additive array fast paths ----------------------------------------------------------
    def chrom_ids(self, chrom_names: Sequence[str]) -> np.ndarray:
        """Map chromosome names to this tokenizer's dense ids (unknown -> 0xFFFFFFFF)."""
        cache: Dict[str, int] = {}
        out = np.empty(len(chrom_names), dtype=np.uint32)
        for i, c in enumerate(chrom_names):
            v = cache.get(c)
            if v is None:
                r = int(lib.gtars_tokenizer_chrom_id(self._h, c.encode()))
                v = UNKNOWN_CHROM if r < 0 else r
                cache[c] = v
            out[i] = v
        return out

    @property
    def chrom_names(self) -> List[str]:
        return [dec(lib.gtars_tokenizer_chrom_name(self._h, i)) for i in range(lib.gtars_tokenizer_n_chrom(self._h))]

    def encode_ids(self, chrom_ids, starts, ends) -> Tuple[np.ndarray, np.ndarray]:
        """-> (offsets u64[n+1], ids u32[H]); per-query CSR, no batch-level unk."""
        c = _lib.as_u32(chrom_ids)
        s = _lib.as_u32(starts)
        e = _lib.as_u32(ends)
        offsets = np.zeros(len(c) + 1, dtype=np.uint64)
        p, n = C.c_void_p(), C.c_uint64()
        check(lib.gtars_tokenizer_encode_ids(self._h, ptr(c), ptr(s), ptr(e), len(c), ptr(offsets), C.byref(p), C.byref(n)))
        return offsets, take_u32(p, n.value)

    def encode_arrays(self, chrom_names: Sequence[str], starts, ends) -> Tuple[np.ndarray, np.ndarray]:
        return self.encode_ids(self.chrom_ids(chrom_names), starts, ends)

    @property
    def engine_index(self) -> int:
        """Borrowed ``gtars_index_t*`` for ``gtars_tokenize_device`` (device-pointer fast path)."""
        return int(lib.gtars_tokenizer_index(self._h) or 0)

    # -- vocabulary ------------------------------------------------------------------------
    def _id_to_token(self, i: int) -> str:
        t = lib.gtars_tokenizer_id_to_token(self._h, i) if 0 <= i <= 0xFFFFFFFF else None
        return dec(t) if t is not None else self.unk_token

    def _token_to_id(self, t: str) -> int:
        r = int(lib.gtars_tokenizer_token_to_id(self._h, t.encode()))
        return self.unk_token_id if r < 0 else r

    def encode(self, tokens: Union[str, List[str]]) -> List[int]:
        if isinstance(tokens, str):
            return [self._token_to_id(tokens)]
        if isinstance(tokens, (list, tuple)) and all(isinstance(t, str) for t in tokens):
            return [self._token_to_id(t) for t in tokens]
        raise ValueError("Invalid input type for convert_ids_to_token")

    def decode(self, ids) -> List[str]:
        if isinstance(ids, (int, np.integer)) and not isinstance(ids, bool):
            return [self._id_to_token(int(ids))]
        try:
            return [self._id_to_token(int(i)) for i in ids]
        except TypeError:
            raise ValueError("Invalid input type for convert_ids_to_token") from None

    def convert_ids_to_tokens(self, id):
        if isinstance(id, (int, np.integer)) and not isinstance(id, bool):
            return self._id_to_token(int(id))
        try:
            return [self._id_to_token(int(i)) for i in id]
        except TypeError:
            raise ValueError("Invalid input type for convert_ids_to_token") from None

    def convert_tokens_to_ids(self, region):
        if isinstance(region, str):
            return self._token_to_id(region)
        if isinstance(region, (list, tuple)) and all(isinstance(t, str) for t in region):
            return [self._token_to_id(t) for t in region]
        raise ValueError("Invalid input type for convert_token_to_ids")

    def get_vocab(self) -> Dict[str, int]:
        out = {}
        idv = C.c_uint32()
        for i in range(self.vocab_size):
            t = lib.gtars_tokenizer_vocab_token(self._h, i, C.byref(idv))
            out[dec(t)] = int(idv.value)
        return out

    def get_special_tokens_mask(self, tokens: List[str]) -> List[bool]:
        sp = set(self._special)
        return [t in sp for t in tokens]

    @property
    def vocab_size(self) -> int:
        return int(lib.gtars_tokenizer_vocab_size(self._h))

    @property
    def special_tokens_map(self) -> Dict[str, str]:
        return {f"{k}_token": v for k, v in zip(_SPECIAL, self._special)}

    def __len__(self) -> int:
        return self.vocab_size

    def __repr__(self) -> str:
        return f"Tokenizer({self.vocab_size} total regions)"


def _add_special_props():
    for k, name in enumerate(_SPECIAL):
        setattr(Tokenizer, f"{name}_token", property(lambda self, k=k: self._special[k]))
        setattr(Tokenizer, f"{name}_token_id", property(lambda self, k=k: self._special_ids[k]))


_add_special_props()


def tokenize_fragment_file(file: str, tokenizer: Tokenizer) -> Dict[str, List[int]]:
    """py_tokenize_fragment_file (gtars-python/src/tokenizers/utils.rs:9-18) -> {barcode: [ids]}."""
    out = C.POINTER(_lib.FragmentTokens)()
    st = lib.gtars_tokenizer_tokenize_fragment_file(tokenizer._h, str(file).encode(), C.byref(out))
    if st != 0:
        raise RuntimeError(_lib.last_error())
    try:
        ft = out.contents
        nb = int(ft.n_barcodes)
        offs = [int(ft.offsets[i]) for i in range(nb + 1)]
        total = offs[nb]
        ids = np.ctypeslib.as_array(ft.ids, shape=(max(total, 1),))[:total].copy()
        return {ft.barcodes[b].decode(): [int(v) for v in ids[offs[b]:offs[b + 1]]] for b in range(nb)}
    finally:
        lib.gtars_fragment_tokens_free(out)


def tokenize_fragment_files(files, tokenizer: Tokenizer, workers: int = 16) -> List[Dict[str, List[int]]]:
    """``tokenize_fragment_file`` over many files with ``workers`` host threads (additive).

    Fragment files are independent: gunzip + parse dominate and run outside the GIL
    (the C call releases it), every thread has its own device workspace, the index is shared read-only.
    Results come back in input order.
    """
    from concurrent.futures import ThreadPoolExecutor

    files = list(files)
    if workers <= 1 or len(files) <= 1:
        return [tokenize_fragment_file(f, tokenizer) for f in files]
    with ThreadPoolExecutor(max_workers=min(workers, len(files))) as ex:
        return list(ex.map(lambda f: tokenize_fragment_file(f, tokenizer), files))


def count_fragments_by_barcode(file: str, tokenizer: Tokenizer) -> Dict[str, Dict[int, int]]:
    """count_fragments_by_barcode (gtars-tokenizers/src/utils/fragments.rs:87-112)."""
    res: Dict[str, Dict[int, int]] = {}
    for bc, ids in tokenize_fragment_file(file, tokenizer).items():
        d = res.setdefault(bc, {})
        for i in ids:
            d[i] = d.get(i, 0) + 1
    return res
