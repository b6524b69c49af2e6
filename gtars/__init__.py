"""``gtars`` -- drop-in import namespace over gtars_amd.

The reference's Python package registers its sub-modules in sys.modules (gtars-python/src/lib.rs:27-104), so user
code reads ``from gtars.tokenizers import Tokenizer``, ``from gtars.models import RegionSet``,
``from gtars.utils import read_tokens_from_gtok``, ``from gtars.lola import RegionDB, run_lola``.  This package makes
those imports resolve to the MI355X implementation (gtars_amd.*) without edits to the calling code.  Sub-modules the
hot path does not cover (refget, vrs, reftx, genomic_distributions) are not provided: importing them raises
ModuleNotFoundError, not a silent stub.  ``gtars.igd``, ``gtars.scoring`` and ``gtars.fragsplit`` are additive (the
reference exposes those crates through Rust / the CLI only).
"""
import sys as _sys

import gtars_amd as _impl
from gtars_amd import fragsplit, igd, lola, models, scoring, tokenizers, utils  # noqa: F401

for _name in ("tokenizers", "models", "utils", "lola", "igd", "scoring", "fragsplit"):
    _sys.modules[f"{__name__}.{_name}"] = getattr(_sys.modules[__name__], _name)

__version__ = _impl.__version__
__all__ = ["tokenizers", "models", "utils", "lola", "igd", "scoring", "fragsplit"]
