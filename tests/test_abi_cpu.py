"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol
that include/gtars_amd.h declares, and refuses to compute without a GPU (no
silent CPU fallback)."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gtars_[a-z0-9_]+)\s*\(", src)))


def _headers():
    return [h for h in os.listdir(os.path.join(ROOT, "include")) if h.endswith(".h")]


def test_library_exports_every_declared_symbol():
    import gtars_amd._lib as L

    out = subprocess.check_output(["nm", "-D", "--defined-only", L.LIB_PATH], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    for h in _headers():
        declared = _declared_symbols(h)
        assert declared, h
        missing = [s for s in declared if s not in exported]
        assert not missing, f"{h}: declared but not exported: {missing}"


def test_ctypes_binding_covers_the_header():
    import gtars_amd._lib as L

    declared = set(_declared_symbols("gtars_amd.h"))
    assert declared == set(L.EXPORTED_SYMBOLS)
    declared_host = set(_declared_symbols("gtars_amd_host.h"))
    assert declared_host == set(L.EXPORTED_HOST_SYMBOLS)
    # test / diagnostics hooks live in a header of their own, outside the drop-in boundary
    assert set(_declared_symbols("gtars_amd_debug.h")) == set(L.EXPORTED_DEBUG_SYMBOLS)
    assert not [s for s in declared | declared_host if s.startswith("gtars_debug_")]


def test_library_is_gfx950_code_object():
    import gtars_amd._lib as L

    blob = open(L.LIB_PATH, "rb").read()
    assert b"gfx950" in blob


@pytest.mark.skipif(__import__("gtars_amd").device_count() > 0, reason="needs a box WITHOUT a GPU")
def test_no_device_is_a_loud_error_not_a_fallback():
    import gtars_amd

    with pytest.raises(gtars_amd.NoDeviceError):
        gtars_amd.OverlapIndex([0], [1], [5])
    with pytest.raises(gtars_amd.NoDeviceError):
        gtars_amd.IgdIndex([0], [1], [5], [0])


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "gtars_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "gtars_oracle" not in src, f


def test_synth_generators_are_deterministic_and_shaped():
    from gtars_amd import synth

    u = synth.make_universe(10_000)
    u2 = synth.make_universe(10_000)
    assert all((u[k] == u2[k]).all() for k in u)
    assert (u["end"] > u["start"]).all()
    # non-overlapping within a chromosome and karyotype ordered
    same = u["chrom"][1:] == u["chrom"][:-1]
    assert (u["start"][1:][same] >= u["end"][:-1][same]).all()
    assert (np.diff(u["chrom"].astype(np.int64)) >= 0).all()
    q = synth.make_queries(u, 50_000)
    assert len(q["chrom"]) == 50_000
    unk = (q["chrom"] == synth.UNKNOWN_CHROM).mean()
    assert 0.0 < unk < 0.01
    # splitmix stream == scalar recurrence
    s = synth.splitmix_stream(42, 4)
    import oracle

    r = oracle.SplitMix64(42)
    assert [int(x) for x in s] == [r.next() for _ in range(4)]


def test_gtars_alias_namespace_resolves_to_the_hip_package():
    """gtars-python/src/lib.rs:27-104 registers gtars.tokenizers / models / utils / lola in sys.modules; user code written
    against the reference imports from there.  The alias package makes the same imports land in gtars_amd."""
    import importlib

    import gtars
    import gtars_amd
    from gtars.lola import RegionDB, build_restricted_universe, check_universe, redefine_user_sets, run_lola  # noqa: F401
    from gtars.models import Region, RegionSet  # noqa: F401
    from gtars.tokenizers import Tokenizer, tokenize_fragment_file  # noqa: F401
    from gtars.utils import read_tokens_from_gtok, read_tokens_from_gtok_as_strings, write_tokens_to_gtok  # noqa: F401

    assert importlib.import_module("gtars.tokenizers") is gtars_amd.tokenizers
    assert Tokenizer is gtars_amd.tokenizers.Tokenizer and RegionSet is gtars_amd.models.RegionSet
    assert gtars.__version__ == gtars_amd.__version__
    import pytest

    with pytest.raises(ModuleNotFoundError):
        importlib.import_module("gtars.refget")
