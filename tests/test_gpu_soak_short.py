"""A short run of the soak fuzzers inside the GPU suite (round-5 verdict: most of the randomised evidence lived outside the
driver-run suite).  The fuzzers proper -- tests/soak/fuzz_tokenize.py, fuzz_igd.py, fuzz_fragments.py: hundreds to a thousand
cases each, minutes of run time -- stay outside; here 120 + 100 + 3 + 60 cases from their own generators with seeds of this file,
every case bit-exact against the oracle like there (their assertions are the test)."""
import importlib.util
import os
import shutil
import tempfile

import pytest

pytestmark = pytest.mark.gpu

SOAK = os.path.join(os.path.dirname(os.path.abspath(__file__)), "soak")


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(SOAK, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture
def clean_switches():
    """the fuzzers set GTARS_* switches per case: whatever they left is removed, and the library takes a new snapshot"""
    before = {k: v for k, v in os.environ.items() if k.startswith("GTARS_")}
    yield
    import gtars_amd

    for k in [k for k in os.environ if k.startswith("GTARS_")]:
        if k not in before:
            del os.environ[k]
    os.environ.update(before)
    gtars_amd.reload_env()


def test_short_soak_tokenize(clean_switches):
    fz = _load("fuzz_tokenize")
    ids = 0
    for seed in range(120):
        ids += fz.one(910_000 + seed)[2]
    assert ids > 0


def test_short_soak_igd(clean_switches):
    fz = _load("fuzz_igd")
    for seed in range(100):
        fz.one(920_000 + seed)
    for seed in range(3):
        fz.one_big(930_000 + seed)


def test_short_soak_fragments(clean_switches):
    import oracle
    from gtars_amd.tokenizers import Tokenizer

    fz = _load("fuzz_fragments")
    ub = os.path.join(os.path.dirname(SOAK), "golden", "tokenizers", "peaks.bed")
    chroms = sorted({l.split()[0] for l in open(ub) if l.strip()})
    tok, otok = Tokenizer.from_bed(ub), oracle.OracleTokenizer(ub)
    tmp = tempfile.mkdtemp(prefix="gtars_fuzzfrag_")
    try:
        for seed in range(60):
            fz.one(940_000 + seed, tmp, tok, otok, chroms, ub)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
