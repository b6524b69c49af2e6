#!/usr/bin/env python3
"""Config 5's fused call (48 files x 1e5 fragments) five times in one process, for a kernel trace:
   rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/frag_trace -- python3 tools/r05_frag_trace.py"""
import os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gtars_amd import synth
from gtars_amd.fragsplit import BarcodeToClusterMap, fragsplit_tokenize
from gtars_amd.tokenizers import Tokenizer

tmp = tempfile.mkdtemp(prefix="gtars_fragtrace_")
try:
    u = synth.make_universe(100_000)
    ub, fd, mp, _ = synth.write_config5_inputs(tmp, u, 48, 100_000, 20)
    tok, m = Tokenizer.from_bed(ub), BarcodeToClusterMap.from_file(mp)
    ts = []
    for _ in range(6):
        t = time.perf_counter(); r = fragsplit_tokenize(fd, m, tok, as_arrays=True); ts.append(time.perf_counter() - t); r = None
    print("calls_ms", [round(x * 1e3, 1) for x in ts])
finally:
    shutil.rmtree(tmp, ignore_errors=True)
