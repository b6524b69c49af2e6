#!/usr/bin/env python3
"""Config 5, the fused call alone, repeated: for same-box A/Bs of library variants (GTARS_AMD_LIB).
usage: r05_frag_stream_ab.py [lib | NAME=VALUE ...]   -- inputs are written once, every variant (a library, or an environment
switch of the in-tree library such as GTARS_ZLIB_INFLATE=1) runs in a child process over the same files."""
import json, os, shutil, statistics, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(tmp):
    import ctypes as C
    import gtars_amd
    from gtars_amd import _lib
    from gtars_amd.fragsplit import BarcodeToClusterMap
    from gtars_amd.tokenizers import Tokenizer
    out = {}
    for name in sorted(os.listdir(tmp)):
        d = os.path.join(tmp, name)
        p = json.load(open(os.path.join(d, "paths.json")))
        tok = Tokenizer.from_bed(p["universe"])
        m = BarcodeToClusterMap.from_file(p["map"])
        fd = p["fragments"]
        ts, stages = [], []
        for i in range(9):
            o_, nr_ = C.POINTER(C.POINTER(_lib.FragmentTokens))(), C.c_uint64()
            t = time.perf_counter()
            _lib.check(_lib.lib.gtars_fragsplit_tokenize(tok._h, os.fspath(fd).encode(), m._h, C.byref(o_), C.byref(nr_)))
            ts.append(time.perf_counter() - t)
            st = (C.c_double * 12)()
            _lib.lib.gtars_fragsplit_last_stages(st)
            stages.append([round(x * 1e3, 1) for x in st[2:12]])
            for c in range(m.n_clusters()): _lib.lib.gtars_fragment_tokens_free(o_[c])
            _lib.lib.gtars_free(o_)
        ts = ts[2:]
        out[name] = {"median_ms": round(statistics.median(ts) * 1e3, 2), "min_ms": round(min(ts) * 1e3, 2), "reads": int(nr_.value),
                     "calls_ms": [round(x * 1e3, 1) for x in ts], "stages_ms[inflate,append,device,device_tail,regroup|h2d,parse,gather,tokenize,d2h]": stages[2:]}
    print(json.dumps({"lib": os.environ.get("GTARS_AMD_LIB", "in-tree"), "switches": {k: v for k, v in os.environ.items() if k.startswith("GTARS_") and k != "GTARS_AMD_LIB"}, **out}), flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    from gtars_amd import synth
    tmp = tempfile.mkdtemp(prefix="gtars_fragab_")
    try:
        u = synth.make_universe(100_000)
        for name, K, n in (("48x100k", 48, 100_000), ("1000x10k", 1000, 10_000)):
            d = os.path.join(tmp, name); os.makedirs(d)
            ub, fd, mp, _ = synth.write_config5_inputs(d, u, K, n, 20)
            json.dump({"universe": ub, "fragments": fd, "map": mp}, open(os.path.join(d, "paths.json"), "w"))
        for rep in range(2):
            for lib in sys.argv[1:] or [""]:
                env = dict(os.environ)
                if "=" in lib and not os.path.exists(lib): env[lib.split("=", 1)[0]] = lib.split("=", 1)[1]
                elif lib: env["GTARS_AMD_LIB"] = os.path.abspath(lib)
                subprocess.run([sys.executable, os.path.abspath(__file__), "--child", tmp], env=env, check=False)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
