#!/usr/bin/env python3
"""Device-side DEFLATE (csrc/inflate_dev.hip, one wave per stream) on fragment-file text: correctness against the original bytes
and throughput, beside zlib on the host.  N streams of config-5-like text (tools/fragsplit_bench.py's generator shape: chrom, start,
end, barcode, count), compressed with zlib at the given level.
usage: python tools/r06_inflate_bench.py [n_streams=1000] [fragments_per_stream=10000] [level=6]"""
import ctypes as C, json, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gtars_amd import _lib

def make_text(rng, n):
    chrom = rng.integers(1, 23, n)
    start = np.sort(rng.integers(0, 240_000_000, n))
    end = start + rng.integers(50, 900, n)
    bc = rng.integers(0, 500, n)
    cnt = rng.integers(1, 4, n)
    acgt = np.array(list("ACGT"))
    codes = ["".join(acgt[(b >> (2 * k)) & 3] for k in range(16)) + "-1" for b in rng.integers(0, 1 << 32, 500)]
    return "".join(f"chr{c}\t{s}\t{e}\t{codes[b]}\t{k}\n" for c, s, e, b, k in zip(chrom, start, end, bc, cnt)).encode()

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    level = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    rng = np.random.default_rng(5)
    distinct = min(n, 16)
    texts = [make_text(rng, per) for _ in range(distinct)]
    extra = [b"", b"A", b"\n" * 70000, bytes(rng.integers(0, 256, 50000, dtype=np.uint8)), b"abc" * 40000, texts[0][:100]]  # edge shapes
    comp = []
    for t in texts + extra:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp.append(co.compress(t) + co.flush())
    co = zlib.compressobj(0, zlib.DEFLATED, -15)  # stored blocks
    extra.append(texts[0][:200000]); comp.append(co.compress(extra[-1]) + co.flush())
    co = zlib.compressobj(6, zlib.DEFLATED, -15, 9, zlib.Z_FIXED)  # fixed code
    extra.append(texts[0][:100000]); comp.append(co.compress(extra[-1]) + co.flush())
    all_texts = texts + extra
    ids = [i % distinct for i in range(n)] + list(range(distinct, len(all_texts)))
    in_off, in_len, out_off, out_cap = [], [], [], []
    ci = co_ = 0
    for i in ids:
        in_off.append(ci); in_len.append(len(comp[i])); ci += (len(comp[i]) + 48 + 15) & ~15
        out_off.append(co_); out_cap.append(len(all_texts[i])); co_ += (len(all_texts[i]) + 15) & ~15
    blob = np.zeros(ci + 64, dtype=np.uint8)
    for o, i in zip(in_off, ids):
        blob[o:o + len(comp[i])] = np.frombuffer(comp[i], dtype=np.uint8)
    dev = torch.device("cuda:0")
    d_blob = torch.from_numpy(blob).to(dev)
    d_out = torch.zeros(co_ + 64, dtype=torch.uint8, device=dev)
    mk = lambda a, dt: torch.from_numpy(np.asarray(a, dtype=dt)).to(dev)
    d_in_off, d_in_len, d_out_off, d_out_cap = mk(in_off, np.uint64), mk(in_len, np.uint32), mk(out_off, np.uint64), mk(out_cap, np.uint32)
    ns = len(ids)
    d_len, d_used, d_st = (torch.zeros(ns, dtype=torch.int32, device=dev) for _ in range(3))
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: _lib.lib.gtars_debug_inflate_streams(d_blob.data_ptr(), d_in_off.data_ptr(), d_in_len.data_ptr(), d_out.data_ptr(), d_out_off.data_ptr(),
                                                        d_out_cap.data_ptr(), ns, d_len.data_ptr(), d_used.data_ptr(), d_st.data_ptr(), st)
    assert call() == 0
    torch.cuda.synchronize()
    stt, ln, used = d_st.cpu().numpy(), d_len.cpu().numpy(), d_used.cpu().numpy()
    out = d_out.cpu().numpy()
    bad = 0
    for k, i in enumerate(ids):
        ok = stt[k] == 0 and ln[k] == len(all_texts[i]) and used[k] == len(comp[i]) and out[out_off[k]:out_off[k] + ln[k]].tobytes() == all_texts[i]
        if not ok:
            bad += 1
            if bad < 6:
                print("MISMATCH stream", k, "text", i, "status", stt[k], "len", ln[k], "want", len(all_texts[i]), "used", used[k], "of", len(comp[i]), flush=True)
    raw = C.CDLL(_lib.lib._name)
    if hasattr(raw, "gtars_debug_inflate_stats"):  # (-DINF_STATS=1 build)
        buf = (C.c_ulonglong * 8)()
        raw.gtars_debug_inflate_stats(buf, 1)
        call(); torch.cuda.synchronize()
        raw.gtars_debug_inflate_stats(buf, 0)
        v = list(buf)
        sy = max(v[0] + v[1], 1)
        print(json.dumps({"per_stream": {"literals": v[0] // ns, "matches": v[1] // ns, "blocks": v[2] // ns, "match_bytes": v[6] // ns, "slow_path_symbols": v[7] // ns,
                                        "cycles_block_setup": v[3] // ns, "cycles_symbols": v[4] // ns, "cycles_total": v[5] // ns},
                          "cycles_per_symbol": round(v[4] / sy, 1), "cycles_per_block_setup": round(v[3] / max(v[2], 1), 1), "bytes_per_symbol": round((v[0] + v[6]) / sy, 2)}))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); call(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    t0 = time.perf_counter()
    for i in range(distinct):
        zlib.decompress(comp[i], -15)
    host = (time.perf_counter() - t0) / distinct
    total_out, total_in = sum(out_cap), sum(in_len)
    ms = sorted(ts)[len(ts) // 2]
    print(json.dumps({"streams": ns, "fragments_per_stream": per, "zlib_level": level, "compressed_MB": round(total_in / 1e6, 1), "text_MB": round(total_out / 1e6, 1),
                      "mismatches": bad, "device_ms": round(ms, 2), "device_runs_ms": [round(x, 2) for x in ts], "device_text_GBps": round(total_out / ms / 1e6, 2),
                      "per_stream_MBps": round(total_out / ns / ms / 1e3, 1), "host_zlib_one_thread_ms_per_stream": round(host * 1e3, 3),
                      "host_zlib_16_threads_est_ms": round(host * 1e3 * ns / 16, 1)}))
    sys.exit(1 if bad else 0)
main()
