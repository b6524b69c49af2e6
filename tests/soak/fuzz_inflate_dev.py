#!/usr/bin/env python3
"""Soak fuzzer of the device-side DEFLATE decoder (csrc/inflate_dev.hip) against zlib: rounds of 64 random streams -- texts of
five kinds, zlib levels 0-9 and strategies -- of which a third are damaged (byte flips, truncation, random bytes).  An intact
stream must come out bit-exact with the exact number of bytes consumed; a damaged one must be refused OR decode to something --
whatever it does, not a byte may be written behind its capacity (guard bytes) and the call must return.
usage: python tests/soak/fuzz_inflate_dev.py [rounds=200] [seed=1]"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_inflate import _run, _fragment_text

def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    strategies = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]
    t0 = time.time()
    n_ok = n_bad = n_refused = 0
    for r in range(rounds):
        datas, streams, caps, intact = [], [], [], []
        for k in range(64):
            kind = int(rng.integers(0, 5))
            n = int(rng.integers(0, 1 << int(rng.integers(4, 19))))
            if kind == 0:
                d = _fragment_text(rng, n // 45 + 1)
            elif kind == 1:
                d = bytes(rng.integers(0, int(rng.integers(1, 6)), n, dtype=np.uint8) + 65)
            elif kind == 2:
                d = bytes(rng.integers(0, 256, n // 8 + 1, dtype=np.uint8)) * int(rng.integers(1, 12))
            elif kind == 3:
                d = bytes(np.repeat(rng.integers(0, 256, n // 50 + 1, dtype=np.uint8), rng.integers(1, 300, n // 50 + 1)))
            else:
                d = bytes(rng.integers(0, 256, n, dtype=np.uint8))
            co = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, -int(rng.integers(9, 16)), int(rng.integers(1, 10)), strategies[int(rng.integers(0, 5))])
            s = co.compress(d) + co.flush()
            ok = True
            mode = int(rng.integers(0, 9))
            if mode == 0 and len(s) > 4:
                b = bytearray(s)
                for _ in range(int(rng.integers(1, 6))):
                    b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
                s, ok = bytes(b), False
            elif mode == 1 and len(s) > 2:
                s, ok = s[:int(rng.integers(0, len(s)))], False
            elif mode == 2:
                s, ok = bytes(rng.integers(0, 256, int(rng.integers(1, 4000)), dtype=np.uint8)), False
            cap = len(d) if ok or rng.integers(0, 2) else int(rng.integers(0, len(d) + 2))
            datas.append(d); streams.append(s); caps.append(cap); intact.append(ok)
        st, ln, used, outs = _run(streams, caps)
        for k in range(64):
            assert set(outs[k][caps[k]:]) <= {0xEE}, ("bytes behind the capacity", r, k)
            if intact[k]:
                assert st[k] == 0 and ln[k] == len(datas[k]) and used[k] == len(streams[k]) and outs[k][:len(datas[k])] == datas[k], ("intact stream", r, k, st[k], ln[k], len(datas[k]))
                n_ok += 1
            else:
                n_bad += 1
                n_refused += int(st[k] != 0)
                if st[k] == 0:  # decoded to something: zlib must then accept the same bytes and agree (a flip can leave a valid stream)
                    try:
                        z = zlib.decompressobj(-15)
                        want = z.decompress(streams[k])
                        if z.eof:
                            assert outs[k][:ln[k]] == want[:ln[k]] and ln[k] == len(want), ("accepted a damaged stream differently from zlib", r, k)
                    except zlib.error:
                        pass  # (zlib refuses what this decoder let through: the caller's length and CRC-32 check is what catches it)
        if (r + 1) % 25 == 0:
            print(f"  {r + 1} rounds, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz_inflate_dev: {n_ok} intact streams bit-exact vs zlib, {n_bad} damaged ones ({n_refused} refused), no byte behind a capacity ({time.time() - t0:.0f} s)")

main()
