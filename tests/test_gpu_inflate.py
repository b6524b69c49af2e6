"""GPU tests of the device-side DEFLATE decoder (csrc/inflate_dev.hip, one wave per stream; entry gtars_debug_inflate_streams in
include/gtars_amd_debug.h): the prototype of the fused fragment pipeline's gzip stage on the GPU (gtars-fragsplit/src/split.rs:84-131
reads fragment files through flate2's MultiGzDecoder).  The checker is zlib -- the library behind flate2 -- through Python's zlib
module: bit-exact output, bytes consumed, and a refusal (status != 0) for whatever is not a valid stream that fits."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15):
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strategy)
    return co.compress(data) + co.flush()


def _run(streams, caps):
    """streams: list of raw DEFLATE byte strings; caps: output capacity per stream -> (status, out_len, consumed, outputs)"""
    import torch
    from gtars_amd import _lib

    dev = torch.device("cuda:0")
    in_off, out_off, ci, co = [], [], 0, 0
    for s, c in zip(streams, caps):
        in_off.append(ci)
        ci += (len(s) + 48 + 15) & ~15
        out_off.append(co)
        co += (c + 15) & ~15
    blob = np.zeros(ci + 64, dtype=np.uint8)
    for o, s in zip(in_off, streams):
        blob[o:o + len(s)] = np.frombuffer(s, dtype=np.uint8)
    mk = lambda a, dt: torch.from_numpy(np.asarray(a, dtype=dt)).to(dev)
    d_blob = torch.from_numpy(blob).to(dev)
    d_out = torch.full((co + 64,), 0xEE, dtype=torch.uint8, device=dev)
    d_in_off, d_in_len = mk(in_off, np.uint64), mk([len(s) for s in streams], np.uint32)
    d_out_off, d_cap = mk(out_off, np.uint64), mk(caps, np.uint32)
    n = len(streams)
    d_len, d_used, d_st = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(3))
    rc = _lib.lib.gtars_debug_inflate_streams(d_blob.data_ptr(), d_in_off.data_ptr(), d_in_len.data_ptr(), d_out.data_ptr(), d_out_off.data_ptr(),
                                              d_cap.data_ptr(), n, d_len.data_ptr(), d_used.data_ptr(), d_st.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    st, ln, used = d_st.cpu().numpy(), d_len.cpu().numpy(), d_used.cpu().numpy()
    outs = [out[o:o + (c + 15 & ~15)].tobytes() for o, c in zip(out_off, caps)]
    return st, ln, used, outs


def _fragment_text(rng, n):
    acgt = np.array(list("ACGT"))
    codes = ["".join(acgt[(b >> (2 * k)) & 3] for k in range(16)) + "-1" for b in rng.integers(0, 1 << 32, 200)]
    start = np.sort(rng.integers(0, 240_000_000, n))
    return "".join(f"chr{rng.integers(1, 23)}\t{s}\t{s + rng.integers(50, 900)}\t{codes[rng.integers(0, 200)]}\t{rng.integers(1, 4)}\n" for s in start).encode()


def test_device_inflate_matches_zlib_on_every_block_type_and_edge_shape():
    rng = np.random.default_rng(11)
    text = _fragment_text(rng, 30000)  # ~1.3 MB: several dynamic blocks, matches across block borders
    far = bytes(rng.integers(0, 256, 32768, dtype=np.uint8))
    cases = [
        (b"", 6, zlib.Z_DEFAULT_STRATEGY),
        (b"A", 6, zlib.Z_DEFAULT_STRATEGY),
        (b"\n" * 70000, 6, zlib.Z_DEFAULT_STRATEGY),                      # distance 1, length 258 runs
        (b"abc" * 40000, 9, zlib.Z_DEFAULT_STRATEGY),                     # distance < length
        (bytes(rng.integers(0, 256, 70000, dtype=np.uint8)), 6, zlib.Z_DEFAULT_STRATEGY),  # incompressible: long codes, stored blocks
        (far + far + far[:1000], 9, zlib.Z_DEFAULT_STRATEGY),             # matches at the maximum distance (32768)
        (text, 1, zlib.Z_DEFAULT_STRATEGY), (text, 6, zlib.Z_DEFAULT_STRATEGY), (text, 9, zlib.Z_DEFAULT_STRATEGY),
        (text[:200000], 0, zlib.Z_DEFAULT_STRATEGY),                      # stored blocks only
        (text[:200000], 6, zlib.Z_FIXED),                                 # the fixed code
        (text[:200000], 6, zlib.Z_HUFFMAN_ONLY),                          # literals only: a distance code of no symbols
        (text[:200000], 6, zlib.Z_RLE),
        (text[:4095], 6, zlib.Z_DEFAULT_STRATEGY), (text[:4096], 6, zlib.Z_DEFAULT_STRATEGY), (text[:4097], 6, zlib.Z_DEFAULT_STRATEGY),  # the flush piece
        (text[:32767], 6, zlib.Z_DEFAULT_STRATEGY), (text[:32769], 6, zlib.Z_DEFAULT_STRATEGY),  # the window
    ]
    streams = [_deflate(d, lv, stg) for d, lv, stg in cases]
    st, ln, used, outs = _run(streams, [len(d) for d, _, _ in cases])
    for k, (d, _, _) in enumerate(cases):
        assert st[k] == 0, (k, st[k])
        assert ln[k] == len(d) and used[k] == len(streams[k]), (k, ln[k], len(d), used[k], len(streams[k]))
        assert outs[k][:len(d)] == d, k
        assert set(outs[k][len(d):]) <= {0xEE}, k  # nothing written behind the stream's bytes


def test_device_inflate_random_streams_against_zlib():
    rng = np.random.default_rng(12)
    datas = []
    for k in range(96):
        kind = k % 4
        n = int(rng.integers(0, 150000))
        if kind == 0:
            d = _fragment_text(rng, n // 45 + 1)
        elif kind == 1:
            d = bytes(rng.integers(0, 4, n, dtype=np.uint8) + 65)            # a four-letter alphabet
        elif kind == 2:
            d = bytes(rng.integers(0, 256, n // 8 + 1, dtype=np.uint8)) * int(rng.integers(1, 12))  # long repeats
        else:
            d = bytes(np.repeat(rng.integers(0, 256, n // 50 + 1, dtype=np.uint8), rng.integers(1, 100, n // 50 + 1)))  # runs
        datas.append(d)
    streams = [_deflate(d, int(rng.integers(1, 10))) for d in datas]
    st, ln, used, outs = _run(streams, [len(d) for d in datas])
    for k, d in enumerate(datas):
        assert st[k] == 0 and ln[k] == len(d) and used[k] == len(streams[k]) and outs[k][:len(d)] == d, (k, st[k], ln[k], len(d))


def test_device_inflate_refuses_what_does_not_fit_or_is_not_a_stream():
    rng = np.random.default_rng(13)
    text = _fragment_text(rng, 5000)
    good = _deflate(text)
    broken = bytearray(good)
    for i in range(40, len(broken), 97):
        broken[i] ^= 0x5A
    streams = [good, good, good[:len(good) // 2], bytes(broken), b"\x07" + good, b"\xff" * 64, good + b"trailing bytes"]
    caps = [len(text), len(text) - 1, len(text), len(text), len(text), 4096, len(text)]
    st, ln, used, outs = _run(streams, caps)
    assert st[0] == 0 and outs[0][:len(text)] == text
    assert st[1] == 5                                   # one byte short of capacity
    assert st[2] != 0                                   # the stream ends early
    assert st[3] != 0 or outs[3][:len(text)] != text    # damaged: refused, or (what the caller's CRC check is for) different bytes
    assert st[4] != 0                                   # block type 3
    assert st[5] != 0
    assert st[6] == 0 and used[6] == len(good) and outs[6][:len(text)] == text  # bytes behind the final block are not consumed
    for k in range(len(streams)):
        assert set(outs[k][caps[k]:]) <= {0xEE}, k     # never a byte beyond the capacity
