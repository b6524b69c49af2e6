"""N > 1 path on CPU: two gloo ranks shard the query batch, compute their slice with the oracle (the
stand-in checker -- there is no GPU here), and the collectives of gtars_amd.sharding must reproduce the
single-process result exactly."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist

    import oracle
    from gtars_amd import sharding, synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        u = synth.make_universe(5_000)
        qs = synth.make_queries(u, 40_001)
        ix = oracle.Index(u["chrom"], u["start"], u["end"], None, n_chrom=synth.N_CHROM)
        lo, hi = sharding.shard_range(len(qs["chrom"]), rank, world)
        off, ids = ix.tokenize(qs["chrom"][lo:hi], qs["start"][lo:hi], qs["end"][lo:hi])
        goff, gids = sharding.all_gather_csr(off, ids)
        full_off, full_ids = ix.tokenize(qs["chrom"], qs["start"], qs["end"])
        ok_tok = np.array_equal(goff, full_off) and np.array_equal(gids, full_ids)
        # IGD support vectors: replicated DB, sharded queries, one all-reduce
        db = synth.make_igd_db(20_000, 37)
        g = oracle.Igd()
        g.add_arrays(db["chrom"], db["start"], db["end"], np.zeros(len(db["chrom"]), dtype=int), db["file"])
        g.n_files = 37
        g.finalize()
        bq = synth.make_background_queries(6_001)
        lo, hi = sharding.shard_range(len(bq["chrom"]), rank, world)
        part = g.count_region_hits(bq["chrom"][lo:hi], bq["start"][lo:hi], bq["end"][lo:hi], 1, n_files=37)
        total = sharding.all_reduce_hits(part)
        ok_igd = np.array_equal(total, g.count_region_hits(bq["chrom"], bq["start"], bq["end"], 1, n_files=37))
        q.put((rank, bool(ok_tok), bool(ok_igd)))
    finally:
        dist.destroy_process_group()


def test_shard_range_tiles_the_batch():
    from gtars_amd.sharding import shard_range

    for n in (0, 1, 7, 8, 1_000_003):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_two_rank_gloo_matches_single_process():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True, True), (1, True, True)]
