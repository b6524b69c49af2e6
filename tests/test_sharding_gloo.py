"""N > 1 path on CPU: two gloo ranks run the DRIVERS of gtars_amd.sharding (chromosome-bucket and range sharding of an IGD
database, LOLA support vectors, range-sharded tokenization with the CSR all-gather) and must reproduce the
single-process result exactly.  There is no GPU here, so the engine behind the drivers is a stand-in with the same four
methods as sharding.HipEngine, computing on the CPU oracle; everything else -- LPT buckets, query routing, the
collectives -- is the code that runs on the GPUs.  (tests/test_gpu_host.py runs the same drivers with HipEngine, two
ranks sharing one GPU.)"""
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


class OracleEngine:
    """CPU stand-in for sharding.HipEngine (test infrastructure: the product never imports oracle/)."""

    name = "oracle"

    def igd(self, chrom, start, end, file_idx, n_chrom, n_files):
        import oracle

        g = oracle.Igd()
        g.add_arrays(chrom, start, end, np.zeros(len(chrom), dtype=int), file_idx)
        g.n_files = n_files
        g.finalize()
        return g

    def igd_count(self, g, qc, qs, qe, min_overlap, binary):
        import torch

        f = g.count_region_hits if binary else g.count_set_overlaps
        return torch.from_numpy(f(qc, qs, qe, min_overlap, n_files=g.n_files).astype(np.int64))

    def upload(self, qc, qs, qe):
        return [np.ascontiguousarray(x) for x in (qc, qs, qe)]

    def igd_count_resident(self, g, d, min_overlap, binary, hits=None, sync=True):
        r = self.igd_count(g, d[0], d[1], d[2], min_overlap, binary)
        if hits is not None:
            hits.copy_(r)
            return hits
        return r

    def index(self, chrom, start, end, n_chrom):
        import oracle

        return oracle.Index(chrom, start, end, None, n_chrom=n_chrom)

    def tokenize(self, ix, qc, qs, qe):
        import torch

        off, ids = ix.tokenize(qc, qs, qe)
        return torch.from_numpy(off.astype(np.int64)), torch.from_numpy(ids.view(np.int32).copy())


def sharded_checks(engine, rank, world):
    """The assertions shared by the CPU (gloo + oracle engine) and the GPU (HipEngine) multi-rank tests.
    Returns a dict of booleans; the single-process expectation is computed with the same engine, unsharded."""
    import torch

    from gtars_amd import sharding, synth

    out = {}
    n_chrom, F = synth.N_CHROM, 37
    db = synth.make_igd_db(20_000, F)
    bq = synth.make_background_queries(6_001)
    bq["chrom"][::97] = 0xFFFFFFFF  # unknown chromosomes hit nothing and belong to no bucket
    whole = engine.igd(db["chrom"], db["start"], db["end"], db["file"], n_chrom, F)
    for binary in (False, True):
        exp = engine.igd_count(whole, bq["chrom"], bq["start"], bq["end"], 1, binary).cpu()
        for mode in ("bucket", "range"):
            sdb = sharding.ShardedIgd(engine, db, n_chrom, F, mode=mode, balance_with=[bq["chrom"]])
            got = sdb.count(bq, 1, binary).cpu()
            got2 = sdb.count_resident(sdb.upload_local(bq), 1, binary).cpu()
            out[f"igd_{mode}_{'binary' if binary else 'pairwise'}"] = bool(torch.equal(got, exp) and torch.equal(got2, exp))
            if mode == "bucket" and world > 1:
                out["bucket_db_is_cut"] = sdb.local_intervals < len(db["chrom"])
            if mode == "bucket":
                # the all-to-all hand-over of the queries (every rank looks at its 1 / world of the batch only): the same totals,
                # and the rank's share is exactly the queries of its chromosomes (as a multiset: they arrive grouped by sender)
                got3 = sdb.count_resident(sdb.upload_local(bq, exchange=True), 1, binary).cpu()
                mine, via = sdb.local_queries(bq), sdb.exchange_queries(bq)
                rows = lambda d: sorted(zip(d["chrom"].tolist(), d["start"].tolist(), d["end"].tolist()))
                out[f"igd_bucket_exchange_{'binary' if binary else 'pairwise'}"] = bool(torch.equal(got3, exp) and rows(mine) == rows(via))
    # LOLA support vectors + contingency cells: two user sets, one all-reduce
    uni = synth.make_universe(30_000, seed=3)
    rng = np.random.default_rng(5)
    users = []
    for n_user in (2_000, 777):
        sel = np.sort(rng.choice(len(uni["chrom"]), n_user, replace=False))
        users.append({k: uni[k][sel] for k in ("chrom", "start", "end")})
    exp_pooled = engine.igd_count(whole, uni["chrom"], uni["start"], uni["end"], 1, True).cpu()
    exp_sup = torch.stack([engine.igd_count(whole, us["chrom"], us["start"], us["end"], 1, True).cpu() for us in users])
    sdb = sharding.ShardedIgd(engine, db, n_chrom, F, mode="bucket", balance_with=[uni["chrom"]])
    sup, pooled = sharding.lola_counts_sharded(sdb, users, uni)
    out["lola_support"] = bool(torch.equal(sup.cpu(), exp_sup) and torch.equal(pooled.cpu(), exp_pooled))
    a, b, c, d = sharding.contingency(sup, pooled, [len(us["chrom"]) for us in users], len(uni["chrom"]))
    out["lola_cells"] = bool(torch.equal((a + b).cpu(), exp_pooled.expand_as(a)) and int((a + b + c + d - len(uni["chrom"])).abs().sum()) == 0)
    # tokenization: replicated universe index, contiguous query ranges, all-gatherv of the CSR
    u = synth.make_universe(5_000)
    q = synth.make_queries(u, 40_001)
    ix = engine.index(u["chrom"], u["start"], u["end"], n_chrom)
    exp_off, exp_ids = engine.tokenize(ix, q["chrom"], q["start"], q["end"])
    g_off, g_ids = sharding.tokenize_sharded(engine, ix, q, gather=True)
    out["tokenize_gather"] = bool(torch.equal(g_off.cpu(), exp_off.cpu()) and torch.equal(g_ids.cpu(), exp_ids.cpu()))
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        # a second gather of the same shapes through the first one's plan: no size exchange, same CSR
        l_off, l_ids = sharding.tokenize_sharded(engine, ix, q, gather=False)
        a_off, a_ids, plan = sharding.all_gather_csr_device(l_off, l_ids, int(l_ids.numel()), return_plan=True)
        b_off, b_ids = sharding.all_gather_csr_device(l_off, l_ids, int(l_ids.numel()), plan=plan)
        out["tokenize_gather_plan_reuse"] = bool(torch.equal(a_off.cpu(), exp_off.cpu()) and torch.equal(b_off.cpu(), exp_off.cpu())
                                                 and torch.equal(a_ids.cpu(), exp_ids.cpu()) and torch.equal(b_ids.cpu(), exp_ids.cpu()))
    return out


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, sharded_checks(OracleEngine(), rank, world)))
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


def test_chrom_buckets_lpt():
    """hg38-shaped weights over 8 ranks: every chromosome has exactly one owner, the heaviest rank stays within
    4/3 - 1/(3m) of the optimum (Graham's bound) -- in fact within a few per cent here -- and the map is deterministic."""
    from gtars_amd import synth
    from gtars_amd.sharding import chrom_buckets, chrom_weights

    w = synth.CHROM_SIZES // 1000
    for world in (1, 2, 4, 8):
        owner = chrom_buckets(w, world)
        assert owner.shape == (synth.N_CHROM,) and owner.min() >= 0 and owner.max() < world
        load = np.bincount(owner, weights=w, minlength=world)
        assert load.sum() == w.sum()
        assert load.max() <= max(w.sum() / world * 1.08, w.max())
        assert np.array_equal(owner, chrom_buckets(w, world))
    assert chrom_buckets([5, 5, 5], 2).tolist() == [0, 1, 0]  # ties: lowest chromosome first, lowest rank first
    cw = chrom_weights(3, np.array([0, 0, 2, 0xFFFFFFFF], dtype=np.uint32), np.array([1], dtype=np.uint32))
    assert cw.tolist() == [2, 1, 1]


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_ranks_match_single_process(world):
    """the sharding drivers on `world` gloo ranks (8 = the node the 1 / 2 / 4 / 8-GPU curve is run on: more ranks than the largest
    chromosome bucket count any 2-rank run reaches, uneven query ranges, ranks without a chromosome of the small test inputs)"""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=480) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == list(range(world))
    for rank, checks in res.items():
        assert checks and all(checks.values()), (rank, checks)
    assert res[0]["bucket_db_is_cut"]


def test_single_process_drivers_need_no_process_group():
    checks = sharded_checks(OracleEngine(), 0, 1)
    assert all(checks.values()), checks


# ------------------------------------------------------------ the fragment pipeline over ranks (BASELINE config 5)


def oracle_fragment_pipeline(paths, om, otok):
    """CPU stand-in for fragsplit.fragsplit_tokenize_files (test infrastructure): oracle.fragsplit over the files in the order
    given, then tokenize_fragment_file's rules (utils/fragments.rs:61-82) on every cluster's routed lines ->
    {label: (barcodes in first-seen order, offsets, ids)}, labels in byte order like the product."""
    import oracle

    d = os.path.dirname(paths[0]) if paths else "."
    routed = oracle.fragsplit(d, om, file_order=[os.path.basename(p) for p in paths])
    out = {}
    for label in sorted(om.cluster_labels, key=lambda x: x.encode()):
        per = {}
        for line in routed[label]:
            if line.startswith("#"):
                continue
            parts = line.split()
            per.setdefault(parts[3], []).extend(otok.encode_regions([(parts[0], int(parts[1]), int(parts[2]))]))
        offs = np.zeros(len(per) + 1, dtype=np.uint64)
        if per:
            offs[1:] = np.cumsum([len(v) for v in per.values()])
        ids = np.array([x for v in per.values() for x in v], dtype=np.uint32)
        out[label] = (list(per), offs, ids)
    return out


def same_cluster_results(a, b):
    if list(a) != list(b):
        return False
    for k in a:
        if list(a[k][0]) != list(b[k][0]) or not np.array_equal(np.asarray(a[k][1], dtype=np.uint64), np.asarray(b[k][1], dtype=np.uint64)):
            return False
        if not np.array_equal(np.asarray(a[k][2], dtype=np.uint32), np.asarray(b[k][2], dtype=np.uint32)):
            return False
    return True


def write_fragment_inputs(tmp, files=11, frags=400, clusters=5):
    from gtars_amd import synth

    u = synth.make_universe(3_000)
    return synth.write_config5_inputs(str(tmp), u, files, frags, clusters)[:3]


def fragment_pipeline_checks(files_dir, mapping, tokenizer, run, expected, world):
    """gathered == the single-process result on every rank; rank-local results + manifests tile the file list and merge to
    the same thing"""
    import torch.distributed as dist

    from gtars_amd import sharding

    out = {}
    got = sharding.fragsplit_tokenize_sharded(files_dir, mapping, tokenizer, gather=True, run=run)
    out["fragments_gathered"] = same_cluster_results(got, expected)
    local = sharding.fragsplit_tokenize_sharded(files_dir, mapping, tokenizer, gather=False, run=run)
    man = local.pop("__manifest__")
    parts = [None] * world
    if world > 1:
        dist.all_gather_object(parts, (man["files"], local))
    else:
        parts = [(man["files"], local)]
    runs = [p[0] for p in parts]
    out["fragments_runs_tile_the_files"] = bool(runs[0][0] == 0 and all(a[1] == b[0] for a, b in zip(runs, runs[1:]))
                                                and runs[-1][1] == len(os.listdir(files_dir)))
    out["fragments_every_rank_has_files"] = all(r[1] > r[0] for r in runs)
    out["fragments_local_merge"] = same_cluster_results(sharding.merge_cluster_results([p[1] for p in parts]), expected)
    return out


def _fragment_worker(rank, world, port, q, ub, fd, mp):
    import torch.distributed as dist

    import oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        om, otok = oracle.OracleBarcodeMap(mp), oracle.OracleTokenizer(ub)
        from gtars_amd import fragsplit

        expected = oracle_fragment_pipeline(fragsplit.list_fragment_files(fd), om, otok)
        q.put((rank, fragment_pipeline_checks(fd, om, otok, oracle_fragment_pipeline, expected, world)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_ranks_fragment_pipeline(tmp_path, world):
    """config 5's file-parallel split on gloo ranks (CPU stand-in pipeline = the oracle): contiguous runs of the sorted file
    list, per-rank results merged per barcode in run order == the single-process result (barcodes recur across files); with 8
    ranks over 11 files some ranks get one file and the merge crosses seven run borders."""
    import torch.multiprocessing as mp

    ub, fd, mpth = write_fragment_inputs(tmp_path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fragment_worker, args=(r, world, port, q, ub, fd, mpth)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=480) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == list(range(world))
    for rank, checks in res.items():
        assert checks and all(checks.values()), (rank, checks)


def test_file_runs_and_merge():
    from gtars_amd.sharding import file_runs, merge_cluster_results

    for sizes in ([5] * 7, [100, 1, 1, 1], [1, 1], [], [3] * 10, [0, 0, 9, 0]):
        for w in (1, 2, 3, 4, 8):
            r = file_runs(sizes, w)
            assert len(r) == w and r[0][0] == 0 and r[-1][1] == len(sizes)
            assert all(a[1] == b[0] and a[0] <= a[1] for a, b in zip(r, r[1:] + [(len(sizes), len(sizes))]))
    assert file_runs([5] * 8, 4) == [(0, 2), (2, 4), (4, 6), (6, 8)]
    a = {"k": (["x", "y"], np.array([0, 2, 3], dtype=np.uint64), np.array([1, 2, 3], dtype=np.uint32))}
    b = {"k": (["y", "z"], np.array([0, 1, 1], dtype=np.uint64), np.array([9], dtype=np.uint32))}
    m = merge_cluster_results([a, b])["k"]
    assert m[0] == ["x", "y", "z"] and m[1].tolist() == [0, 2, 4, 4] and m[2].tolist() == [1, 2, 3, 9]
