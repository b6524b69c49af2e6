"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed ("nccl" is RCCL on ROCm).

What shards how (BASELINE.json north_star):

* **IGD / LOLA counts** (many-vs-many): the reference database AND the query sets shard by *chromosome bucket*.
  ``chrom_buckets`` assigns chromosomes to ranks by LPT (longest processing time first) over the per-chromosome
  weight ``Ndb_c + Nq_c``; rank r indexes only the database intervals of its chromosomes and counts only the
  queries that fall on them, so the database is cut, not replicated (5e7 intervals -> 6e6 per GPU at N = 8).
  Per-file hit vectors are sums over queries, hence one RCCL all-reduce(SUM) of F int64 values (16 KB at
  F = 2000) yields the global vector on every rank -- exactly what the contingency cells of LOLA need
  (support ``a`` per user set x DB set and the universe's pooled support, gtars-lola/src/enrichment.rs:198-221).
  ``mode="range"`` (database replicated, queries cut into contiguous ranges) is kept as the cross-check: both
  must equal the single-process result.
* **Tokenize / count / find**: every query is independent and the universe index is ~1.6 MB, so the index is
  replicated and the batch is cut into ``world`` contiguous ranges (``shard_range``).  Rank r's CSR is the slice
  of the global CSR of its range; nothing is exchanged unless one consumer needs the whole batch -- then
  ``all_gather_csr_device``: all-gatherv of per-query counts (u32) and ids (u32) over RCCL, offsets rebuilt by a
  device-side cumulative sum.  Payloads stay on the device and stay 32 bits wide.

The drivers take an *engine* (the thing that builds an index and counts on one device).  ``HipEngine`` is the
product path (libgtars_amd.so on ``cuda:<local rank>``); the gloo test on CPU plugs a stand-in behind the same four
methods, so that the sharding logic and the collectives that run there are the ones that run on the GPUs.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

UNKNOWN_CHROM = 0xFFFFFFFF


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) of ``n`` items for ``rank``; ranges tile [0, n) in rank order."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def chrom_weights(n_chrom: int, *chrom_arrays: np.ndarray) -> np.ndarray:
    """Per-chromosome weight = number of intervals (database + queries) on it; unknown chromosomes are ignored."""
    w = np.zeros(n_chrom, dtype=np.int64)
    for c in chrom_arrays:
        c = np.asarray(c)
        c = c[c < n_chrom]
        w += np.bincount(c, minlength=n_chrom)[:n_chrom]
    return w


def chrom_buckets(weights: Sequence[int], world: int) -> np.ndarray:
    """LPT assignment of chromosomes to ranks: heaviest chromosome first, each to the least-loaded rank (ties: lowest
    rank; equal weights: lowest chromosome id first, so every rank computes the same map).  Returns owner[c]."""
    w = np.asarray(weights, dtype=np.int64)
    owner = np.zeros(len(w), dtype=np.int32)
    load = np.zeros(max(world, 1), dtype=np.int64)
    for c in sorted(range(len(w)), key=lambda i: (-int(w[i]), i)):
        r = int(np.argmin(load))
        owner[c] = r
        load[r] += int(w[c])
    return owner


def _select_owned(owner: np.ndarray, rank: int, chrom: np.ndarray) -> np.ndarray:
    c = np.asarray(chrom)
    known = c < len(owner)
    sel = np.zeros(len(c), dtype=bool)
    sel[known] = owner[c[known]] == rank
    return sel


# --------------------------------------------------------------------------- collectives on device tensors

def _backend(group=None) -> str:
    import torch.distributed as dist

    return dist.get_backend(group)


def all_reduce_hits_(hits, group=None):
    """C1: in-place element-wise SUM of per-file hit vectors (int64 tensor on the engine's device) over ranks.
    RCCL reduces the device buffer directly; under gloo (CPU test, or several ranks sharing one GPU) the tensor is
    staged through the host."""
    import torch.distributed as dist

    if _backend(group) == "gloo" and hits.is_cuda:
        t = hits.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        hits.copy_(t)
    else:
        dist.all_reduce(hits, op=dist.ReduceOp.SUM, group=group)
    return hits


class CsrGatherPlan:
    """Per-rank sizes of a CSR all-gather and the index tensors that strip the padding: returned by ``all_gather_csr_device``
    and accepted back by it, so that a caller that gathers batches of the SAME per-rank shape again (every rank must pass the
    plan, or none: the size exchange is a collective) skips the size exchange and its host round trip."""

    def __init__(self, nqs, nhs, dev):
        import torch

        self.nqs, self.nhs = list(nqs), list(nhs)

        def strip(lens):
            pad = max(max(lens), 1)
            idx = [torch.arange(r * pad, r * pad + n, dtype=torch.int64) for r, n in enumerate(lens)]
            return pad, torch.cat(idx).to(dev) if idx else torch.zeros(0, dtype=torch.int64, device=dev)

        self.pad_q, self.sel_q = strip(self.nqs)
        self.pad_h, self.sel_h = strip(self.nhs)


def all_gather_csr_device(offsets, ids, n_hits: int, group=None, plan: Optional[CsrGatherPlan] = None, return_plan: bool = False):
    """C2: the global CSR from per-rank CSRs of consecutive query ranges, on the device.

    offsets: int64[nq_r + 1] (the u64 offsets of the C ABI), ids: int32[>= n_hits] (u32 token ids, bit-identical).
    Returns (global offsets int64[sum nq_r + 1], global ids int32[sum H_r]) (+ the plan with ``return_plan``).  Wire format:
    one all-gather of (nq_r, H_r) -- skipped when a ``plan`` from an earlier call with the same per-rank shapes is passed --,
    one of the per-query counts as int32, one of the ids as int32: all-gatherv emulated by padding to the largest rank (ranks
    are balanced to +-1 query, ids to a few per cent); the padding is stripped by ONE index_select per payload (the plan's
    index tensors), not by a Python loop over ranks."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    stage = _backend(group) == "gloo" and offsets.is_cuda
    dev = offsets.device
    cdev = torch.device("cpu") if stage else dev
    nq = offsets.numel() - 1
    if plan is None:
        meta = torch.tensor([nq, n_hits], dtype=torch.int64, device=cdev)
        metas = torch.empty(2 * world, dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(metas, meta, group=group)
        metas = metas.cpu().view(world, 2)  # (the one host round trip: sizes of the result buffers)
        plan = CsrGatherPlan([int(x) for x in metas[:, 0]], [int(x) for x in metas[:, 1]], cdev)
    elif len(plan.nqs) != world:
        raise ValueError("all_gather_csr_device: the plan was made for another world size")

    def gatherv(x, pad, sel):
        buf = torch.zeros(pad, dtype=torch.int32, device=cdev)
        buf[: x.numel()] = x.to(cdev)
        out = torch.empty(pad * world, dtype=torch.int32, device=cdev)
        dist.all_gather_into_tensor(out, buf, group=group)
        return out.index_select(0, sel).to(dev)

    counts = (offsets[1:] - offsets[:-1]).to(torch.int32)  # hits per query < 2^32
    g_counts = gatherv(counts, plan.pad_q, plan.sel_q)
    g_ids = gatherv(ids[:n_hits], plan.pad_h, plan.sel_h)
    g_off = torch.zeros(sum(plan.nqs) + 1, dtype=torch.int64, device=dev)
    # counts are u32 carried in int32: widen through the unsigned value before summing
    torch.cumsum(g_counts.to(torch.int64) & 0xFFFFFFFF, dim=0, out=g_off[1:])
    return (g_off, g_ids, plan) if return_plan else (g_off, g_ids)


# --------------------------------------------------------------------------- the product engine

class HipEngine:
    """One MI355X through libgtars_amd.so.  Inputs are host numpy columns; results are tensors on ``device``."""

    name = "hip"

    def __init__(self, device=None):
        import torch

        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")

    def _dev(self, a):
        import torch

        return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(self.device)

    def igd(self, chrom, start, end, file_idx, n_chrom: int, n_files: int):
        from .engine import IgdIndex

        return IgdIndex(chrom, start, end, file_idx, n_chrom=n_chrom, n_files=n_files)

    def upload(self, qc, qs, qe):
        """queries resident on the device: what igd_count_resident takes"""
        return [self._dev(x) for x in (qc, qs, qe)]

    def igd_count_resident(self, g, d, min_overlap: int, binary: bool, hits=None, sync: bool = True):
        import torch

        if hits is None:
            hits = torch.zeros(g.n_files, dtype=torch.int64, device=self.device)
        n = d[0].numel()
        if n:
            g.count_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), n, hits.data_ptr(), min_overlap, binary,
                           torch.cuda.current_stream().cuda_stream)
        else:
            hits.zero_()
        if sync:
            torch.cuda.current_stream().synchronize()
        return hits

    def igd_count(self, g, qc, qs, qe, min_overlap: int, binary: bool):
        return self.igd_count_resident(g, self.upload(qc, qs, qe), min_overlap, binary)

    def upload_sets(self, sets):
        """several query sets [(chrom, start, end), ...] as ONE resident batch + their row offsets: what
        igd_count_sets_resident takes (up to four sets then share one pass over the database)"""
        off = np.zeros(len(sets) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(x[0]) for x in sets])
        cols = [np.concatenate([np.ascontiguousarray(x[j], dtype=np.uint32) for x in sets]) if sets else np.zeros(0, dtype=np.uint32)
                for j in range(3)]
        return [self._dev(c) for c in cols], off

    def igd_count_sets_resident(self, g, handle, min_overlap: int, binary: bool, hits=None, sync: bool = True):
        """-> int64[len(sets), n_files] on the device (gtars_igd_count_sets_device)"""
        import torch

        d, off = handle
        if hits is None:
            hits = torch.zeros(len(off) - 1, g.n_files, dtype=torch.int64, device=self.device)
        if d[0].numel():
            g.count_sets_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), off, hits.data_ptr(), min_overlap, binary,
                                torch.cuda.current_stream().cuda_stream)
        else:
            hits.zero_()
        if sync:
            torch.cuda.current_stream().synchronize()
        return hits

    def igd_count_sets(self, g, sets, min_overlap: int, binary: bool):
        return self.igd_count_sets_resident(g, self.upload_sets(sets), min_overlap, binary)

    def index(self, chrom, start, end, n_chrom: int):
        from .engine import OverlapIndex

        return OverlapIndex(chrom, start, end, n_chrom=n_chrom)

    def tokenize(self, ix, qc, qs, qe):
        import torch

        n = len(qc)
        d = [self._dev(x) for x in (qc, qs, qe)]
        offsets = torch.zeros(n + 1, dtype=torch.int64, device=self.device)
        st = torch.cuda.current_stream().cuda_stream
        h = ix.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), n, offsets.data_ptr(), 0, 0, st, sync=True)
        ids = torch.empty(max(h, 1), dtype=torch.int32, device=self.device)
        ix.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), n, offsets.data_ptr(), ids.data_ptr(), ids.numel(),
                           st, sync=True)
        return offsets, ids[:h]


# --------------------------------------------------------------------------- drivers

def _rank_world(group=None) -> Tuple[int, int]:
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def _in_process_group() -> bool:
    """a process group exists (even of one rank): the drivers then run their collectives -- a world-size-1 RCCL group on one
    GPU executes exactly the device-tensor branch the 8-GPU job takes"""
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized()


class ShardedIgd:
    """An IGD database spread over the ranks of a process group.

    mode "bucket": rank r holds the intervals of the chromosomes ``owner == r`` (LPT over Ndb_c + Nq_c when query
    chromosome columns are passed as ``balance_with``); mode "range": every rank holds the whole database."""

    def __init__(self, engine, db, n_chrom: int, n_files: int, mode: str = "bucket",
                 balance_with: Sequence[np.ndarray] = (), group=None):
        """``db``: the database columns (dict of chrom / start / end / file arrays), or a callable returning a fresh
        iterator over row CHUNKS of it (dicts of the same columns; e.g. one chunk per BED file of a LOLA folder, or
        ``synth.igd_db_chunks``).  With chunks a bucket-mode rank keeps only the rows of the chromosomes it owns: it never
        materialises the others' (two passes over the source: chromosome weights, then its rows)."""
        if mode not in ("bucket", "range"):
            raise ValueError("mode must be 'bucket' or 'range'")
        self.engine, self.mode, self.group = engine, mode, group
        self.n_chrom, self.n_files = n_chrom, n_files
        self.rank, self.world = _rank_world(group)
        self.collective = _in_process_group()
        self.owner: Optional[np.ndarray] = None
        cols = ("chrom", "start", "end", "file")
        bucket = mode == "bucket" and self.world > 1
        if callable(db):
            if bucket:
                w = chrom_weights(n_chrom, *balance_with)
                for ch in db():
                    w += chrom_weights(n_chrom, ch["chrom"])
                self.owner = chrom_buckets(w, self.world)
            parts = {k: [] for k in cols}
            for ch in db():
                keep = _select_owned(self.owner, self.rank, ch["chrom"]) if bucket else slice(None)
                for k in cols:
                    parts[k].append(np.ascontiguousarray(ch[k][keep]))
            db = {k: (np.concatenate(v) if v else np.zeros(0, dtype=np.uint32)) for k, v in parts.items()}
        elif bucket:
            self.owner = chrom_buckets(chrom_weights(n_chrom, db["chrom"], *balance_with), self.world)
            keep = _select_owned(self.owner, self.rank, db["chrom"])
            db = {k: np.ascontiguousarray(v[keep]) for k, v in db.items()}
        self.local_intervals = len(db["chrom"])
        self.g = engine.igd(db["chrom"], db["start"], db["end"], db["file"], n_chrom, n_files)

    def local_queries(self, q: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
        """this rank's share of a query set (bucket: its chromosomes; range: its contiguous slice)"""
        if self.world == 1:
            return q
        if self.mode == "bucket":
            sel = _select_owned(self.owner, self.rank, q["chrom"])
            return {k: np.ascontiguousarray(q[k][sel]) for k in ("chrom", "start", "end")}
        lo, hi = shard_range(len(q["chrom"]), self.rank, self.world)
        return {k: q[k][lo:hi] for k in ("chrom", "start", "end")}

    def exchange_queries(self, q: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
        """Bucket mode without the redundant host pass: ``local_queries`` makes EVERY rank filter the WHOLE batch for its
        chromosomes (world x the batch of host work per call).  Here rank r looks only at ITS contiguous 1 / world of the batch
        (``shard_range``), groups those queries by owner rank, and ONE variable-size all-to-all (counts, then the packed
        (chrom, start, end) rows: RCCL on device tensors, gloo through the host) hands every query to the rank that owns its
        chromosome.  Queries on chromosomes nobody owns (unknown ids) are dropped where they are found: no count sees them
        (igd.rs:504-512: an unknown contig contributes nothing).  The rows arrive grouped by sending rank -- per-file counts are
        sums over queries, so their order does not matter.  Same totals as ``local_queries`` (tests/test_sharding_gloo.py).
        ``q``: the global batch as every rank sees it (only the rank's own range of it is read)."""
        if self.world == 1 or self.mode != "bucket" or not self.collective:
            return self.local_queries(q)
        import torch
        import torch.distributed as dist

        lo, hi = shard_range(len(q["chrom"]), self.rank, self.world)
        c = np.asarray(q["chrom"][lo:hi])
        lut = np.full(len(self.owner) + 1, self.world, dtype=np.int64)  # (+ one slot for "no owner")
        lut[: len(self.owner)] = self.owner
        dest = lut[np.minimum(c.astype(np.int64), len(self.owner))]
        keep = dest < self.world
        order = np.argsort(dest[keep], kind="stable")
        counts = np.bincount(dest[keep], minlength=self.world)[: self.world].astype(np.int64)
        rows = np.empty((int(keep.sum()), 3), dtype=np.uint32)
        for j, k in enumerate(("chrom", "start", "end")):
            rows[:, j] = np.asarray(q[k][lo:hi])[keep][order]
        on_device = _backend(self.group) != "gloo" and hasattr(self.engine, "device")
        dev = self.engine.device if on_device else torch.device("cpu")
        cnt_in = torch.from_numpy(counts).to(dev)
        cnt_out = torch.empty(self.world, dtype=torch.int64, device=dev)
        dist.all_to_all_single(cnt_out, cnt_in, group=self.group)
        n_out = cnt_out.cpu().tolist()
        send = torch.from_numpy(rows.view(np.int32)).to(dev)
        recv = torch.empty((sum(n_out), 3), dtype=torch.int32, device=dev)
        dist.all_to_all_single(recv, send, output_split_sizes=n_out, input_split_sizes=counts.tolist(), group=self.group)
        got = recv.cpu().numpy().view(np.uint32)
        return {k: np.ascontiguousarray(got[:, j]) for j, k in enumerate(("chrom", "start", "end"))}

    def count_local(self, q, min_overlap: int = 1, binary: bool = False):
        lq = self.local_queries(q)
        return self.engine.igd_count(self.g, lq["chrom"], lq["start"], lq["end"], min_overlap, binary)

    def upload_local(self, q, exchange: bool = False):
        """this rank's share of ``q``, resident on the engine's device (for repeated counts: ``count_resident``).  ``exchange``:
        through ``exchange_queries`` (a collective: every rank of the group must then call with it)."""
        lq = self.exchange_queries(q) if exchange else self.local_queries(q)
        return self.engine.upload(lq["chrom"], lq["start"], lq["end"])

    def upload_local_sets(self, sets):
        """this rank's shares of several query sets as one resident batch (for ``count_sets_resident``)"""
        local = [self.local_queries(q) for q in sets]
        return self.engine.upload_sets([(l["chrom"], l["start"], l["end"]) for l in local])

    def _timed(self, count, timing):
        """count() enqueues the local counts and returns the hit tensor; then the one all-reduce.  ``timing`` (a dict, or None):
        filled with ``device_ms`` -- this rank's kernels, by events on the stream they were launched on -- and ``collective_ms``
        -- from the end of those kernels to the end of the all-reduce as this rank's stream sees it (RCCL: the wait its stream
        is given; gloo: the staging through the host as well).  Timing synchronises the stream at the end of the call."""
        if timing is None or not hasattr(self.engine, "device"):
            hits = count()
            if self.collective:
                all_reduce_hits_(hits, self.group)
            return hits
        import torch

        st = torch.cuda.current_stream()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(st)
        hits = count()
        e1.record(st)
        if self.collective:
            all_reduce_hits_(hits, self.group)
        e2.record(st)
        e2.synchronize()
        timing["device_ms"] = e0.elapsed_time(e1)
        timing["collective_ms"] = e1.elapsed_time(e2) if self.collective else 0.0
        return hits

    def count_sets_resident(self, handle, min_overlap: int = 1, binary: bool = False, hits=None, timing=None):
        """[len(sets), F] support / hit vectors of the uploaded sets -- one pass over the local database for up to four sets --
        and ONE all-reduce of the whole block (``timing``: see _timed)"""
        return self._timed(lambda: self.engine.igd_count_sets_resident(self.g, handle, min_overlap, binary, hits, sync=False), timing)

    def count_sets_local(self, sets, min_overlap: int = 1, binary: bool = False):
        """local [len(sets), F] block (no collective); engines without a batch form count set by set"""
        import torch

        local = [self.local_queries(q) for q in sets]
        batch = getattr(self.engine, "igd_count_sets", None)
        if batch is not None:
            return batch(self.g, [(l["chrom"], l["start"], l["end"]) for l in local], min_overlap, binary)
        return torch.stack([self.engine.igd_count(self.g, l["chrom"], l["start"], l["end"], min_overlap, binary) for l in local])

    def count_resident(self, handle, min_overlap: int = 1, binary: bool = False, hits=None, timing=None):
        """count + all-reduce on queries uploaded with ``upload_local``; nothing but the F-long vector leaves the device
        (``timing``: see _timed)"""
        return self._timed(lambda: self.engine.igd_count_resident(self.g, handle, min_overlap, binary, hits, sync=False), timing)

    def count(self, q, min_overlap: int = 1, binary: bool = False):
        """global per-file hit vector (int64 tensor on the engine's device), identical on every rank"""
        hits = self.count_local(q, min_overlap, binary)
        if self.collective:
            all_reduce_hits_(hits, self.group)
        return hits


def lola_counts_sharded(sdb: ShardedIgd, user_sets: Sequence[Dict[str, np.ndarray]], universe: Dict[str, np.ndarray],
                        min_overlap: int = 1):
    """Support vectors of LOLA (enrichment.rs:176-221) with the database sharded: ``a[u][f]`` = regions of user set u
    that hit DB set f, ``pooled[f]`` the same for the universe.  The universe and the user sets go through the local database
    as one batch of sets (up to four per pass), and all (1 + U) local vectors are reduced by ONE all-reduce of (1 + U) * F
    int64 values.  Returns (support [U, F], pooled [F]) as int64 tensors on the engine's device."""
    import torch

    stacked = sdb.count_sets_local([universe] + list(user_sets), min_overlap, True)
    if sdb.collective:
        all_reduce_hits_(stacked, sdb.group)
    return stacked[1:], stacked[0]


def contingency(support, pooled, user_sizes: Sequence[int], universe_size: int):
    """a, b, c, d of enrichment.rs:198-221 for every (user set, DB set), int64 (d may go negative as in the reference)."""
    import torch

    a = support
    b = pooled.unsqueeze(0) - a
    sizes = torch.tensor(list(user_sizes), dtype=torch.int64, device=a.device).unsqueeze(1)
    c = sizes - a
    d = universe_size - a - b - c
    return a, b, c, d


def tokenize_sharded(engine, ix, q: Dict[str, np.ndarray], gather: bool = True, group=None):
    """Range-sharded tokenization: this rank's CSR of its query range; with ``gather`` the global CSR on every rank."""
    rank, world = _rank_world(group)
    lo, hi = shard_range(len(q["chrom"]), rank, world)
    offsets, ids = engine.tokenize(ix, q["chrom"][lo:hi], q["start"][lo:hi], q["end"][lo:hi])
    if _in_process_group() and gather:
        return all_gather_csr_device(offsets, ids, int(ids.numel()), group)
    return offsets, ids


# --------------------------------------------------------------------------- fragment pipeline (BASELINE config 5)

def file_runs(sizes: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Cut a list of files (given by their sizes, in visiting order) into ``world`` CONTIGUOUS runs of about equal total size:
    run r = [lo, hi).  Contiguous, not round-robin: the per-cluster results of consecutive runs then merge by plain
    concatenation per barcode (the pipeline's result depends on the file order only through the order of a barcode's ids).
    Deterministic, so every rank computes the same cut from the same directory listing."""
    n = len(sizes)
    tot = float(sum(max(int(x), 1) for x in sizes))
    cuts, acc, r = [0], 0.0, 1
    for i, x in enumerate(sizes):
        acc += max(int(x), 1)
        while r < world and acc >= tot * r / world:
            cuts.append(i + 1)
            r += 1
    while len(cuts) < world + 1:
        cuts.append(n)
    cuts[world] = n
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def merge_cluster_results(parts: Sequence[Dict[str, tuple]]) -> Dict[str, tuple]:
    """Per-cluster ``(barcodes, offsets, ids)`` results of CONSECUTIVE file runs -> the result of the whole list: a barcode's
    ids are the concatenation of its ids in run order (a cluster file holds the routed lines of the files in order, and
    tokenize_fragment_file appends per barcode, utils/fragments.rs:61-82), barcodes in first-seen order."""
    out = {}
    labels = list(parts[0].keys()) if parts else []
    for label in labels:
        index: Dict[str, int] = {}
        chunks: List[List[np.ndarray]] = []
        for part in parts:
            names, offs, ids = part[label]
            for b, name in enumerate(names):
                k = index.get(name)
                if k is None:
                    k = index[name] = len(chunks)
                    chunks.append([])
                chunks[k].append(ids[int(offs[b]):int(offs[b + 1])])
        offs = np.zeros(len(chunks) + 1, dtype=np.uint64)
        if chunks:
            offs[1:] = np.cumsum([sum(len(c) for c in ch) for ch in chunks])
        flat = [c for ch in chunks for c in ch]
        ids = np.concatenate(flat).astype(np.uint32, copy=False) if flat else np.zeros(0, dtype=np.uint32)
        out[label] = (list(index.keys()), offs, ids)
    return out


def fragsplit_tokenize_sharded(files_dir: str, mapping, tokenizer, gather: bool = True, group=None, run=None):
    """The fragsplit -> tokenizer pipeline (gtars-fragsplit/src/split.rs:36-151 feeding utils/fragments.rs:61-82) over the
    ranks of a process group: files are independent, so the folder's sorted file list is cut into ``world``
    contiguous runs balanced by file size, every rank runs the fused pipeline on its run (its own host decompress threads,
    its own GPU), and nothing crosses ranks on the data path.

    ``gather=True``: the per-cluster ``{label: (barcodes, offsets, ids)}`` of the WHOLE folder on every rank (one
    all_gather_object of the per-rank results, merged per barcode in run order) -- identical to the single-process
    ``fragsplit.fragsplit_tokenize(files_dir, mapping, tokenizer, as_arrays=True)``.  ``gather=False``: this rank's results
    only, plus a manifest ``{"rank", "world", "files": [lo, hi), "paths"}`` under the key ``"__manifest__"`` -- what a
    per-rank consumer (one output shard per GPU) wants at the config's 1e9 fragments.

    ``run(paths, mapping, tokenizer)`` is the per-rank pipeline (default: the product's ``fragsplit_tokenize_files``)."""
    import os

    from . import fragsplit

    rank, world = _rank_world(group)
    paths = fragsplit.list_fragment_files(files_dir)
    lo, hi = file_runs([os.path.getsize(p) for p in paths], world)[rank]
    if run is None:
        run = lambda ps, m, t: fragsplit.fragsplit_tokenize_files(ps, m, t, as_arrays=True)
    local = run(paths[lo:hi], mapping, tokenizer)
    if _in_process_group() and gather:
        import torch.distributed as dist

        parts = [None] * world
        dist.all_gather_object(parts, local, group=group)
        return merge_cluster_results(parts)
    if not gather:
        local = dict(local)
        local["__manifest__"] = {"rank": rank, "world": world, "files": [lo, hi], "paths": paths[lo:hi]}
    return local
