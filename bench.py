#!/usr/bin/env python3
"""bench.py -- headline benchmark of the gtars_amd hot path.

Metric (BASELINE.json): query intervals/sec tokenized vs a 100k-region hg38
universe.  Workload = BASELINE config 2: synthetic hg38-shaped BED, 100,000
non-overlapping universe regions, 1,000,000 shuffled query regions per step
(70 % near a universe region, 30 % background, 0.1 % unknown chromosome;
generators in gtars_amd/synth.py).  One "step" = one fused
tokenization pass (CSR u64 offsets + u32 token ids in Bits order) over one
batch that is already resident in HBM.

  python bench.py [--gpus N --steps K --warmup W] [--queries Q] [--universe U]

How the number is taken:
  * every step works on a DIFFERENT batch and writes DIFFERENT output buffers: the run rotates through
    --batches (default 32) device-resident batches, 0.7 GB in all, so that neither the 12 MB of queries
    nor the 10 MB of results of a step are still in the 256 MB Infinity Cache when their turn comes again;
  * a repetition = K back-to-back steps bracketed by barrier + torch.cuda.synchronize() on both sides, MAX
    over ranks; `value` is the MEDIAN over the repetitions, of which there are as many as it takes to keep
    the GPU busy for >= --min-seconds (default 0.6 s; never fewer than 11);
  * roofline.achieved = algorithmic bytes per launch / average kernel duration, the duration measured with
    HIP events on the launch stream around the K launches of the same repetitions (the events see only
    that stream; one step is exactly one k_tok_lds launch);
  * roofline.traffic is measured in the run when rocprofv3 is on PATH: two child processes -- separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of a short invocation of this script (never
    combined with a trace domain; the parent is idle meanwhile) -- give the mean bytes per k_tok_lds
    dispatch (FETCH_SIZE x2 per the gfx950 note of MI355X_MICROARCH.md).  If a pass fails the figure of the
    committed PMC summary under profiles/ is replayed instead; `traffic_source` says which.

N > 1: one rank per GPU over RCCL (`torch.distributed`, backend "nccl").  Under an external launcher
(`python -m torch.distributed.run ... bench.py --gpus N`: RANK / WORLD_SIZE set) this process IS a rank.  Started
plainly (`python bench.py --gpus N`), the parent -- before it touches any GPU -- starts the N ranks itself as a CHILD
`torch.distributed.run` (never an exec) and exits with its code; it exits non-zero when fewer than N devices are visible.
The tokenizer shards by independent query ranges with the universe index replicated, so there is no data-path collective
(weak scaling: every rank tokenizes its own Q-query batches); `with_allgather` additionally reports the rate when every
rank must end up with the whole batch's CSR (RCCL all-gatherv of offsets and ids, gtars_amd/sharding.py).  What
`north_star` shards by chromosome bucket -- the IGD database + its query batch (config 3) and the LOLA region DB with the
universe and user set (config 4) -- is measured at N > 1 in `igd_config3_sharded` / `lola_config4_sharded`: every rank
ingests only its chromosomes, counts, and ONE all-reduce of the per-file vector(s) yields the global counts
(what must be reduced: gtars-lola/src/enrichment.rs:198-221).  `ranks` lists what every rank saw (device, backend).

Rank 0 prints ONE JSON line.  At N = 1 it also carries: larger batches (`roofline_large`), a hit-heavy batch
(`roofline_hit_heavy`: 1 Mbp-wide queries, ~33 ids each; `roofline_hit_heavy_overlapping`: the same on the ChIP-like universe),
universes of 140k / 200k / 1M regions at 64M queries (`roofline_universe_*`: beyond the LDS key budget of the tokenizer kernel),
BASELINE config 3
(`igd_config3`) and config 4 on one GPU (`lola_config4`), each with a sampled CPU baseline and a parity check of the
sample against the oracle, config 5 at a reduced file count (`fragsplit_config5`), the PCIe-inclusive host-buffer rate
and the CPU baselines (the oracle on the GPU box's host cores).  Every timed output is checked in the run (`verified`).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md chip table)
PROFILE_ROUND = "r06"


def algorithmic_bytes(nq: int, h: int, nu: int) -> int:
    """12*Nq read (chrom,start,end) + 8*(Nq+1) written offsets + 4*H written ids + 12*Nu index read."""
    return 12 * nq + 8 * (nq + 1) + 4 * h + 12 * nu


def cpu_baseline(u, q, budget_s: float = 12.0):
    """The oracle (C restatement of the reference's single-threaded path) timed on this host: 1 core."""
    import oracle
    from gtars_amd import synth

    ref = oracle.Index(u["chrom"], u["start"], u["end"], None, n_chrom=synth.N_CHROM)
    qc, qs, qe = (np.ascontiguousarray(q[k]) for k in ("chrom", "start", "end"))
    nq = len(qc)
    offsets = np.zeros(nq + 1, dtype=np.uint64)
    ids = np.zeros(4 * nq, dtype=np.uint32)
    L = oracle.lib()
    L.orc_tokenize(ref._h, qc, qs, qe, nq, offsets, ids, len(ids))  # warm-up
    reps, t0 = 0, time.perf_counter()
    while True:
        L.orc_tokenize(ref._h, qc, qs, qe, nq, offsets, ids, len(ids))
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or reps >= 200:
            break
    return {
        "value": reps * nq / dt,
        "unit": "query intervals/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{reps} x {nq} queries of the same workload through oracle/gtars_oracle.c orc_tokenize "
                  f"(single thread, {dt:.1f} s); host has {os.cpu_count()} logical cores",
    }


def _cpu_max() -> str:
    try:
        return open("/sys/fs/cgroup/cpu.max").read().strip()
    except OSError:
        return "n/a"


def cpu_baseline_all_cores(u, q, budget_s: float = 4.0):
    """Courtesy multi-core figure (SURVEY section 8d ii): the oracle on every host core the
    container may use, the batch split in contiguous static chunks (one per thread, outputs land in input
    order); ctypes releases the GIL.  Reported next to, not instead of, the single-thread baseline."""
    from concurrent.futures import ThreadPoolExecutor

    import oracle
    from gtars_amd import synth

    ref = oracle.Index(u["chrom"], u["start"], u["end"], None, n_chrom=synth.N_CHROM)
    qc, qs, qe = (np.ascontiguousarray(q[k]) for k in ("chrom", "start", "end"))
    nq = len(qc)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    quota = _cpu_max().split()
    if len(quota) == 2 and quota[0].isdigit() and int(quota[1]) > 0:
        avail = min(avail, max(1, -(-int(quota[0]) // int(quota[1]))))  # container CPU quota, rounded up
    threads = max(1, min(avail, 128))
    L = oracle.lib()
    bounds = np.linspace(0, nq, threads + 1).astype(np.int64)
    chunks = [(int(bounds[t]), int(bounds[t + 1])) for t in range(threads)]
    bufs = [(np.zeros(b - a + 1, dtype=np.uint64), np.zeros(2 * (b - a) + 16, dtype=np.uint32)) for a, b in chunks]

    def work(t):
        a, b = chunks[t]
        off, ids = bufs[t]
        L.orc_tokenize(ref._h, qc[a:b], qs[a:b], qe[a:b], b - a, off, ids, len(ids))

    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(work, range(threads)))  # warm-up
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            list(ex.map(work, range(threads)))
            reps += 1
        dt = time.perf_counter() - t0
    return {"value": reps * nq / dt, "unit": "query intervals/s", "cores": threads, "kind": "port",
            "sample": f"{reps} x {nq} queries of the same workload through orc_tokenize, each batch split in {threads} "
                      f"contiguous chunks on {threads} host threads ({dt:.1f} s; CPUs usable under the affinity mask and "
                      f"cgroup quota: {avail})"}


def _dev(a, dev):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)


def verify_tokenization(u, q, offsets_t, ids_t, h, where):
    """A timed launch's output against the oracle, bit for bit (offsets u64 and ids u32 of the whole batch)."""
    import oracle
    from gtars_amd import synth

    ref = oracle.Index(u["chrom"], u["start"], u["end"], None, n_chrom=synth.N_CHROM)
    off_o, ids_o = ref.tokenize(q["chrom"], q["start"], q["end"])
    ok = (h == len(ids_o) and np.array_equal(offsets_t.cpu().numpy().view(np.uint64), off_o)
          and np.array_equal(ids_t[:h].cpu().numpy().view(np.uint32), ids_o))
    if not ok:
        raise SystemExit(f"bench.py: {where} differs from the oracle")
    return True


def bench_large(ix, u, q, nu, sizes, dev, stream, in_order=False):
    """The same kernel on larger batches: queries/s and fraction of the HBM peak, median of 5 single-launch timings each
    (HIP events on the launch stream).  The batch is the 1M-query base batch TILED on the device (query values repeat with
    period 1M; the batch stays shuffled).  Checked in the run: the first and the last period's ids equal the oracle's
    tokenization of the base batch, offsets ascend and end at the hit count.
    in_order: the tiled batch is then SORTED on the device by (chromosome, start) -- what a file-loaded RegionSet / a sorted BED
    file delivers (gtars-core/src/models/region_set.rs:182) at that size; checked: the first and the last million queries of the
    sorted batch against the oracle's tokenization of exactly those queries."""
    import torch

    import oracle
    from gtars_amd import synth

    ref = oracle.Index(u["chrom"], u["start"], u["end"], None, n_chrom=synth.N_CHROM)
    off_o, ids_o = ref.tokenize(q["chrom"], q["start"], q["end"])
    ids_o_t = torch.from_numpy(ids_o.view(np.int32)).to(dev)
    out = []
    base = {k: _dev(q[k], dev) for k in ("chrom", "start", "end")}
    nq0 = base["chrom"].numel()
    for n2 in sizes:
        rep = max(n2 // nq0, 1)
        big = {k: v.repeat(rep) for k, v in base.items()}
        n2 = nq0 * rep
        if in_order:
            key = ((big["chrom"].to(torch.int64) & 0xFFFFFFFF) << 32) | (big["start"].to(torch.int64) & 0xFFFFFFFF)
            perm = torch.argsort(key, stable=True)
            del key
            big = {k: v[perm].contiguous() for k, v in big.items()}
            del perm
        off = torch.empty(n2 + 1, dtype=torch.int64, device=dev)
        run = lambda ids, s=False: ix.tokenize_device(big["chrom"].data_ptr(), big["start"].data_ptr(), big["end"].data_ptr(),
                                                      n2, off.data_ptr(), ids.data_ptr() if ids is not None else 0,
                                                      ids.numel() if ids is not None else 0, stream, sync=s)
        h2 = run(None, True)  # offsets only: sizes the id buffer
        ids = torch.empty(h2 + 1024, dtype=torch.int32, device=dev)
        run(ids, True)
        ts = torch.cuda.current_stream()
        times = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ts)
            run(ids)
            e1.record(ts)
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) * 1e-3)
        dt = statistics.median(times)
        h0 = len(ids_o)
        ok = h2 == h0 * rep and int(off[-1]) == h2 and bool((off[1:] >= off[:-1]).all())
        if not in_order:
            ok = ok and torch.equal(ids[:h0], ids_o_t) and torch.equal(ids[h2 - h0:h2], ids_o_t)
        else:
            for lo in (0, n2 - min(n2, 1_000_000)):
                hi = lo + min(n2, 1_000_000)
                qs_ = {k: big[k][lo:hi].cpu().numpy().view(np.uint32) for k in ("chrom", "start", "end")}
                off_s, ids_s = ref.tokenize(qs_["chrom"], qs_["start"], qs_["end"])
                o_d = off[lo:hi + 1].cpu().numpy().view(np.uint64)
                ok = ok and np.array_equal(o_d - o_d[0], off_s) and np.array_equal(
                    ids[int(o_d[0]):int(o_d[-1])].cpu().numpy().view(np.uint32), ids_s)
        if not ok:
            raise SystemExit(f"bench.py: the {n2}-query launch differs from the oracle")
        byts = algorithmic_bytes(n2, h2, nu)
        out.append({"queries": n2, "hits": h2, "ms": dt * 1e3, "qps": n2 / dt, "achieved_GBps": byts / dt / 1e9,
                    "frac": byts / dt / 1e9 / HBM_PEAK_GBS, "verified": True,
                    "batch": (f"the 1M-query base batch tiled {rep}x on the device, then sorted on the device by (chromosome, start)" if in_order
                              else f"the 1M-query base batch tiled {rep}x on the device (values repeat with period {nq0}; still shuffled)")})
        del big, off, ids
        torch.cuda.empty_cache()
    return out


def _profiler_attached() -> bool:
    env = os.environ
    return ("rocprofiler" in env.get("LD_PRELOAD", "").lower() or "rocprof" in env.get("LD_PRELOAD", "").lower()
            or any(k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER_")) for k in env))


def _child_env(tmp):
    """environment of a child rocprofv3: nothing of a profiler that may be attached to THIS process is inherited"""
    env = {k: v for k, v in os.environ.items()
           if k != "LD_PRELOAD" and not k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER_"))}
    env["TMPDIR"] = tmp
    return env


def traffic_from_counter_csvs(csv_paths):
    """{"FETCH_SIZE": path, "WRITE_SIZE": path} (rocprofv3 counter_collection CSVs of two SEPARATE passes) ->
    {kernel: {"fetch": bytes summed over the dispatches of the FETCH_SIZE pass (x2: the gfx950 note of MI355X_MICROARCH.md; the
    counters are in KB), "write": bytes summed over the WRITE_SIZE pass, "fetch_dispatches": n, "write_dispatches": n,
    "fetch_per_dispatch", "write_per_dispatch", "bytes_per_dispatch"}}.
    The two passes are two runs and may launch a kernel a different number of times (a --min-seconds loop does), so EVERY
    counter's sum is divided by the dispatch count of ITS OWN pass (round 5 divided both by the FETCH_SIZE pass's count: 550
    vs 450 dispatches scaled the write bytes, and the figure drifted from run to run on an unchanged kernel)."""
    import csv

    res = {}
    for counter, path in csv_paths.items():
        key = "fetch" if counter == "FETCH_SIZE" else "write"
        scale = 2048.0 if counter == "FETCH_SIZE" else 1024.0
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("gtars::", "").strip()
                e = res.setdefault(name, {"fetch": 0.0, "write": 0.0, "fetch_dispatches": 0, "write_dispatches": 0})
                e[key] += float(row["Counter_Value"]) * scale
                e[key + "_dispatches"] += 1
    for e in res.values():
        e["fetch_per_dispatch"] = e["fetch"] / e["fetch_dispatches"] if e["fetch_dispatches"] else None
        e["write_per_dispatch"] = e["write"] / e["write_dispatches"] if e["write_dispatches"] else None
        e["bytes_per_dispatch"] = (None if e["fetch_per_dispatch"] is None or e["write_per_dispatch"] is None
                                   else e["fetch_per_dispatch"] + e["write_per_dispatch"])
    return res


def measure_traffic_with_rocprof(child_args):
    """HBM-side bytes per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE; children of this process,
    never combined with a trace domain) over a short invocation of this script -> traffic_from_counter_csvs' dict, or None.
    Not attempted when a profiler is already attached to this process (bench.py under rocprofv3 needs --no-pmc: a child
    profiler started from a process the tool library has initialised would be the forbidden exec-after-GPU-init hop)."""
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None or _profiler_attached():
        return None
    tmp = tempfile.mkdtemp(prefix="gtars_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        paths = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__)] + child_args
            r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300, env=_child_env(tmp))
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            paths[counter] = max(files, key=os.path.getmtime)
        return traffic_from_counter_csvs(paths) or None
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


IGD3 = dict(ndb=50_000_000, nq=10_000_000, n_files=1000)
# total hits of config 3 at its seeds (pairwise, binary): what one GPU returns (BENCH_r02.json, checked there against the
# oracle by tests/test_gpu_parity.py::test_config3_igd_full_size_properties) -- any sharding must reproduce them
IGD3_TOTALS = (149452392, 148309462)
IGD_PMC_CALLS = 6
IGD_COUNT_KERNELS = ("k_igd_begin", "k_igd_call_init", "k_igd_order_check", "k_igd_prep", "k_igd_route", "k_ms_", "k_split_", "k_igd_chrom_segments", "k_igd_tile_ranges", "k_igd_sweep")
LOLA4 = dict(n_sets=2000, per_set=25_000, n_universe=1_000_000, n_user=100_000)
LOLA4_SUPPORT_SUM = 1793489


def _oracle_igd(db, n_files):
    import oracle

    t = time.perf_counter()
    g = oracle.Igd()
    g.add_arrays(db["chrom"], db["start"], db["end"], np.arange(len(db["chrom"]), dtype=np.int32), db["file"])
    g.finalize()
    g.n_files = n_files
    return g, time.perf_counter() - t


def bench_igd_config3(dev, stream, ndb=IGD3["ndb"], nq=IGD3["nq"], n_files=IGD3["n_files"], cpu=True, pmc=True):
    """BASELINE config 3: 10M shuffled synthetic intervals vs a 50M-interval, 1000-file IGD database on one GPU.
    Bytes = 12*Nq + 16*Ndb + 8*F.  The headline is the batch AS SPECIFIED (shuffled); the same batch in
    (chromosome, start) order -- what a sorted BED file delivers -- is reported next to it.  `cpu_baseline`: the oracle's
    Igd (B1, one thread) on the full database and a SAMPLE of the batch; the GPU's counts of the same sample must equal it."""
    import torch

    import gtars_amd
    from gtars_amd import synth

    t = time.time()
    db = synth.make_igd_db(ndb, n_files)
    q = synth.make_background_queries(nq)
    tgen = time.time() - t
    t = time.time()
    g = gtars_amd.IgdIndex(db["chrom"], db["start"], db["end"], db["file"], n_chrom=synth.N_CHROM, n_files=n_files)
    tbuild = time.time() - t
    hits = torch.zeros(n_files, dtype=torch.int64, device=dev)
    byts = 12 * nq + 16 * ndb + 8 * n_files
    out = {"db_intervals": ndb, "queries": nq, "files": n_files, "gen_s": round(tgen, 2), "build_s": round(tbuild, 2),
           "algorithmic_bytes": byts}
    order = np.lexsort((q["start"], q["chrom"]))
    # The timed calls ROTATE through NB distinct device-resident copies of the batch (3 x 120 MB > the 256 MB Infinity Cache), like
    # the headline's 32 tokenizer batches: a call never finds its queries in the cache because the previous call left them there.
    # The shuffled copies are the same queries in three different orders (same totals), the sorted ones three buffers.
    NB = 3
    out["query_batches_rotated"] = NB
    for label, sel in (("shuffled", None), ("sorted_input", order)):
        batches = []
        for b in range(NB):
            pick = sel if sel is not None else (None if b == 0 else np.random.default_rng(100 + b).permutation(nq))
            batches.append(tuple(_dev(q[k] if pick is None else q[k][pick], dev) for k in ("chrom", "start", "end")))
        qc, qs, qe = batches[0]
        for binary in (False, True):
            turn = [0]

            def f():
                c_, s_, e_ = batches[turn[0] % NB]
                turn[0] += 1
                g.count_device(c_.data_ptr(), s_.data_ptr(), e_.data_ptr(), nq, hits.data_ptr(), 1, binary, stream)

            f()
            torch.cuda.synchronize()
            # the call is asynchronous on its stream (no host round trip inside): K calls enqueued back to back, HIP events
            # around them on the launch stream; the single-call host wall time (launch + synchronize) is reported beside it
            ts, K = torch.cuda.current_stream(), 5
            times, walls = [], []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(ts)
                for _k in range(K):
                    f()
                e1.record(ts)
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1) * 1e-3 / K)
                t0 = time.perf_counter()
                f()
                torch.cuda.synchronize()
                walls.append(time.perf_counter() - t0)
            dt = statistics.median(times)
            tot = int(hits.sum())
            key = ("binary" if binary else "pairwise") + "_" + label
            out[key] = {"ms": round(dt * 1e3, 3), "qps": round(nq / dt), "frac": round(byts / dt / 1e9 / HBM_PEAK_GBS, 5),
                        "total_hits": tot, "ms_single_call_host_wall": round(statistics.median(walls) * 1e3, 3),
                        "timing": f"median of 5 x ({K} calls enqueued back to back over {NB} rotating batches, HIP events on the launch "
                                  f"stream) / {K}"}
            if (ndb, nq, n_files) == (IGD3["ndb"], IGD3["nq"], IGD3["n_files"]):
                if tot != IGD3_TOTALS[1 if binary else 0]:
                    raise SystemExit(f"bench.py: igd_config3 {key}: {tot} hits, expected {IGD3_TOTALS[1 if binary else 0]}")
                out[key]["verified"] = "total hits == the config's known total; sample == oracle (cpu_baseline)"
        del qc, qs, qe, batches
    if cpu:
        # B1 on a sample: the oracle indexes the WHOLE database (build time stated, not part of the rate), then counts the
        # first `ns` queries of the shuffled batch; the GPU counts the same sample and must return the same vectors
        og, t_ob = _oracle_igd(db, n_files)
        ns = min(nq, 6_000_000)
        sq = {k: np.ascontiguousarray(q[k][:ns]) for k in ("chrom", "start", "end")}
        t0 = time.perf_counter()
        hp = og.count_set_overlaps(sq["chrom"], sq["start"], sq["end"], 1, n_files=n_files)
        t_p = time.perf_counter() - t0
        nb = ns // 2
        t0 = time.perf_counter()
        hb = og.count_region_hits(sq["chrom"][:nb], sq["start"][:nb], sq["end"][:nb], 1, n_files=n_files)
        t_b = time.perf_counter() - t0
        d = [_dev(sq[k], dev) for k in ("chrom", "start", "end")]
        g.count_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), ns, hits.data_ptr(), 1, False, stream)
        torch.cuda.synchronize()
        same = np.array_equal(hits.cpu().numpy().astype(np.uint64), hp.astype(np.uint64))
        g.count_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nb, hits.data_ptr(), 1, True, stream)
        torch.cuda.synchronize()
        same = same and np.array_equal(hits.cpu().numpy().astype(np.uint64), hb.astype(np.uint64))
        if not same:
            raise SystemExit("bench.py: igd_config3: the GPU's per-file counts of the sample differ from the oracle's")
        out["cpu_baseline"] = {
            "value": ns / t_p, "unit": "queries/s", "cores": 1, "kind": "port", "binary_value": nb / t_b,
            "sample": f"oracle Igd (oracle/gtars_oracle.c, one thread) over the full {ndb}-record database (built in {t_ob:.1f} s, not "
                      f"counted): count_set_overlaps of the first {ns} queries of the shuffled batch in {t_p:.1f} s, count_region_hits "
                      f"of the first {nb} in {t_b:.1f} s; the GPU's vectors for the same samples are identical",
        }
        del og, d
    del db
    if pmc:
        live = measure_traffic_with_rocprof(["--igd-pmc-child"])
        tr = None
        if live is not None:
            # every kernel of the count path (prep, partition, tile ranges, sweep), summed over the child's IGD_PMC_CALLS calls
            path = {k: v for k, v in live.items() if k.startswith(IGD_COUNT_KERNELS)}
            # (the child makes a FIXED number of calls, so both passes launch every kernel equally often: sums / calls)
            per_call = sum(v["fetch"] + v["write"] for v in path.values()) / IGD_PMC_CALLS
            if path:
                tr = {"bytes_per_call": per_call, "vs_algorithmic": per_call / byts,
                      "fetch_bytes_per_call": sum(v["fetch"] for v in path.values()) / IGD_PMC_CALLS,
                      "write_bytes_per_call": sum(v["write"] for v in path.values()) / IGD_PMC_CALLS,
                      "dispatch_counts_agree": all(v["fetch_dispatches"] == v["write_dispatches"] for v in path.values()),
                      "by_kernel_per_call": {k: round((v["fetch"] + v["write"]) / IGD_PMC_CALLS) for k, v in sorted(path.items())},
                      "source": f"measured in this run: child rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py "
                                f"--igd-pmc-child` ({IGD_PMC_CALLS} shuffled-batch calls, pairwise and binary alternating; FETCH_SIZE x2)"}
        if tr is None:
            tpath = os.path.join(ROOT, "profiles", PROFILE_ROUND, "traffic_igd_config3.json")
            if os.path.exists(tpath):
                tr = json.load(open(tpath))
                tr["source"] = f"replayed from profiles/{PROFILE_ROUND}/traffic_igd_config3.json"
        out["traffic"] = tr
    del g
    torch.cuda.empty_cache()
    return out


def igd_pmc_child():
    """`bench.py --igd-pmc-child` (under rocprofv3 --pmc): a few config-3 calls, nothing else."""
    import torch

    import gtars_amd
    from gtars_amd import synth

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    db = synth.make_igd_db(IGD3["ndb"], IGD3["n_files"])
    q = synth.make_background_queries(IGD3["nq"])
    g = gtars_amd.IgdIndex(db["chrom"], db["start"], db["end"], db["file"], n_chrom=synth.N_CHROM, n_files=IGD3["n_files"])
    del db
    qc, qs, qe = (_dev(q[k], dev) for k in ("chrom", "start", "end"))
    hits = torch.zeros(IGD3["n_files"], dtype=torch.int64, device=dev)
    g.count_device(qc.data_ptr(), qs.data_ptr(), qe.data_ptr(), 1000, hits.data_ptr(), 1, True, stream)  # builds pme_file (per-query path)
    torch.cuda.synchronize()
    for i in range(IGD_PMC_CALLS):
        g.count_device(qc.data_ptr(), qs.data_ptr(), qe.data_ptr(), IGD3["nq"], hits.data_ptr(), 1, bool(i & 1), stream)
    torch.cuda.synchronize()


def bench_igd_broad_peaks(dev, stream, ndb=IGD3["ndb"], nq=IGD3["nq"], n_files=IGD3["n_files"], frac=0.01, wmax=100_000):
    """Config 3 with LONG records in the database (not a BASELINE config: a robustness figure): `frac` of the records get a
    width of U[5000, wmax) instead of 200 + U[0, 800) -- broad peaks.  The flat record layout degrades by 30x on this database
    (one long record inflates the prefix maxima and the ownership of everything behind it); the library counts it through its
    pieces view.  Timed: the min_overlap == 1 batch counts; verified: identical per-file vectors from an index
    built WITHOUT the pieces view (one call each)."""
    import torch

    import gtars_amd
    from gtars_amd import synth

    db = synth.make_igd_db(ndb, n_files)
    rng = np.random.default_rng(1)
    wide = rng.random(ndb) < frac
    db["end"] = np.where(wide, db["start"].astype(np.int64) + rng.integers(5_000, wmax, ndb), db["end"]).astype(db["end"].dtype)
    q = synth.make_background_queries(nq)
    qc, qs, qe = (_dev(q[k], dev) for k in ("chrom", "start", "end"))
    hits = torch.zeros(n_files, dtype=torch.int64, device=dev)
    out = {"db_intervals": ndb, "queries": nq, "files": n_files, "long_fraction": frac, "long_width": [5000, wmax]}
    vecs = {}
    for label, env in (("pieces_view", None), ("flat_layout", "1")):
        if env:
            os.environ["GTARS_IGD_NO_PIECES"] = env
        else:
            os.environ.pop("GTARS_IGD_NO_PIECES", None)
        gtars_amd.reload_env()  # (the library snapshots its switches at first use)
        try:
            t = time.time()
            g = gtars_amd.IgdIndex(db["chrom"], db["start"], db["end"], db["file"], n_chrom=synth.N_CHROM, n_files=n_files)
            o = {"build_s": round(time.time() - t, 2)}
        finally:
            os.environ.pop("GTARS_IGD_NO_PIECES", None)
            gtars_amd.reload_env()
        for binary in (False, True):
            f = lambda: g.count_device(qc.data_ptr(), qs.data_ptr(), qe.data_ptr(), nq, hits.data_ptr(), 1, binary, stream)
            f()
            torch.cuda.synchronize()
            reps = 3 if env is None else 1
            t0 = time.perf_counter()
            for _ in range(reps):
                f()
            torch.cuda.synchronize()
            key = "binary" if binary else "pairwise"
            o[key + "_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 3)
            h = hits.cpu().numpy().copy()
            if env is None:
                vecs[key] = h
                o[key + "_hits"] = int(h.sum())
            elif not np.array_equal(h, vecs[key]):
                raise SystemExit(f"bench.py: igd_broad_peaks: {key} vectors of the pieces view and the flat layout differ")
        out[label] = o
        del g
        torch.cuda.empty_cache()
    out["verified"] = "per-file vectors identical to the flat layout's (one call each)"
    return out


def bench_lola_config4(dev, stream, n_sets=LOLA4["n_sets"], per_set=LOLA4["per_set"], n_universe=LOLA4["n_universe"],
                       n_user=LOLA4["n_user"], cpu=True):
    """BASELINE config 4 on ONE GPU: support counts of one user set and of the universe against a 2000-set region DB
    (two binary IGD counts) + the contingency cells (enrichment.rs:198-221)."""
    import torch

    import gtars_amd
    from gtars_amd import synth
    from gtars_amd._lib import check, lib

    db = synth.make_igd_db(n_sets * per_set, n_sets, seed=6)
    uni = synth.make_universe(n_universe, seed=3)
    rng = np.random.default_rng(9)
    sel = np.sort(rng.choice(len(uni["chrom"]), n_user, replace=False))
    t = time.time()
    g = gtars_amd.IgdIndex(db["chrom"], db["start"], db["end"], db["file"], n_chrom=synth.N_CHROM, n_files=n_sets)
    tb = time.time() - t
    # the universe and the user set as ONE resident batch of two sets: both support vectors come out of one pass over the
    # region DB (gtars_igd_count_sets_device; the reference walks the database once per set, enrichment.rs:198-215)
    both = [_dev(np.concatenate([uni[k], uni[k][sel]]), dev) for k in ("chrom", "start", "end")]
    nuni = len(uni["chrom"])
    set_off = [0, nuni, nuni + n_user]
    support = torch.zeros(2, n_sets, dtype=torch.int64, device=dev)
    uh, sh = support[0], support[1]
    cells = [torch.empty(n_sets, dtype=torch.int64, device=dev) for _ in range(4)]

    def run():
        g.count_sets_device(both[0].data_ptr(), both[1].data_ptr(), both[2].data_ptr(), set_off, support.data_ptr(), 1, True, stream)
        check(lib.gtars_lola_contingency_device(sh.data_ptr(), uh.data_ptr(), n_sets, n_user, nuni,
                                                *[c.data_ptr() for c in cells], stream))

    def run_set_by_set():  # what the shared pass replaces: one count per set (round 2's form of this object)
        g.count_device(both[0].data_ptr(), both[1].data_ptr(), both[2].data_ptr(), nuni, uh.data_ptr(), 1, True, stream)
        g.count_device(both[0][nuni:].data_ptr(), both[1][nuni:].data_ptr(), both[2][nuni:].data_ptr(), n_user, sh.data_ptr(), 1,
                       True, stream)
        check(lib.gtars_lola_contingency_device(sh.data_ptr(), uh.data_ptr(), n_sets, n_user, nuni,
                                                *[c.data_ptr() for c in cells], stream))

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        times = []
        for _ in range(7):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        return statistics.median(times)

    dt_sets = timed(run_set_by_set)
    by_set = support.cpu().numpy().copy()
    dt = timed(run)
    if not np.array_equal(by_set, support.cpu().numpy()):
        raise SystemExit("bench.py: lola_config4: the shared pass and the set-by-set counts differ")
    a, b, c, d = [x.cpu().numpy() for x in cells]
    uh_h, sh_h = uh.cpu().numpy(), sh.cpu().numpy()
    ok = bool(((a + b) == uh_h).all() and ((a + c) == n_user).all() and ((a + b + c + d) == nuni).all())
    byts = 12 * (nuni + n_user) + 16 * n_sets * per_set + 8 * 2 * n_sets
    out = {"sets": n_sets, "db_intervals": n_sets * per_set, "universe": nuni, "user": n_user, "build_s": round(tb, 2),
           "counts_ms": round(dt * 1e3, 3), "algorithmic_bytes": byts, "frac": round(byts / dt / 1e9 / HBM_PEAK_GBS, 5),
           "counts_ms_set_by_set": round(dt_sets * 1e3, 3),
           "how": "universe + user set as two sets of one batch: ONE pass over the region DB (gtars_igd_count_sets_device); "
                  "counts_ms_set_by_set = one count per set (same vectors, checked)",
           "identities_hold": ok, "support_sum": int(a.sum())}
    if not ok:
        raise SystemExit("bench.py: lola_config4: the contingency identities do not hold")
    out.update(_lola_end_to_end(db, uni, sel, n_sets, per_set, a, b, c, d))
    if cpu:
        # B1: the oracle's Igd over the whole region DB; the user set's support vector in full (the GPU's must equal it) and
        # the first `ns` universe regions as the timing sample of the pooled-support count
        og, t_ob = _oracle_igd(db, n_sets)
        user = {k: np.ascontiguousarray(uni[k][sel]) for k in ("chrom", "start", "end")}
        t0 = time.perf_counter()
        su = og.count_region_hits(user["chrom"], user["start"], user["end"], 1, n_files=n_sets)
        ns = nuni
        og.count_region_hits(uni["chrom"][:ns], uni["start"][:ns], uni["end"][:ns], 1, n_files=n_sets)
        t_c = time.perf_counter() - t0
        if not np.array_equal(su.astype(np.int64), sh_h):
            raise SystemExit("bench.py: lola_config4: the user set's support vector differs from the oracle's")
        out["verified"] = "a,b,c,d identities; the user set's support vector == oracle"
        out["cpu_baseline"] = {
            "value": (n_user + ns) / t_c, "unit": "regions/s (binary support counts)", "cores": 1, "kind": "port",
            "est_counts_ms": round((n_user + nuni) / ((n_user + ns) / t_c) * 1e3, 1),
            "sample": f"oracle Igd over the full {n_sets * per_set}-record region DB (built in {t_ob:.1f} s, not counted): "
                      f"count_region_hits of the whole {n_user}-region user set + the first {ns} universe regions in {t_c:.1f} s "
                      f"(one thread); est_counts_ms extrapolates to user set + universe",
        }
        del og
    del db, g
    torch.cuda.empty_cache()
    return out


def _lola_end_to_end(db, uni, sel, n_sets, per_set, a, b, c, d):
    """The drop-in call itself, `gtars.lola.run_lola(user_sets, universe, region_db)` (gtars-python/src/lola/mod.rs:180-271), timed
    end to end on config 4: host region sets in, the reference's column dict out -- encode + copy of the 1.1M query regions,
    the shared count pass, the contingency cells, the statistics tail for the 2000 tables (Fisher p-values, conditional-MLE
    odds ratios, ranks, global order, BH q-values: gtars_lola_stats, compiled and threaded; round 5: a Python loop, ~15 s) and
    the column layout.  `statistics_ms` is the tail alone on the same cells."""
    import ctypes as C

    from gtars_amd import lola, synth
    from gtars_amd._lib import check, cstr_array, lib, ptr
    from gtars_amd.igd import Igd
    from gtars_amd.models import RegionSet

    names = list(synth.CHROM_NAMES)
    narr, _k1 = cstr_array(names)
    fnames = [f"set{i:04d}.bed" for i in range(n_sets)]
    farr, _k2 = cstr_array(fnames)
    cnt = np.bincount(db["file"], minlength=n_sets).astype(np.uint32)
    avgw = np.bincount(db["file"], weights=(db["end"].astype(np.int64) - db["start"].astype(np.int64)), minlength=n_sets) / np.maximum(cnt, 1)
    h = C.c_void_p()
    t0 = time.perf_counter()
    check(lib.gtars_igddb_from_arrays(C.cast(narr, C.c_void_p), len(names), ptr(np.ascontiguousarray(db["chrom"], dtype=np.uint32)),
                                      ptr(np.ascontiguousarray(db["start"]).view(np.int32)), ptr(np.ascontiguousarray(db["end"]).view(np.int32)),
                                      ptr(np.zeros(len(db["chrom"]), np.int32)), ptr(np.ascontiguousarray(db["file"], dtype=np.uint32)),
                                      len(db["chrom"]), C.cast(farr, C.c_void_p), ptr(cnt), ptr(avgw.astype(np.float64)), n_sets, C.byref(h)))
    igd = Igd._from_db(h)
    t_db = time.perf_counter() - t0

    class _Sized:  # a database set as run_lola's `size` column sees it
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

    rdb = lola.RegionDB(igd, [_Sized(int(x)) for x in cnt], [lola.RegionDB._anno(f, collection="synthetic") for f in fnames])
    cn = np.asarray(names, dtype=object)
    uni_rs = RegionSet.from_vectors(cn[uni["chrom"]].tolist(), uni["start"], uni["end"])
    user_rs = RegionSet.from_vectors(cn[uni["chrom"][sel]].tolist(), uni["start"][sel], uni["end"][sel])
    res = lola.run_lola([user_rs], uni_rs, rdb)  # warm-up (workspaces, host threads)
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        res = lola.run_lola([user_rs], uni_rs, rdb)
        times.append(time.perf_counter() - t0)
    st_times = []
    for _ in range(5):
        t0 = time.perf_counter()
        st = lola.lola_stats(a, b, c, d, True)
        st_times.append(time.perf_counter() - t0)
    # the drop-in's rows against the cells of the device-resident run above
    dbs = np.asarray(res["dbSet"])
    ok = (len(dbs) == n_sets and np.array_equal(np.asarray(res["support"]), a[dbs]) and np.array_equal(np.asarray(res["b"]), b[dbs])
          and np.array_equal(np.asarray(res["d"]), d[dbs]) and np.array_equal(st["order"].astype(np.int64), dbs)
          and np.array_equal(np.asarray(res["pValueLog"]), st["pValueLog"][0][dbs])
          and bool((np.diff(np.asarray(res["pValueLog"])) <= 0).all()))
    if not ok:
        raise SystemExit("bench.py: lola_config4: run_lola's rows differ from the device-resident cells / statistics")
    del rdb, igd
    return {"run_lola_end_to_end_ms": round(statistics.median(times) * 1e3, 2), "run_lola_calls_ms": [round(t * 1e3, 2) for t in times],
            "statistics_ms": round(statistics.median(st_times) * 1e3, 2), "statistics_host_threads": int(lib.gtars_host_threads(64)),
            "run_lola_how": "gtars.lola.run_lola([user RegionSet], universe RegionSet, RegionDB of the 2000 sets), host objects in, "
                            "column dict out: encode + H2D of 1.1M regions, one shared count pass, cells, gtars_lola_stats (2000 tables), "
                            f"columns; rows checked against the device-resident cells; RegionDB handle built in {t_db:.2f} s (not counted)"}


def _usable_host_threads() -> int:
    """hardware threads this process may use: affinity mask and the container's CPU quota (what the library's host pipelines
    start at most -- csrc/host.cpp, host_thread_budget)"""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    quota = _cpu_max().split()
    if len(quota) == 2 and quota[0].isdigit() and int(quota[1]) > 0:
        avail = min(avail, max(1, -(-int(quota[0]) // int(quota[1]))))
    return avail


def bench_fragsplit_config5(files=48, frags=100_000, clusters=20, cpu_files=48):
    """BASELINE config 5 at the config's PER-FILE size (1e5 fragments, 500 barcodes: SURVEY 8d C5) and a reduced file count (the
    config names 10,000 files on 8 GPUs; files are independent, so per-file cost is what scales): gzip'd fragment files ->
    barcode routing -> per-cluster tokenization, end to end from the .gz files."""
    import shutil
    import tempfile

    import oracle
    from gtars_amd import synth, utils
    from gtars_amd.fragsplit import BarcodeToClusterMap, fragsplit_tokenize, pseudobulk_fragment_files
    from gtars_amd.tokenizers import Tokenizer, tokenize_fragment_files

    tmp = tempfile.mkdtemp(prefix="gtars_c5_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        u = synth.make_universe(100_000)
        t = time.time()
        ub, fd, mp, gz_bytes = synth.write_config5_inputs(tmp, u, files, frags, clusters)
        t_gen = time.time() - t
        tok = Tokenizer.from_bed(ub)
        m = BarcodeToClusterMap.from_file(mp)
        paths = sorted(os.path.join(fd, f) for f in os.listdir(fd))
        t = time.perf_counter()
        n_parsed = 0
        for pth in paths:  # host share: gunzip + parse only, file by file (the pipelines below read the files on several host threads)
            n_parsed += len(utils.read_fragments(pth)["start"])
        t_parse = time.perf_counter() - t
        od = os.path.join(tmp, "out")
        t = time.perf_counter(); st = pseudobulk_fragment_files(fd, m, od); t_split = time.perf_counter() - t
        cluster_files = [os.path.join(od, f"cluster_{l}.bed.gz") for l in m.cluster_labels()]
        t = time.perf_counter(); res2 = tokenize_fragment_files(cluster_files, tok, workers=16); t_tok = time.perf_counter() - t
        fragsplit_tokenize(fd, m, tok, as_arrays=True)  # warm-up (device buffers)
        t_runs = []
        fused = None
        for _ in range(7):
            fused = None  # (the previous call's result is released outside the timed region: its arrays are views of C memory)
            t = time.perf_counter(); fused = fragsplit_tokenize(fd, m, tok, as_arrays=True); t_runs.append(time.perf_counter() - t)
        t_fused = statistics.median(t_runs)
        # where the last call's time went (the library's own stage clock: gtars_fragsplit_last_stages)
        import ctypes as C_

        from gtars_amd import _lib as L_
        sg = (C_.c_double * 12)()
        L_.lib.gtars_fragsplit_last_stages(C_.cast(sg, C_.c_void_p))
        stages = {"text_parsed_on": "device (fragparse.hip: line split, fields, barcode and chromosome lookup, tokenization, ids regrouped by (file, barcode), gzip CRC-32)" if sg[0]
                  else "host threads", "waves": int(sg[1]),
                  "host_read_inflate_s" if sg[0] else "host_read_inflate_parse_route_s": round(sg[2], 4),
                  "device_waves_s_summed_over_the_device_threads" if sg[0] else "tokenizer_calls_s": round(sg[4], 4),
                  "of_which_behind_the_last_wave_s": round(sg[5], 4), "regroup_by_barcode_host_share_s" if sg[0] else "regroup_by_barcode_s": round(sg[6], 4)}
        if sg[0]:
            stages.update({"device_text_in_s": round(sg[7], 4), "device_split_parse_sort_s": round(sg[8], 4), "device_gather_s": round(sg[9], 4),
                           "device_tokenize_s": round(sg[10], 4), "device_regroup_and_results_out_s": round(sg[11], 4),
                           "note": "the host threads only read and inflate (csrc/inflate_fast.h, into pinned memory); everything else of split.rs:84-131 / fragments.rs:12-56 runs on "
                                   "the GPU (two device threads, a stream each) while the next files inflate; the call's floor is the inflate time on the box's usable host threads"})
        else:
            stages["host_per_cluster_append_s"] = round(sg[3], 4)
        ids_two = sum(sum(len(v) for v in d.values()) for d in res2)
        ids_fused = sum(int(v[1][-1]) for v in fused.values())
        n = files * frags
        out = {"files": files, "fragments_per_file": frags, "fragments": n, "clusters": m.n_clusters(),
               "input_gz_MB": round(gz_bytes / 1e6, 1), "gen_s": round(t_gen, 1), "host_threads": _usable_host_threads(), "host_logical_cores": os.cpu_count(),
               "scale_note": f"{files} files x {frags} fragments = {n} of the config's 1e9 fragments (1/{round(1e9 / n)}); one GPU",
               "routed_fragments": st["written"], "token_ids": ids_fused,
               "host_gunzip_parse": {"s": round(t_parse, 3), "fragments_per_s": round(n_parsed / t_parse), "note": "gtars_fragments_read, one file at a time"},
               "two_step": {"fragsplit_s": round(t_split, 3), "tokenize_cluster_files_s": round(t_tok, 3),
                            "fragments_per_s": round(n / (t_split + t_tok))},
               "fused": {"s": round(t_fused, 4), "fragments_per_s": round(n / t_fused), "runs_s": [round(x, 4) for x in t_runs],
                         "timing": "median of 7 calls from Python, result conversion included", "stages_of_the_last_call": stages},
               "value": n / t_fused, "unit": "fragments/s end to end (fused route + tokenize)"}
        if ids_two != ids_fused:
            raise SystemExit("bench.py: fragsplit_config5: fused and two-step pipelines disagree")
        # B1 on a sample of the files: the oracle's compiled restatement (fragsplit_oracle.c: split.rs:36-151 routing +
        # fragments.rs:61-82 per-line tokenization), one thread, from the same .gz files; its per-cluster id counts are held
        # against the device pipeline's over the same files
        sample = paths[:cpu_files]
        om, otok = oracle.OracleBarcodeMap(mp), oracle.OracleTokenizer(ub)
        t = time.perf_counter()
        cpu = oracle.fragsplit_tokenize_compiled(sample, om, otok)
        t_cpu = time.perf_counter() - t
        from gtars_amd.fragsplit import fragsplit_tokenize_files
        dev = fragsplit_tokenize_files(sample, m, tok, as_arrays=True)
        if {k: int(v[1][-1]) for k, v in dev.items()} != {k: v[0] for k, v in cpu.items()}:
            raise SystemExit("bench.py: fragsplit_config5: device pipeline and the compiled CPU restatement disagree")
        out["cpu_baseline"] = {"value": len(sample) * frags / t_cpu, "unit": "fragments/s", "cores": 1, "kind": "port",
                               "sample": f"{len(sample)} of the {files} files ({len(sample) * frags} fragments) through oracle/fragsplit_oracle.c "
                                         f"(compiled C restatement of split.rs + fragments.rs: gunzip, parse, route, one "
                                         f"tokenize per line; one thread, {t_cpu:.2f} s); per-cluster id counts equal the device's"}
        out["verified"] = "fused == two-step id counts (every file is compared with the oracle by tests/test_gpu_host.py)"
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def bench_fragsplit_many_files(files=1000, frags=10_000, clusters=20, cpu_files=48):
    """Config 5 towards the config's FILE COUNT (10,000 files; `fragsplit_config5` times 48 files of the config's per-file size): a
    folder of `files` small fragment files through the fused fragsplit -> tokenizer pipeline -- many gzip members, many barcode
    tables, many batches.  The first call (cold: pinned pool, workspaces, device tables) is reported apart; `value` is the median of
    the six calls behind it."""
    import shutil
    import tempfile

    import oracle
    from gtars_amd import synth
    from gtars_amd.fragsplit import BarcodeToClusterMap, fragsplit_tokenize, fragsplit_tokenize_files
    from gtars_amd.tokenizers import Tokenizer

    tmp = tempfile.mkdtemp(prefix="gtars_c5m_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        u = synth.make_universe(100_000)
        t = time.time()
        ub, fd, mp, gz_bytes = synth.write_config5_inputs(tmp, u, files, frags, clusters)
        t_gen = time.time() - t
        tok = Tokenizer.from_bed(ub)
        m = BarcodeToClusterMap.from_file(mp)
        runs = []
        fused = None
        for _ in range(7):
            fused = None
            t = time.perf_counter(); fused = fragsplit_tokenize(fd, m, tok, as_arrays=True); runs.append(time.perf_counter() - t)
        n = files * frags
        steady = statistics.median(runs[1:])
        import ctypes as C_

        from gtars_amd import _lib as L_
        sg = (C_.c_double * 12)()
        L_.lib.gtars_fragsplit_last_stages(C_.cast(sg, C_.c_void_p))
        out = {"files": files, "fragments_per_file": frags, "fragments": n, "clusters": m.n_clusters(), "input_gz_MB": round(gz_bytes / 1e6, 1),
               "gen_s": round(t_gen, 1), "host_threads": _usable_host_threads(),
               "first_call_s": round(runs[0], 4), "steady_calls_s": [round(x, 4) for x in runs[1:]], "s": round(steady, 4),
               "value": n / steady, "unit": "fragments/s end to end (fused route + tokenize), median of the 6 calls behind the first",
               "token_ids": sum(int(v[1][-1]) for v in fused.values()),
               "stages_of_the_last_call": {"batches": int(sg[1]), "host_read_inflate_s": round(sg[2], 4),
                                           "device_batches_s_summed_over_the_device_threads": round(sg[4], 4),
                                           "of_which_behind_the_last_batch_s": round(sg[5], 4), "regroup_host_share_s": round(sg[6], 4),
                                           "device_text_in_s": round(sg[7], 4), "device_split_parse_sort_s": round(sg[8], 4),
                                           "device_tokenize_s": round(sg[10], 4), "device_regroup_and_results_out_s": round(sg[11], 4)},
               "scale_note": f"{files} files x {frags} fragments = {n} of the config's 1e9 fragments in 10,000 files (a tenth of the files, a "
                             "tenth of the per-file size)"}
        paths = sorted(os.path.join(fd, f) for f in os.listdir(fd))[:cpu_files]
        cpu = oracle.fragsplit_tokenize_compiled(paths, oracle.OracleBarcodeMap(mp), oracle.OracleTokenizer(ub))
        dev = fragsplit_tokenize_files(paths, m, tok, as_arrays=True)
        if {k: int(v[1][-1]) for k, v in dev.items()} != {k: v[0] for k, v in cpu.items()}:
            raise SystemExit("bench.py: fragsplit_config5_1000: device pipeline and the compiled CPU restatement disagree")
        out["verified"] = f"per-cluster id counts of the first {len(paths)} files == the compiled CPU restatement's"
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ------------------------------------------------------------------------------------------------ N > 1

def _lib_host_threads() -> int:
    from gtars_amd import _lib

    return _lib.lib.gtars_host_threads(64)


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` started plainly: start the N ranks as a child torch.distributed.run (this process has
    not touched a GPU: torch.cuda.device_count() does not initialise one on this image) and return its exit code."""
    import socket
    import subprocess

    import torch

    backend = os.environ.get("GTARS_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < 1 or (backend == "nccl" and ndev < n):
        print(f"bench.py: --gpus {n} needs {n} visible devices under the nccl backend, found {ndev} "
              f"(GTARS_BENCH_BACKEND=gloo lets ranks share a device: a plumbing test, not a measurement)", file=sys.stderr)
        return 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, GTARS_BENCH_SPAWNED="1")).returncode


STRONG_NOTE = ("strong scaling of a 0.2-0.4 ms single-GPU call: at 8 ranks a rank's share is ~50 us of kernels, so the host-wall figure is "
               "barrier + all-reduce latency; read per_rank_ms (device time by HIP events, collective time apart) and the *_weak objects "
               "for what the GPUs do")


def timed_sharded_calls(dist, world, barrier, max_over_ranks, call, reps=7):
    """`call(timing_dict)` once untimed, then `reps` times between barriers: -> (median host wall seconds incl. the collective, max
    over ranks; per rank: median device ms of its kernels by HIP events and median ms its stream then spends in the collective)"""
    call(None)
    times, dev_ms, coll_ms = [], [], []
    for _ in range(reps):
        tm = {}
        barrier()
        t0 = time.perf_counter()
        call(tm)
        barrier()
        times.append(max_over_ranks(time.perf_counter() - t0))
        dev_ms.append(tm.get("device_ms", 0.0))
        coll_ms.append(tm.get("collective_ms", 0.0))
    mine = {"device_ms": round(statistics.median(dev_ms), 4), "collective_ms": round(statistics.median(coll_ms), 4)}
    per_rank = [None] * world
    dist.all_gather_object(per_rank, mine)
    return statistics.median(times), per_rank


def bench_igd_weak(dist, backend, dev, rank, world, max_over_ranks, barrier, scale=1):
    """Config 3 weak-scaled: the GENOME grows with the ranks.  Rank r owns chromosomes 25 r .. 25 r + 24 of a (25 x world)-chromosome
    genome -- the chromosome-bucket sharding of north_star with buckets of equal weight -- with config 3's 5e7 database records and
    1e7 queries of its own on them (rank r's seeds: 6 + 2 r / 7 + 2 r, so rank 0 holds exactly the single-GPU config), and one
    all-reduce of the F counters per call, as in the strong-scaled object.  Every rank generates only its own share."""
    import torch

    from gtars_amd import sharding, synth

    eng = sharding.HipEngine(dev)
    ndb, nq, F = IGD3["ndb"] // scale, IGD3["nq"] // scale, IGD3["n_files"]
    db = synth.make_igd_db(ndb, F, seed=6 + 2 * rank)
    q = synth.make_background_queries(nq, seed=7 + 2 * rank)
    sdb = sharding.ShardedIgd(eng, db, synth.N_CHROM, F, mode="range")  # (local ids 0..24: the rank's own chromosomes, whole)
    del db
    h = eng.upload(q["chrom"], q["start"], q["end"])  # its own queries, all of them (range mode would cut a shared batch)
    byts = (12 * nq + 16 * ndb + 8 * F) * world
    o = {"db_intervals": ndb * world, "queries": nq * world, "files": F, "algorithmic_bytes": byts, "scaling": "weak",
         "per_rank": {"db_intervals": ndb, "queries": nq},
         "sharding": "chromosome buckets of a genome that grows with the ranks: rank r owns chromosomes 25r..25r+24, its own 5e7 records and "
                     "1e7 queries on them (the single-GPU config per rank)",
         "collective": f"one all-reduce(SUM) of {F} int64 per call ({backend})",
         "timing": "host wall time of one call incl. the all-reduce and the barriers around it, max over ranks, median of 7; per_rank_ms: "
                   "device time of the rank's kernels by HIP events, and the collective apart"}
    for binary in (False, True):
        hits = torch.zeros(F, dtype=torch.int64, device=dev)
        local = [None]

        def call(tm):
            sdb.count_resident(h, 1, binary, hits, timing=tm)

        # this rank's own total (no collective), to check the reduced vector against the sum of the ranks' own
        local_hits = eng.igd_count_resident(sdb.g, h, 1, binary)
        mine = int(local_hits.sum())
        dt, per_rank = timed_sharded_calls(dist, world, barrier, max_over_ranks, call)
        totals = [None] * world
        dist.all_gather_object(totals, mine)
        tot = int(hits.sum())
        if tot != sum(totals) or (scale == 1 and totals[0] != IGD3_TOTALS[1 if binary else 0]):
            raise SystemExit(f"bench.py: igd_config3_weak: reduced total {tot}, ranks' own {totals}, single-GPU {IGD3_TOTALS}")
        o["binary" if binary else "pairwise"] = {"ms": round(dt * 1e3, 3), "qps": round(nq * world / dt), "total_hits": tot,
                                                  "frac_of_all_gpus": round(byts / dt / 1e9 / HBM_PEAK_GBS / world, 5),
                                                  "per_rank_ms": per_rank,
                                                  "verified": "reduced total == sum of the ranks' own totals; rank 0's == the single-GPU total"}
    del sdb, h, q
    torch.cuda.empty_cache()
    return o


def bench_sharded(dist, backend, dev, rank, world, max_over_ranks, barrier, scale=1):
    """What north_star shards by chromosome bucket, at N > 1: config 3 (IGD database + query batch) and config 4 (LOLA region
    DB, universe and user set).  Strong scaling: the configs' sizes are fixed, every rank ingests and counts only its
    chromosomes (LPT buckets over database + query weights), one all-reduce of the per-file vector(s) per call."""
    import torch

    from gtars_amd import sharding, synth

    eng = sharding.HipEngine(dev)
    out = {}
    # ---- config 3 ----
    ndb, nq, F = IGD3["ndb"] // scale, IGD3["nq"] // scale, IGD3["n_files"]
    q = synth.make_background_queries(nq)
    t = time.time()
    sdb = sharding.ShardedIgd(eng, synth.igd_db_chunks(ndb, F), synth.N_CHROM, F, mode="bucket", balance_with=[q["chrom"]])
    t_ingest = time.time() - t
    h = sdb.upload_local(q)
    local_q = int(h[0].numel())
    byts = 12 * nq + 16 * ndb + 8 * F
    o3 = {"db_intervals": ndb, "queries": nq, "files": F, "algorithmic_bytes": byts, "scaling": "strong",
          "sharding": "chromosome buckets (LPT over database + query weights); every rank ingests only its chromosomes' rows",
          "collective": f"one all-reduce(SUM) of {F} int64 per call ({backend})",
          "timing": "host wall time of one call incl. the all-reduce and the barriers around it, max over ranks, median of 7",
          "ingest_s_rank0": round(t_ingest, 2)}
    locals_ = [None] * world
    dist.all_gather_object(locals_, {"db_intervals": sdb.local_intervals, "queries": local_q})
    o3["per_rank"] = locals_
    for binary in (False, True):
        hits = torch.zeros(F, dtype=torch.int64, device=dev)
        dt, per_rank = timed_sharded_calls(dist, world, barrier, max_over_ranks, lambda tm: sdb.count_resident(h, 1, binary, hits, timing=tm))
        tot = int(hits.sum())
        if scale == 1 and tot != IGD3_TOTALS[1 if binary else 0]:
            raise SystemExit(f"bench.py: igd_config3_sharded: {tot} hits, expected {IGD3_TOTALS[1 if binary else 0]}")
        o3["binary" if binary else "pairwise"] = {"ms": round(dt * 1e3, 3), "qps": round(nq / dt), "total_hits": tot,
                                                  "frac_of_all_gpus": round(byts / dt / 1e9 / HBM_PEAK_GBS / world, 5),
                                                  "per_rank_ms": per_rank,
                                                  "verified": "total hits == the single-GPU total" if scale == 1 else None}
    o3["note"] = STRONG_NOTE
    out["igd_config3_sharded"] = o3
    del sdb, h, q
    torch.cuda.empty_cache()
    # ---- config 3, weak: the genome, the database and the batch grow with the ranks ----
    out["igd_config3_weak"] = bench_igd_weak(dist, backend, dev, rank, world, max_over_ranks, barrier, scale)
    # ---- config 4 ----
    n_sets, per_set = LOLA4["n_sets"], LOLA4["per_set"] // scale
    uni = synth.make_universe(LOLA4["n_universe"] // scale, seed=3)
    sel = np.sort(np.random.default_rng(9).choice(len(uni["chrom"]), LOLA4["n_user"] // scale, replace=False))
    user = {k: uni[k][sel] for k in ("chrom", "start", "end")}
    nuni, n_user = len(uni["chrom"]), len(sel)
    sdb = sharding.ShardedIgd(eng, synth.igd_db_chunks(n_sets * per_set, n_sets, seed=6), synth.N_CHROM, n_sets, mode="bucket",
                              balance_with=[uni["chrom"]])
    hboth = sdb.upload_local_sets([uni, user])  # this rank's share of both sets, one resident batch
    stacked = torch.zeros(2, n_sets, dtype=torch.int64, device=dev)

    cells_box = [None]

    def run(tm=None):
        sdb.count_sets_resident(hboth, 1, True, stacked, timing=tm)  # one pass over the local region DB + the one all-reduce
        cells_box[0] = sharding.contingency(stacked[1:], stacked[0], [n_user], nuni)

    dt, lola_per_rank = timed_sharded_calls(dist, world, barrier, max_over_ranks, run)
    cells = cells_box[0]
    a, b, c, d = [x[0].cpu().numpy() for x in cells]
    uh = stacked[0].cpu().numpy()
    ok = bool(((a + b) == uh).all() and ((a + c) == n_user).all() and ((a + b + c + d) == nuni).all()
              and (scale != 1 or int(a.sum()) == LOLA4_SUPPORT_SUM))
    if not ok:
        raise SystemExit("bench.py: lola_config4_sharded: identities / support sum differ from the single-GPU result")
    out["lola_config4_sharded"] = {"sets": n_sets, "db_intervals": n_sets * per_set, "universe": nuni, "user": n_user,
                                   "scaling": "strong", "counts_ms": round(dt * 1e3, 3), "support_sum": int(a.sum()),
                                   "timing": "host wall time of one call incl. the all-reduce and the barriers around it, max over ranks, "
                                             "median of 7",
                                   "collective": f"one all-reduce(SUM) of 2 x {n_sets} int64 per call ({backend}): the user set's "
                                                 f"and the universe's support vectors (enrichment.rs:198-221)",
                                   "local_db_intervals_rank0": sdb.local_intervals, "per_rank_ms": lola_per_rank, "note": STRONG_NOTE,
                                   "verified": "a,b,c,d identities; support sum == the single-GPU value"}
    del sdb, hboth
    torch.cuda.empty_cache()
    # ---- config 5: the fragment pipeline, files dealt to the ranks (no collective on the data path) ----
    out["fragsplit_config5_sharded"] = bench_fragsplit_sharded(dist, rank, world, max_over_ranks, barrier, files=max(8, 32 // scale),
                                                               frags=100_000 // scale)
    # ... and with the single-GPU object's 48 files PER RANK (weak: what "x times the throughput at N GPUs" is read from)
    out["fragsplit_config5_weak"] = bench_fragsplit_sharded(dist, rank, world, max_over_ranks, barrier,
                                                            files=max(2, 48 // scale) * world, frags=100_000 // scale, weak=True)
    if scale != 1:
        out["configs_scaled_down_by"] = scale
    return out


def bench_fragsplit_sharded(dist, rank, world, max_over_ranks, barrier, files=32, frags=100_000, clusters=20, weak=False):
    """BASELINE config 5 over N ranks: rank 0 writes the synthetic fragment files (one node: the ranks share the directory), every
    rank runs the fused fragsplit -> tokenizer pipeline on its contiguous run of the sorted file list (balanced by compressed
    size) with its own host threads and its own GPU, results stay rank-local (`gather=False`: one output shard per GPU, what the
    config's 1e9 fragments call for).  Weak in files per rank it is not: the file count is fixed, so this is strong scaling."""
    import shutil
    import tempfile

    from gtars_amd import sharding, synth
    from gtars_amd.fragsplit import BarcodeToClusterMap, fragsplit_tokenize
    from gtars_amd.tokenizers import Tokenizer

    box = [None]
    if rank == 0:
        tmp = tempfile.mkdtemp(prefix="gtars_c5s_", dir=os.environ.get("TMPDIR", "/tmp"))
        t = time.time()
        ub, fd, mp, gz_bytes = synth.write_config5_inputs(tmp, synth.make_universe(100_000), files, frags, clusters)
        box[0] = (tmp, ub, fd, mp, gz_bytes, time.time() - t)
    dist.broadcast_object_list(box, src=0)
    tmp, ub, fd, mp, gz_bytes, t_gen = box[0]
    try:
        tok, m = Tokenizer.from_bed(ub), BarcodeToClusterMap.from_file(mp)
        sharding.fragsplit_tokenize_sharded(fd, m, tok, gather=False)  # warm-up (device buffers, page cache)
        times = []
        for _ in range(3):
            barrier()
            t0 = time.perf_counter()
            local = sharding.fragsplit_tokenize_sharded(fd, m, tok, gather=False)
            barrier()
            times.append(max_over_ranks(time.perf_counter() - t0))
        dt = statistics.median(times)
        man = local.pop("__manifest__")
        mine = {"files": man["files"], "token_ids": sum(int(v[1][-1]) for v in local.values())}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        n = files * frags
        o = {"files": files, "fragments_per_file": frags, "fragments": n, "clusters": clusters, "input_gz_MB": round(gz_bytes / 1e6, 1),
             "gen_s": round(t_gen, 1), "scaling": "weak" if weak else "strong", "value": n / dt, "unit": "fragments/s end to end (all ranks)",
             "host_threads_per_rank": int(_lib_host_threads()),
             "s": round(dt, 3), "per_rank": per_rank,
             "sharding": "contiguous runs of the sorted file list, balanced by compressed size; results rank-local (no collective on the "
                         "data path)",
             "timing": "host wall time incl. barriers, max over ranks, median of 3",
             "scale_note": f"{files} files x {frags} fragments = {n} of the config's 1e9 fragments (1/{round(1e9 / n)})"}
        if rank == 0:
            whole = fragsplit_tokenize(fd, m, tok, as_arrays=True)
            want = sum(int(v[1][-1]) for v in whole.values())
            got = sum(r["token_ids"] for r in per_rank)
            if want != got:
                raise SystemExit(f"bench.py: fragsplit_config5_sharded: {got} token ids over the ranks, {want} in one process")
            o["verified"] = "token ids summed over the ranks == the single-process pipeline's (tests/test_gpu_host.py compares every id)"
        return o
    finally:
        barrier()
        if rank == 0:
            shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--queries", type=int, default=1_000_000, help="query regions per step per GPU")
    ap.add_argument("--universe", type=int, default=100_000)
    ap.add_argument("--batches", type=int, default=32, help="distinct device-resident batches the steps rotate through")
    ap.add_argument("--min-seconds", type=float, default=0.6, help="GPU time to spend in timed repetitions (at least 11 repetitions)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not spawn the rocprofv3 --pmc passes that measure roofline.traffic")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline_large / igd_config3 / lola_config4 / host path")
    ap.add_argument("--large", type=str, default="16000000,64000000,256000000,1000000000", help="batch sizes of roofline_large (SURVEY 8d: 1.6e7 / 2.56e8 / 1e9, and 64M)")
    ap.add_argument("--c5-files", type=int, default=48, help="fragment files (1e5 fragments each) of fragsplit_config5 (the config names 10,000)")
    ap.add_argument("--scale-configs", type=int, default=1, help="tests only: divide the sizes of configs 3 / 4 (sharded objects) by this")
    ap.add_argument("--igd-pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.igd_pmc_child:
        return igd_pmc_child()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly: this process becomes the launcher of the N ranks (no GPU call has been made here)
        sys.exit(spawn_ranks(args.gpus))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE {world}: the launcher's world size is what runs", file=sys.stderr)
    n_gpus = max(world, 1)

    import gtars_amd
    from gtars_amd import _lib, synth

    if not torch.cuda.is_available() or gtars_amd.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: gtars_amd has no CPU fallback")
    # one rank per GPU; GTARS_BENCH_BACKEND=gloo lets several ranks share a GPU (plumbing test on a 1-GPU box)
    backend = os.environ.get("GTARS_BENCH_BACKEND", "nccl")
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: {world} ranks under nccl need {world} devices, {torch.cuda.device_count()} visible")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist_mod.init_process_group(backend="nccl", device_id=dev)
        else:
            dist_mod.init_process_group(backend=backend)
        dist = dist_mod

    # ---- synthetic workload: NB distinct batches per rank (its slices of the global stream of batches) ----
    nb = max(1, args.batches)
    nq = args.queries
    u = synth.make_universe(args.universe, seed=3)
    nu = len(u["chrom"])
    ix = gtars_amd.OverlapIndex(u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM)
    stream = torch.cuda.current_stream().cuda_stream
    q0 = None
    batches = []
    for b in range(nb):
        qb = synth.make_queries(u, nq, seed=4 + 100 * rank + 7919 * b)
        if b == 0:
            q0 = qb
        qc, qs, qe = (_dev(qb[k], dev) for k in ("chrom", "start", "end"))
        offsets = torch.empty(nq + 1, dtype=torch.int64, device=dev)
        batches.append([qc, qs, qe, offsets, None, 0])
    for bt in batches:  # size the id buffers: one offsets-only pass per batch
        qc, qs, qe, offsets = bt[:4]
        h = ix.tokenize_device(qc.data_ptr(), qs.data_ptr(), qe.data_ptr(), nq, offsets.data_ptr(), 0, 0, stream, sync=True)
        bt[4] = torch.empty(h + 1024, dtype=torch.int32, device=dev)
        bt[5] = h
    resident_bytes = sum(sum(t.numel() * t.element_size() for t in bt[:5]) for bt in batches)
    h_mean = sum(bt[5] for bt in batches) / nb

    counter = [0]

    def step(sync=False):
        qc, qs, qe, offsets, ids, _ = batches[counter[0] % nb]
        counter[0] += 1
        return ix.tokenize_device(qc.data_ptr(), qs.data_ptr(), qe.data_ptr(), nq, offsets.data_ptr(),
                                  ids.data_ptr(), ids.numel(), stream, sync=sync)

    for bt in batches:  # validates capacity / scan status of every batch once
        assert step(sync=True) == batches[(counter[0] - 1) % nb][5]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x: float) -> float:
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        step()
    # ---- timed repetitions: K steps each, barrier + synchronize on both sides, MAX over ranks; median reported ----
    ts = torch.cuda.current_stream()
    rep_wall, rep_kernel_ms = [], []
    gpu_s, reps_min, reps_max = 0.0, 11, 20000
    while len(rep_wall) < reps_min or (gpu_s < args.min_seconds and len(rep_wall) < reps_max):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        e0.record(ts)
        for _ in range(args.steps):
            step()
        e1.record(ts)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        rep_wall.append(dt)
        rep_kernel_ms.append(e0.elapsed_time(e1) / args.steps)
        gpu_s += dt
    elapsed = statistics.median(rep_wall)
    avg_ms = statistics.median(rep_kernel_ms)
    # the outputs the timed launches left behind, against the oracle (bit-exact): batch 0 on every rank
    verified = verify_tokenization(u, q0, batches[0][3], batches[0][4], batches[0][5], f"rank {rank}: the timed output of batch 0")
    # what every rank saw (device, backend): "did RCCL see N ranks" is answerable from the line
    me = {"rank": rank, "local_rank": local_rank, "device": dev_index, "name": torch.cuda.get_device_name(dev_index),
          "pci_bus_id": getattr(torch.cuda.get_device_properties(dev_index), "pci_bus_id", None),
          "visible_devices": torch.cuda.device_count(), "backend": backend, "pid": os.getpid()}
    ranks = [me]
    if dist is not None:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)

    # ---- the dominant kernel: its name from the library's profiling hooks, its duration from the events above ----
    roofline = None
    if rank == 0:
        _lib.lib.gtars_prof_reset()
        _lib.lib.gtars_prof_enable(1)
        step()
        torch.cuda.synchronize()
        names = [n for n in _lib.prof_read() if n.startswith("k_")]  # (kernels; the library also notes facts such as the build it chose)
        _lib.lib.gtars_prof_enable(0)
        name = names[0] if len(names) == 1 else "+".join(names)
        bytes_per_launch = algorithmic_bytes(nq, round(h_mean), nu)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        traffic, traffic_source, fetch_b, write_b = None, "not measured in this run (needs rocprofv3 --pmc passes)", None, None
        tpath = os.path.join(ROOT, "profiles", PROFILE_ROUND, "traffic_tokenize_1M.json")
        if world == 1 and not args.no_extras and not args.no_pmc:
            live = measure_traffic_with_rocprof(["--steps", "20", "--warmup", "5", "--queries", str(nq), "--universe", str(args.universe),
                                                 "--no-cpu-baseline", "--no-extras", "--no-pmc", "--min-seconds", "0.05"])
            tok = [v for k, v in (live or {}).items() if "k_tok_lds" in k and v["fetch_dispatches"] and v["write_dispatches"]]
            if tok:
                # each counter's sum over ITS OWN pass's dispatches (the passes are separate runs of a --min-seconds loop)
                fetch_b = sum(v["fetch"] for v in tok) / sum(v["fetch_dispatches"] for v in tok)
                write_b = sum(v["write"] for v in tok) / sum(v["write_dispatches"] for v in tok)
                traffic = fetch_b + write_b
                traffic_source = ("measured in this run: two child rocprofv3 --pmc passes (FETCH_SIZE x2 for gfx950, then WRITE_SIZE) of a "
                                  "short invocation of this script; each counter's mean over the k_tok_lds dispatches of its own pass")
        if traffic is None and os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj["workload"]["queries_per_step"] == nq and tj["workload"]["universe_regions"] == args.universe:
                traffic = tj["traffic_bytes_per_launch"]
                fetch_b, write_b = tj.get("fetch_bytes_per_launch"), tj.get("write_bytes_per_launch")
                traffic_source = f"replayed from profiles/{PROFILE_ROUND}/traffic_tokenize_1M.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"
        roofline = {
            "bound": "hbm",
            "kernel": name,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_vs_algorithmic": (traffic / bytes_per_launch) if traffic else None,
            "fetch_bytes": fetch_b,
            "write_bytes": write_b,
            "traffic_source": traffic_source,
            "avg_kernel_ms": avg_ms,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "bytes_per_query": bytes_per_launch / nq,
            "frac_of_measured_copy_6.29TBps": achieved / 6290.0,
            "ceiling": "single-pass design (every workgroup stages the 132-KB search image, one chained scan): floor at 1M queries ~15 us = "
                       "0.20 of the roofline (3.4 us prologue of 34 MB L2->LDS traffic chip-wide + 3.6 us look-back wait on the slowest "
                       "predecessor); large-batch ceiling 0.48 (one divergent 32-byte record per shuffled query: 2.3 clk of the CU's one "
                       "vector-memory path); the 0.40 target is not met at 1M -- see roofline_large for 16M..1e9 and DESIGN.md section 3",
        }

    # ---- the same steps on POSITION-SORTED batches: what Tokenizer.tokenize(path) delivers (a file-loaded RegionSet is sorted by
    # (chr, start), gtars-core/src/models/region_set.rs:182) -- reported next to the shuffled headline, never instead of it ----
    sorted_input = None
    if rank == 0 and world == 1 and not args.no_extras:
        ns_b = min(nb, 8)  # 8 x 45 MB of queries and results: past the Infinity Cache like the headline's rotation
        sb = []
        for b in range(ns_b):
            qb = q0 if b == 0 else synth.make_queries(u, nq, seed=4 + 7919 * b)
            o = np.lexsort((qb["start"], qb["chrom"]))
            qs_sorted = {k: np.ascontiguousarray(qb[k][o]) for k in ("chrom", "start", "end")}
            d = [_dev(qs_sorted[k], dev) for k in ("chrom", "start", "end")]
            sb.append((d, torch.empty(nq + 1, dtype=torch.int64, device=dev), torch.empty(batches[b][5] + 1024, dtype=torch.int32, device=dev),
                       qs_sorted if b == 0 else None))
        turn = [0]

        def sstep():
            d, off, ids, _ = sb[turn[0] % ns_b]
            turn[0] += 1
            return ix.tokenize_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), nq, off.data_ptr(), ids.data_ptr(), ids.numel(),
                                      stream, sync=False)

        for _ in range(max(args.warmup, ns_b)):
            sstep()
        torch.cuda.synchronize()
        ks = []
        for _ in range(11):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ts)
            for _ in range(args.steps):
                sstep()
            e1.record(ts)
            torch.cuda.synchronize()
            ks.append(e0.elapsed_time(e1) / args.steps)
        sms = statistics.median(ks)
        d0, off0, ids0, q_sorted0 = sb[0]
        sv = verify_tokenization(u, q_sorted0, off0, ids0, batches[0][5], "the timed output of sorted batch 0")
        sorted_input = {"value": nq / (sms * 1e-3), "unit": "query intervals/s", "ms_per_step": sms,
                        "frac": algorithmic_bytes(nq, round(h_mean), nu) / (sms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "what": f"the same {nq}-query batches in (chromosome, start) order, {ns_b} rotating; HIP events over {args.steps} "
                                f"steps, median of 11", "verified": sv}
        del sb
        torch.cuda.empty_cache()

    # ---- N > 1: the same steps when one consumer needs the whole batch (all-gatherv of the CSR over RCCL) ----
    with_allgather = None
    if dist is not None:
        from gtars_amd import sharding

        # warm-up: communicator setup, and one size exchange per distinct batch -- the per-rank (queries, ids) shapes of a
        # batch do not change between steps, so the timed steps reuse its plan (no host round trip in the steady state)
        plans = []
        for qc, qs, qe, offsets, ids, h in batches:
            plans.append(sharding.all_gather_csr_device(offsets, ids, h, return_plan=True)[2])
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            qc, qs, qe, offsets, ids, h = batches[i % nb]
            ix.tokenize_device(qc.data_ptr(), qs.data_ptr(), qe.data_ptr(), nq, offsets.data_ptr(), ids.data_ptr(),
                               ids.numel(), stream, sync=False)
            sharding.all_gather_csr_device(offsets, ids, h, plan=plans[i % nb])
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        with_allgather = {"value": nq * n_gpus * args.steps / dt, "unit": "query intervals/s", "ms_per_step": dt / args.steps * 1e3,
                          "note": "tokenize + all-gatherv of per-query counts (u32) and ids (u32) so that every rank holds the global "
                                  "CSR; per-rank sizes exchanged once per distinct batch (warm-up), host wall time incl. barriers"}

    sharded = None
    if dist is not None and not args.no_extras:
        del batches[:]
        torch.cuda.empty_cache()
        sharded = bench_sharded(dist, backend, dev, rank, world, max_over_ranks, barrier, max(1, args.scale_configs))

    if rank == 0:
        total_q = nq * n_gpus * args.steps
        out = {
            "metric": "query intervals/sec tokenized vs 100k-region hg38 universe",
            "value": total_q / elapsed,
            "unit": "query intervals/s",
            "n_gpus": n_gpus,
            "world_size": world,
            "backend": backend if world > 1 else None,
            "ranks": ranks,
            "verified": "every rank: offsets + ids of timed batch 0 == oracle (bit-exact)" if verified else None,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE config 2: tokenize hg38-shaped query regions vs 100k-region universe "
                            "(Bits order, CSR u64 offsets + u32 token ids), inputs resident in HBM",
                "queries_per_step_per_gpu": nq,
                "universe_regions": nu,
                "hits_per_step_per_gpu": round(h_mean),
                "distinct_batches": nb,
                "resident_bytes_rotated": resident_bytes,
                "sharding": "independent query ranges per rank, universe index replicated, no collective",
            },
            "timing": {"repetitions": len(rep_wall), "statistic": "median over repetitions of K back-to-back steps",
                       "timed_seconds": sum(rep_wall), "min_ms_per_step": min(rep_wall) / args.steps * 1e3,
                       "max_ms_per_step": max(rep_wall) / args.steps * 1e3},
            "roofline": roofline,
        }
        if world > 1:
            out["headline_scaling_note"] = (
                "the headline is weak-scaled replicas by construction: every rank tokenizes its own 1M-query batches against a replicated "
                "index and nothing crosses ranks, so value at N GPUs is ~N x the single-GPU value and says nothing about the interconnect. "
                "Read scaling from the objects that exchange or partition data: with_allgather (CSR all-gatherv), igd_config3_weak "
                "(chromosome buckets + one all-reduce per call, the genome grows with the ranks), igd_config3_sharded / "
                "lola_config4_sharded (strong-scaled, latency-bound at 8 ranks: see their per_rank_ms) and fragsplit_config5_weak")
        if sorted_input:
            out["sorted_input"] = sorted_input
        if with_allgather:
            out["with_allgather"] = with_allgather
        if sharded:
            out.update(sharded)
        if world == 1 and not args.no_extras:
            sizes = [int(t) for t in args.large.split(",") if t]
            out["roofline_large"] = bench_large(ix, u, q0, nu, sizes, dev, stream)
            # the same kernel on a 64M-query batch IN ORDER (what Tokenizer.tokenize(path) delivers at that size): neighbouring
            # lanes' record requests coalesce.  Next to the shuffled headline, never instead of it.
            out["sorted_input_large"] = bench_large(ix, u, q0, nu, [64_000_000], dev, stream, in_order=True)[0]
            # hit-heavy batches (not a BASELINE config): every query of the base batch widened to 1 Mbp -- ~33 ids per query,
            # the ids dominate the bytes; wide queries' tails are measured instead of walked and their ids leave by wave-wide
            # stores (tokenize_lds.hip: tail_run / coop_runs)
            qw = dict(q0)
            qw["end"] = np.minimum(q0["start"].astype(np.int64) + 1_000_000, 0x7FFFFFFF).astype(q0["end"].dtype)
            hh = bench_large(ix, u, qw, nu, [16_000_000], dev, stream)[0]
            hh["ids_per_query"] = round(hh["hits"] / hh["queries"], 1)
            hh["batch"] = "the 1M-query base batch with every query widened to 1 Mbp, tiled 16x on the device"
            out["roofline_hit_heavy"] = hh
            # ... and on the ChIP-like universe (C2': overlapping neighbours, 1 % intervals of 5-100 kbp; sorted file order so that
            # the ids follow from the position): the run form with records tested in front of the run
            uo = synth.make_universe(nu, overlapping=True)
            order = np.lexsort((uo["end"], uo["start"], uo["chrom"]))
            uo = {k: np.ascontiguousarray(v[order]) for k, v in uo.items()}
            ixo = gtars_amd.OverlapIndex(uo["chrom"], uo["start"], uo["end"], n_chrom=synth.N_CHROM)
            ho = bench_large(ixo, uo, qw, nu, [16_000_000], dev, stream)[0]
            ho["ids_per_query"] = round(ho["hits"] / ho["queries"], 1)
            ho["batch"] = hh["batch"]
            ho["universe"] = "synth.make_universe(overlapping=True), position-sorted"
            out["roofline_hit_heavy_overlapping"] = ho
            del ixo
            # universes beyond the LDS key budget of k_tok_lds (~65k 16-bit keys = 130k regions): 2 blocks per key at 140k / 200k
            # regions, 8 at 1M ("the universe-size cliff") -- the same 64M-query measurement as roofline_large,
            # against each universe's own oracle tokenization
            for nu2, key in ((140_000, "roofline_universe_140k"), (200_000, "roofline_universe_200k"), (1_000_000, "roofline_universe_1M")):
                u2 = synth.make_universe(nu2)
                q2 = synth.make_queries(u2, nq)
                ix2 = gtars_amd.OverlapIndex(u2["chrom"], u2["start"], u2["end"], n_chrom=synth.N_CHROM)
                r2 = bench_large(ix2, u2, q2, nu2, [64_000_000], dev, stream)[0]
                r2["universe_regions"] = nu2
                r2["batch"] = f"1M queries drawn against this universe (synth.make_queries), tiled 64x on the device"
                out[key] = r2
                r3 = bench_large(ix2, u2, q2, nu2, [64_000_000], dev, stream, in_order=True)[0]
                r3["universe_regions"] = nu2
                out[key + "_in_order"] = r3
                del ix2
                torch.cuda.empty_cache()
            # PCIe-inclusive rates through the host-pointer entry points (H2D of the queries, kernel, D2H of offsets +
            # ids).  Reported for context only; never `value`.  `streaming`: gtars_tokenize_into with
            # output arrays the caller reuses (chunked copy / kernel / copy-back pipeline, nothing allocated);
            # `allocating`: gtars_tokenize, which returns freshly allocated arrays on every call.
            hb = {"unit": "query intervals/s", "note": "pageable numpy buffers in and out, one call per 1M-query batch, median of 9 calls"}
            out_bufs = (np.empty(nq + 1, dtype=np.uint64), np.empty(2 * nq + 1024, dtype=np.uint32))
            for key, kw in (("streaming", {"out": out_bufs}), ("allocating", {})):
                ix.tokenize(q0["chrom"], q0["start"], q0["end"], **kw)
                ts_ = []
                for _ in range(9):
                    t1 = time.perf_counter()
                    ix.tokenize(q0["chrom"], q0["start"], q0["end"], **kw)
                    ts_.append(time.perf_counter() - t1)
                hb[key] = nq / statistics.median(ts_)
            hb["value"] = hb["streaming"]
            out["host_buffers_end_to_end"] = hb
            del batches[:]
            torch.cuda.empty_cache()
            cpu = not args.no_cpu_baseline
            out["igd_config3"] = bench_igd_config3(dev, stream, cpu=cpu, pmc=not args.no_pmc)
            out["igd_config3_broad_peaks"] = bench_igd_broad_peaks(dev, stream)
            out["lola_config4"] = bench_lola_config4(dev, stream, cpu=cpu)
            out["fragsplit_config5"] = bench_fragsplit_config5(files=args.c5_files)
            out["fragsplit_config5_1000"] = bench_fragsplit_many_files()
        if not args.no_cpu_baseline and world == 1:  # a reported baseline: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(u, q0)
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(u, q0)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
