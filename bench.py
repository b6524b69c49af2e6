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

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU).  The tokenizer shards by
independent query ranges with the universe index replicated, so there is no data-path collective (weak
scaling: every rank tokenizes its own Q-query batches); `with_allgather` additionally reports the rate when
every rank must end up with the whole batch's CSR (RCCL all-gatherv of offsets and ids, gtars_amd/sharding.py).

Rank 0 prints ONE JSON line.  At N = 1 it also carries: larger batches (`roofline_large`), BASELINE config 3
(`igd_config3`) and config 4 on one GPU (`lola_config4`), the PCIe-inclusive host-buffer rate and the CPU
baselines (the oracle on the GPU box's host cores).
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
PROFILE_ROUND = "r02"


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


def bench_large(ix, q, nu, sizes, dev, stream):
    """The same kernel on larger batches (the 1M-query base batch tiled on the device): queries/s and fraction of
    the HBM peak, median of 5 single-launch timings each (HIP events on the launch stream)."""
    import torch

    out = []
    base = {k: _dev(q[k], dev) for k in ("chrom", "start", "end")}
    nq0 = base["chrom"].numel()
    for n2 in sizes:
        rep = max(n2 // nq0, 1)
        big = {k: v.repeat(rep) for k, v in base.items()}
        n2 = nq0 * rep
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
        byts = algorithmic_bytes(n2, h2, nu)
        out.append({"queries": n2, "hits": h2, "ms": dt * 1e3, "qps": n2 / dt, "achieved_GBps": byts / dt / 1e9,
                    "frac": byts / dt / 1e9 / HBM_PEAK_GBS})
        del big, off, ids
        torch.cuda.empty_cache()
    return out


def bench_igd_config3(dev, stream, ndb=50_000_000, nq=10_000_000, n_files=1000):
    """BASELINE config 3: 10M shuffled synthetic intervals vs a 50M-interval, 1000-file IGD database on one GPU.
    Bytes = 12*Nq + 16*Ndb + 8*F.  The headline is the batch AS SPECIFIED (shuffled); the same batch in
    (chromosome, start) order -- what a sorted BED file delivers -- is reported next to it."""
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
    del db
    hits = torch.zeros(n_files, dtype=torch.int64, device=dev)
    byts = 12 * nq + 16 * ndb + 8 * n_files
    out = {"db_intervals": ndb, "queries": nq, "files": n_files, "gen_s": round(tgen, 2), "build_s": round(tbuild, 2),
           "algorithmic_bytes": byts}
    order = np.lexsort((q["start"], q["chrom"]))
    for label, sel in (("shuffled", None), ("sorted_input", order)):
        qc, qs, qe = (_dev(q[k] if sel is None else q[k][sel], dev) for k in ("chrom", "start", "end"))
        for binary in (False, True):
            f = lambda: g.count_device(qc.data_ptr(), qs.data_ptr(), qe.data_ptr(), nq, hits.data_ptr(), 1, binary, stream)
            f()
            torch.cuda.synchronize()
            times = []
            for _ in range(5):
                t0 = time.perf_counter()
                f()
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            dt = statistics.median(times)
            out[("binary" if binary else "pairwise") + "_" + label] = {
                "ms": round(dt * 1e3, 3), "qps": round(nq / dt), "frac": round(byts / dt / 1e9 / HBM_PEAK_GBS, 5),
                "total_hits": int(hits.sum())}
        del qc, qs, qe
    del g
    torch.cuda.empty_cache()
    return out


def measure_traffic_with_rocprof(nq, universe):
    """HBM-side bytes per k_tok_lds launch from two rocprofv3 --pmc passes (children of this process; nothing else
    traced) over a short run of this script on the same workload.  -> (bytes, source) or None."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    tmp = tempfile.mkdtemp(prefix="gtars_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    means = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "20", "--warmup", "5", "--queries", str(nq), "--universe", str(universe), "--no-cpu-baseline",
                   "--no-extras", "--no-pmc", "--min-seconds", "0.05"]
            r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=240, env=dict(os.environ, TMPDIR=tmp))
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            vals = [float(row["Counter_Value"]) for row in csv.DictReader(open(max(files, key=os.path.getmtime)))
                    if "k_tok_lds" in row["Kernel_Name"] and row["Counter_Name"] == counter]
            if not vals:
                return None
            means[counter] = sum(vals) / len(vals) * 1024.0  # the counters are in KB
        traffic = 2.0 * means["FETCH_SIZE"] + means["WRITE_SIZE"]
        return traffic, ("measured in this run: mean over k_tok_lds dispatches of two child rocprofv3 --pmc passes "
                         "(FETCH_SIZE x2 for gfx950 + WRITE_SIZE) of a short invocation of this script")
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def bench_lola_config4(dev, stream, n_sets=2000, per_set=25_000, n_universe=1_000_000, n_user=100_000):
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
    del db
    uq = [_dev(uni[k], dev) for k in ("chrom", "start", "end")]
    sq = [_dev(uni[k][sel], dev) for k in ("chrom", "start", "end")]
    uh = torch.zeros(n_sets, dtype=torch.int64, device=dev)
    sh = torch.zeros(n_sets, dtype=torch.int64, device=dev)
    cells = [torch.empty(n_sets, dtype=torch.int64, device=dev) for _ in range(4)]
    nuni = len(uni["chrom"])

    def run():
        g.count_device(uq[0].data_ptr(), uq[1].data_ptr(), uq[2].data_ptr(), nuni, uh.data_ptr(), 1, True, stream)
        g.count_device(sq[0].data_ptr(), sq[1].data_ptr(), sq[2].data_ptr(), n_user, sh.data_ptr(), 1, True, stream)
        check(lib.gtars_lola_contingency_device(sh.data_ptr(), uh.data_ptr(), n_sets, n_user, nuni,
                                                *[c.data_ptr() for c in cells], stream))

    run()
    torch.cuda.synchronize()
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    a, b, c, d = [x.cpu().numpy() for x in cells]
    ok = bool(((a + b) == uh.cpu().numpy()).all() and ((a + c) == n_user).all() and ((a + b + c + d) == nuni).all())
    del g
    torch.cuda.empty_cache()
    return {"sets": n_sets, "db_intervals": n_sets * per_set, "universe": nuni, "user": n_user, "build_s": round(tb, 2),
            "counts_ms": round(dt * 1e3, 3), "identities_hold": ok, "support_sum": int(a.sum())}


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
    ap.add_argument("--large", type=str, default="64000000,256000000,1000000000", help="batch sizes of roofline_large")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    n_gpus = max(world, 1)

    import gtars_amd
    from gtars_amd import _lib, synth

    if not torch.cuda.is_available() or gtars_amd.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: gtars_amd has no CPU fallback")
    # one rank per GPU; GTARS_BENCH_BACKEND=gloo lets several ranks share a GPU (plumbing test on a 1-GPU box)
    backend = os.environ.get("GTARS_BENCH_BACKEND", "nccl")
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

    # ---- the dominant kernel: its name from the library's profiling hooks, its duration from the events above ----
    roofline = None
    if rank == 0:
        _lib.lib.gtars_prof_reset()
        _lib.lib.gtars_prof_enable(1)
        step()
        torch.cuda.synchronize()
        names = list(_lib.prof_read())
        _lib.lib.gtars_prof_enable(0)
        name = names[0] if len(names) == 1 else "+".join(names)
        bytes_per_launch = algorithmic_bytes(nq, round(h_mean), nu)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        traffic, traffic_source = None, "not measured in this run (needs rocprofv3 --pmc passes)"
        tpath = os.path.join(ROOT, "profiles", PROFILE_ROUND, "traffic_tokenize_1M.json")
        if world == 1 and not args.no_extras and not args.no_pmc:
            live = measure_traffic_with_rocprof(nq, args.universe)
            if live is not None:
                traffic, traffic_source = live
        if traffic is None and os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj["workload"]["queries_per_step"] == nq and tj["workload"]["universe_regions"] == args.universe:
                traffic = tj["traffic_bytes_per_launch"]
                traffic_source = f"replayed from profiles/{PROFILE_ROUND}/traffic_tokenize_1M.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"
        roofline = {
            "bound": "hbm",
            "kernel": name,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "avg_kernel_ms": avg_ms,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "bytes_per_query": bytes_per_launch / nq,
            "frac_of_measured_copy_6.29TBps": achieved / 6290.0,
        }

    # ---- N > 1: the same steps when one consumer needs the whole batch (all-gatherv of the CSR over RCCL) ----
    with_allgather = None
    if dist is not None:
        from gtars_amd import sharding

        qc, qs, qe, offsets, ids, h = batches[0]
        sharding.all_gather_csr_device(offsets, ids, h)  # warm-up (communicator setup)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            qc, qs, qe, offsets, ids, h = batches[i % nb]
            ix.tokenize_device(qc.data_ptr(), qs.data_ptr(), qe.data_ptr(), nq, offsets.data_ptr(), ids.data_ptr(),
                               ids.numel(), stream, sync=False)
            sharding.all_gather_csr_device(offsets, ids, h)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        with_allgather = {"value": nq * n_gpus * args.steps / dt, "unit": "query intervals/s", "ms_per_step": dt / args.steps * 1e3,
                          "note": "tokenize + all-gatherv of offsets (u64) and ids (u32) so that every rank holds the global CSR"}

    if rank == 0:
        total_q = nq * n_gpus * args.steps
        out = {
            "metric": "query intervals/sec tokenized vs 100k-region hg38 universe",
            "value": total_q / elapsed,
            "unit": "query intervals/s",
            "n_gpus": n_gpus,
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
        if with_allgather:
            out["with_allgather"] = with_allgather
        if world == 1 and not args.no_extras:
            sizes = [int(t) for t in args.large.split(",") if t]
            out["roofline_large"] = bench_large(ix, q0, nu, sizes, dev, stream)
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
            out["igd_config3"] = bench_igd_config3(dev, stream)
            out["lola_config4"] = bench_lola_config4(dev, stream)
        if not args.no_cpu_baseline and world == 1:  # a reported baseline: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(u, q0)
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(u, q0)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
