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

N > 1 is launched by the driver through torch.distributed.run (one rank per
GPU).  The path shards by independent query ranges with the universe index
replicated, so there is no data-path collective (weak scaling: every rank
tokenizes its own Q-query slice of the global batch); torch.distributed is used
only for the timing barrier and the max-over-ranks reduction.

Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md chip table)


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
    """Courtesy multi-core figure (SURVEY section 8d ii): the same oracle on many host threads at once (ctypes
    releases the GIL), every thread tokenizing whole batches into its own outputs.  Reported next to, not
    instead of, the reference-faithful single-thread baseline."""
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
    stop = [False]

    def work(t):
        # whole batches per call (~65 ms each), so the GIL is only touched between calls
        off = np.zeros(nq + 1, dtype=np.uint64)
        ids = np.zeros(2 * nq + 16, dtype=np.uint32)
        done = 0
        while not stop[0]:
            L.orc_tokenize(ref._h, qc, qs, qe, nq, off, ids, len(ids))
            done += nq
        return done

    with ThreadPoolExecutor(max_workers=threads) as ex:
        t0 = time.perf_counter()
        futs = [ex.submit(work, t) for t in range(threads)]
        time.sleep(budget_s)
        stop[0] = True
        total = sum(f.result() for f in futs)
        dt = time.perf_counter() - t0
    return {"value": total / dt, "unit": "query intervals/s", "cores": threads, "kind": "port",
            "sample": f"{total // nq} x {nq} queries of the same workload through orc_tokenize on {threads} host "
                      f"threads (each thread tokenizes whole batches, {dt:.1f} s; CPUs usable under the affinity mask "
                      f"and cgroup quota: {avail})"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--queries", type=int, default=1_000_000, help="query regions per step per GPU")
    ap.add_argument("--universe", type=int, default=100_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sweep", type=str, default="", help="comma-separated extra batch sizes to report")
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

    # ---- synthetic workload (per rank: its own slice of the global batch) ----
    u = synth.make_universe(args.universe, seed=3)
    q = synth.make_queries(u, args.queries, seed=4 + 100 * rank)
    nu, nq = len(u["chrom"]), len(q["chrom"])
    ix = gtars_amd.OverlapIndex(u["chrom"], u["start"], u["end"], n_chrom=synth.N_CHROM)
    qc, qs, qe = (torch.from_numpy(q[k].view(np.int32)).to(dev) for k in ("chrom", "start", "end"))
    offsets = torch.empty(nq + 1, dtype=torch.int64, device=dev)
    ids = torch.empty(2 * nq + 1024, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step(sync=False):
        return ix.tokenize_device(qc.data_ptr(), qs.data_ptr(), qe.data_ptr(), nq, offsets.data_ptr(),
                                  ids.data_ptr(), ids.numel(), stream, sync=sync)

    h = step(sync=True)  # also validates capacity / scan status once

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- dominant-kernel duration with HIP events on the launch stream ----
    # One step is exactly one k_tok_lds launch (the chained-scan workspace is epoch-tagged, so there is
    # no memset kernel); K launches are bracketed by ONE pair of HIP events recorded on the stream the
    # kernel is launched on, so the average includes the ~1 us inter-launch gap but no event overhead.
    roofline = None
    if rank == 0:
        _lib.lib.gtars_prof_reset()
        _lib.lib.gtars_prof_enable(1)
        step()
        torch.cuda.synchronize()
        names = list(_lib.prof_read())
        _lib.lib.gtars_prof_enable(0)
        name = names[0] if len(names) == 1 else "+".join(names)
        prof_steps = max(args.steps, 50)
        ts = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ts)
        for _ in range(prof_steps):
            step()
        e1.record(ts)
        torch.cuda.synchronize()
        avg_ms = e0.elapsed_time(e1) / prof_steps
        bytes_per_launch = algorithmic_bytes(nq, h, nu)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01", "traffic_tokenize_1M.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj["workload"]["queries_per_step"] == nq and tj["workload"]["universe_regions"] == args.universe:
                traffic = tj["traffic_bytes_per_launch"]  # rocprofv3 PMC passes, see that file for the recipe
        roofline = {
            "bound": "hbm",
            "kernel": name,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "avg_kernel_ms": avg_ms,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "bytes_per_query": bytes_per_launch / nq,
        }

    sweep = []
    if rank == 0 and args.sweep:
        for tok in args.sweep.split(","):
            n2 = int(tok)
            rep = max(n2 // nq, 1)
            big = {k: torch.from_numpy(np.tile(q[k], rep).view(np.int32)).to(dev) for k in ("chrom", "start", "end")}
            n2 = nq * rep
            off2 = torch.empty(n2 + 1, dtype=torch.int64, device=dev)
            ids2 = torch.empty(h * rep + 1024, dtype=torch.int32, device=dev)
            run = lambda s=False: ix.tokenize_device(big["chrom"].data_ptr(), big["start"].data_ptr(),
                                                     big["end"].data_ptr(), n2, off2.data_ptr(), ids2.data_ptr(),
                                                     ids2.numel(), stream, sync=s)
            h2 = run(True)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            k = max(2, min(50, int(2e9 // n2)))
            t1 = time.perf_counter()
            for _ in range(k):
                run()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / k
            sweep.append({"queries": n2, "ms_per_step": dt * 1e3, "qps": n2 / dt,
                          "hbm_frac": algorithmic_bytes(n2, h2, nu) / dt / 1e9 / HBM_PEAK_GBS})
            del big, off2, ids2

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
                "hits_per_step_per_gpu": h,
                "sharding": "independent query ranges per rank, universe index replicated, no collective",
            },
            "roofline": roofline,
        }
        if sweep:
            out["batch_sweep"] = sweep
        if world == 1:
            # PCIe-inclusive rate through the host-pointer entry point (gtars_tokenize: H2D, kernel, D2H of
            # offsets + ids).  Reported for context only; it is never `value`.
            ix.tokenize(q["chrom"], q["start"], q["end"])
            t1 = time.perf_counter()
            for _ in range(5):
                ix.tokenize(q["chrom"], q["start"], q["end"])
            out["host_buffers_end_to_end"] = {"value": 5 * nq / (time.perf_counter() - t1), "unit": "query intervals/s",
                                              "note": "pageable numpy buffers in and out, one call per batch"}
            out["roofline"]["frac_of_measured_copy_6.29TBps"] = roofline["achieved"] / 6290.0
        if not args.no_cpu_baseline and world == 1:  # a reported baseline: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(u, q)
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(u, q)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
