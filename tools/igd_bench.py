#!/usr/bin/env python3
"""BASELINE config 3: IGD batch query, synthetic intervals vs an indexed multi-file database.

  python tools/igd_bench.py                                  # one GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         tools/igd_bench.py --gpus N [--mode bucket|range]   # N GPUs: database + queries shard by chromosome bucket,
                                                             # one RCCL all-reduce of the F-long hit vector per count

Sizes: NDB / NQ / F environment variables (5e7 / 1e7 / 1000).  GTARS_BENCH_BACKEND=gloo lets several ranks share one
GPU (plumbing test).  Rank 0 prints one JSON line.
"""
import argparse, json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--mode", default="bucket", choices=["bucket", "range"])
    args = ap.parse_args()
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    import torch
    import gtars_amd
    from gtars_amd import sharding, synth

    backend = os.environ.get("GTARS_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, **({"device_id": dev} if backend == "nccl" else {}))

    ndb = int(os.environ.get("NDB", "50000000")); nq = int(os.environ.get("NQ", "10000000")); F = int(os.environ.get("F", "1000"))
    t = time.time(); db = synth.make_igd_db(ndb, F); q = synth.make_background_queries(nq); tgen = time.time() - t
    eng = sharding.HipEngine(dev)
    t = time.time()
    sdb = sharding.ShardedIgd(eng, db, synth.N_CHROM, F, mode=args.mode, balance_with=[q["chrom"]])
    tbuild = time.time() - t
    del db
    byts = 12 * nq + 16 * ndb + 8 * F
    out = {"n_gpus": world, "mode": args.mode if world > 1 else "single", "ndb": ndb, "nq": nq, "F": F, "gen_s": round(tgen, 2),
           "build_s": round(tbuild, 2), "local_db_intervals": sdb.local_intervals}

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def timed(handle, binary):
        hits = torch.zeros(F, dtype=torch.int64, device=dev)
        sdb.count_resident(handle, 1, binary, hits)
        times = []
        for _ in range(5):
            barrier()
            t0 = time.perf_counter()
            sdb.count_resident(handle, 1, binary, hits)
            barrier()
            times.append(time.perf_counter() - t0)
        dt = statistics.median(times)
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return {"ms": round(dt * 1e3, 3), "qps": round(nq / dt), "hbm_frac_of_all_gpus": round(byts / dt / 8e12 / world, 5),
                "total_hits": int(hits.sum())}

    h = sdb.upload_local(q)
    out["local_queries"] = int(h[0].numel())
    for binary in (False, True):
        out["binary" if binary else "pairwise"] = timed(h, binary)
    # the same batch already in (chromosome, start) order, as a sorted BED file would deliver it: no device sort
    order = np.lexsort((q["start"], q["chrom"]))
    hs = sdb.upload_local({k: q[k][order] for k in ("chrom", "start", "end")})
    for binary in (False, True):
        out[("binary" if binary else "pairwise") + "_sorted_input"] = timed(hs, binary)
    del hs
    if world == 1:
        from gtars_amd import _lib
        _lib.lib.gtars_prof_reset(); _lib.lib.gtars_prof_enable(1)
        for binary in (False, True):
            sdb.count_resident(h, 1, binary)
        torch.cuda.synchronize()
        out["kernels_ms"] = {k: round(v["total_ms"], 3) for k, v in _lib.prof_read().items()}
        _lib.lib.gtars_prof_enable(0)
    # parity at this size is asserted by tests/test_gpu_parity.py::test_config3_igd_full_size_properties
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
