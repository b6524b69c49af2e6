#!/usr/bin/env python3
"""BASELINE config 4: LOLA support counts, 1 user set + universe vs a 2,000-set region DB.

  python tools/lola_bench.py                                  # one GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         tools/lola_bench.py --gpus N                         # the DB and both query sets shard by chromosome bucket;
                                                              # ONE RCCL all-reduce of the 2 x F support counts

Rank 0 prints one JSON line (the integer cells a, b, c, d are checked against the identities of enrichment.rs:198-221)."""
import argparse, json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    args = ap.parse_args()
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    import torch
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

    F = int(os.environ.get("F", "2000")); per = int(os.environ.get("PER", "25000"))
    nuni = int(os.environ.get("NUNI", "1000000")); nuser = int(os.environ.get("NUSER", "100000"))
    db = synth.make_igd_db(F * per, F, seed=6)
    uni = synth.make_universe(nuni, seed=3)
    rng = np.random.default_rng(9)
    sel = np.sort(rng.choice(len(uni["chrom"]), nuser, replace=False))
    user = {k: uni[k][sel] for k in ("chrom", "start", "end")}
    eng = sharding.HipEngine(dev)
    t = time.time()
    sdb = sharding.ShardedIgd(eng, db, synth.N_CHROM, F, mode="bucket", balance_with=[uni["chrom"]])
    tb = time.time() - t
    del db
    hboth = sdb.upload_local_sets([uni, user])
    stacked = torch.zeros(2, F, dtype=torch.int64, device=dev)
    by_set = os.environ.get("SET_BY_SET") == "1"  # A/B: one count per set instead of the shared pass
    hu, hs = (sdb.upload_local(uni), sdb.upload_local(user)) if by_set else (None, None)

    def run():
        # both support vectors into one 2 x F buffer (one pass over the local region DB), ONE all-reduce, then the cells
        if by_set:
            eng.igd_count_resident(sdb.g, hu, 1, True, stacked[0], sync=False)
            eng.igd_count_resident(sdb.g, hs, 1, True, stacked[1], sync=False)
            if world > 1:
                sharding.all_reduce_hits_(stacked)
        else:
            sdb.count_sets_resident(hboth, 1, True, stacked)
        return sharding.contingency(stacked[1:], stacked[0], [nuser], len(uni["chrom"]))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    run()
    times = []
    for _ in range(5):
        barrier()
        t0 = time.perf_counter()
        cells = run()
        barrier()
        times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    a, b, c, d = [x[0].cpu().numpy() for x in cells]
    uh = stacked[0].cpu().numpy()
    ok = bool(((a + b) == uh).all() and ((a + c) == nuser).all() and ((a + b + c + d) == len(uni["chrom"])).all())
    if rank == 0:
        print(json.dumps({"n_gpus": world, "F": F, "db_intervals": F * per, "local_db_intervals": sdb.local_intervals,
                          "universe": len(uni["chrom"]), "user": nuser, "build_s": round(tb, 2), "counts_ms": round(dt * 1e3, 3),
                          "identities_hold": ok, "support_sum": int(a.sum())}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
