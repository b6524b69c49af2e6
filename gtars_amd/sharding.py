"""Multi-GPU sharding of the hot path (one process per GPU, torch.distributed; "nccl" is RCCL on ROCm).

* Tokenize / count / find: every query is independent and the universe index is ~1 MB, so the index is
  REPLICATED and the query batch is cut into ``world`` contiguous index ranges.  Rank r's output is exactly
  the slice of the global CSR that belongs to its range, so concatenation in rank order reproduces the
  single-GPU result; no collective is needed unless one consumer wants the whole batch -- then
  ``all_gather_csr`` (C2: all-gather of lengths, then of padded payloads).
* IGD / LOLA counts: per-file hit vectors are sums over queries, so ranks count their query range against a
  replicated database and ``all_reduce_hits`` (C1: one SUM all-reduce of F u64 values, ~16 KB at F = 2,000)
  gives the global vector on every rank.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) of ``n`` items for ``rank``; ranges tile [0, n) in rank order."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_reduce_hits(hits: np.ndarray, group=None, device=None) -> np.ndarray:
    """C1: element-wise SUM of per-file hit vectors over ranks (u64 carried as int64: counts < 2^63)."""
    import torch
    import torch.distributed as dist

    t = torch.from_numpy(np.ascontiguousarray(hits).astype(np.int64))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy().astype(np.uint64)


def all_gather_csr(offsets: np.ndarray, ids: np.ndarray, group=None, device=None) -> Tuple[np.ndarray, np.ndarray]:
    """C2: assemble the global CSR (offsets u64[N+1], ids u32[H]) from per-rank CSRs of consecutive query ranges."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    counts = np.diff(offsets.astype(np.int64))
    meta = torch.tensor([len(counts), len(ids)], dtype=torch.int64, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    nq = [int(m[0]) for m in metas]
    nh = [int(m[1]) for m in metas]

    def gather(arr: np.ndarray, lens: List[int], dtype) -> np.ndarray:
        pad = max(max(lens), 1)
        buf = torch.zeros(pad, dtype=dtype, device=device)
        if len(arr):
            buf[: len(arr)] = torch.from_numpy(arr.astype(np.int64)).to(dtype).to(buf.device)
        bufs = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(bufs, buf, group=group)
        return np.concatenate([b[:l].cpu().numpy() for b, l in zip(bufs, lens)]) if sum(lens) else np.zeros(0, np.int64)

    all_counts = gather(counts, nq, torch.int64)
    all_ids = gather(ids.astype(np.int64), nh, torch.int64)
    goff = np.zeros(len(all_counts) + 1, dtype=np.uint64)
    np.cumsum(all_counts, out=goff[1:])
    return goff, all_ids.astype(np.uint32)
