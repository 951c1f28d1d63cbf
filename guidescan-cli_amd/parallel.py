"""Multi-GPU layer: replicate the index, shard the guide batch, no data-path collective.

The reference's only parallelism is a static round-robin split of guides over std::threads
sharing one read-only index (src/guidescan.cxx:226-251); across nodes the manual says
"split the kmers file and concatenate the outputs" (manual/manual.tex:551-583).  Here:
one process per GPU (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" in CPU
tests), every rank holds the whole index in its own HBM, rank r enumerates the contiguous
guide range shard_bounds(n, world)[r], and results are gathered to rank 0 in guide order.
torch.distributed is used for rendezvous, barriers and the final gather only."""
from __future__ import annotations

import time

import numpy as np


def shard_bounds(n: int, world: int):
    """contiguous, balanced: first n % world ranks get one extra guide"""
    base, extra = divmod(n, world)
    bounds = [0]
    for r in range(world):
        bounds.append(bounds[-1] + base + (1 if r < extra else 0))
    return bounds


def enumerate_sharded(enumerate_fn, seqs: np.ndarray, pams: np.ndarray, dist=None, gather=True,
                      **kw):
    """enumerate_fn(seqs, pams, **kw) -> (offsets uint64[n_local+1], hits structured[], stats).
    Every rank passes the FULL batch; each enumerates only its shard.  With gather, rank 0
    returns (offsets[n+1], hits) for the whole batch in guide order, other ranks (None, None)."""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    b = shard_bounds(seqs.shape[0], world)
    lo, hi = b[rank], b[rank + 1]
    offsets, hits, stats = enumerate_fn(seqs[lo:hi], pams[lo:hi], **kw)
    if dist is None or world == 1 or not gather:
        return offsets, hits, stats
    parts = [None] * world if rank == 0 else None
    dist.gather_object((np.asarray(offsets), np.asarray(hits)), parts, dst=0)
    if rank != 0:
        return None, None, stats
    all_off = [np.zeros(1, dtype=np.uint64)]
    all_hits = []
    base = np.uint64(0)
    for off, h in parts:
        all_off.append(off[1:].astype(np.uint64) + base)
        base = base + np.uint64(off[-1])
        all_hits.append(h)
    return np.concatenate(all_off), np.concatenate(all_hits), stats


def timed_steps(step_fn, steps: int, warmup: int, sync_fn, dist=None):
    """bench contract: W untimed steps, then exactly K steps bracketed by barrier+sync on both
    sides; returns the MAX elapsed seconds over ranks."""
    def fence():
        sync_fn()
        if dist is not None and dist.get_world_size() > 1:
            dist.barrier()
            sync_fn()
    for i in range(warmup):
        step_fn(i)
    fence()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        step_fn(i)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None and dist.get_world_size() > 1:
        import torch
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed
