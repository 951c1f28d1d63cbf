"""Multi-GPU layer: replicate the index, deal the guide batch, no data-path collective.

The reference's only parallelism is a round-robin deal of guides over std::threads sharing one
read-only index (src/guidescan.cxx:226-251: guide i goes to thread i mod n, because a contiguous
split would hand one thread a repeat-dense stretch); across nodes the manual says "split the kmers
file and concatenate the outputs" (manual/manual.tex:551-583).  Here: one process per GPU
(torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" in CPU tests), every rank holds the whole
index in its own HBM, and the guides of a job are dealt in CHUNKS of consecutive guides:

  * weak scaling (bench.py's default): every rank has its own batch, nothing is dealt;
  * strong scaling, static: rank r takes the contiguous range shard_bounds(n, world)[r] - one
    enumerate call per rank, but the slowest shard sets the job's time;
  * strong scaling, dealt (ChunkDealer): chunks of ~64 k guides are handed out from ONE shared counter
    (the rendezvous store's atomic add: a TCP round trip per chunk, no collective), so a rank that
    drew repeat-dense chunks simply draws fewer - what the reference's round-robin deal is for.

Results stay per rank: every rank writes its chunks' hit lists to a file of its own with a small
index (write_chunk / merge_chunk_files: the manual's "concatenate the outputs", in guide order) -
a gather of pickled hit lists through the rendezvous, the first form, cannot carry config 5's
1.6 x 10^11 bytes of hits.  torch.distributed is used for rendezvous, barriers, the shared counter
and the MAX / per-rank reductions of elapsed time only."""
from __future__ import annotations

import struct
import time
from pathlib import Path

import numpy as np


def shard_bounds(n: int, world: int):
    """contiguous, balanced: first n % world ranks get one extra guide"""
    base, extra = divmod(n, world)
    bounds = [0]
    for r in range(world):
        bounds.append(bounds[-1] + base + (1 if r < extra else 0))
    return bounds


def chunk_bounds(n: int, chunk: int):
    """[lo, hi) of every chunk of `chunk` consecutive guides (the last one shorter)"""
    chunk = max(1, int(chunk))
    return [(lo, min(n, lo + chunk)) for lo in range(0, n, chunk)]


class ChunkDealer:
    """Chunks of a job handed out from one shared counter.  `key` names the job (a step of the bench,
    a kmers file); every rank calls take() until it returns None.  With torch.distributed the counter
    lives in the process group's store (TCPStore.add is atomic); without it, in this process."""

    # how many dealers this process has made per key: every rank makes them in the same order (one program, many ranks),
    # so the count is the same number everywhere and a key used again - the same kmers file twice, a retry after a failure -
    # gets a counter of its own in the store instead of finding the last job's, already past its chunks
    _generation = {}

    def __init__(self, n_chunks: int, key: str, dist=None):
        self.n_chunks, self.key, self.dist = int(n_chunks), key, dist
        self._local = 0
        self._taken = 0
        self.store = None
        self.world = 1
        gen = ChunkDealer._generation.get(key, 0)
        ChunkDealer._generation[key] = gen + 1
        self.store_key = f"gs_deal/{key}/{gen}"
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            from torch.distributed import distributed_c10d
            self.store = distributed_c10d._get_default_store()
            self.world = dist.get_world_size()

    def take(self):
        if self.store is not None:
            c = int(self.store.add(self.store_key, 1)) - 1
            # a rank's FIRST draw can be past the chunks only by the draws of the other ranks that found the job finished:
            # anything beyond is a counter that was not this job's
            if self._taken == 0 and self.n_chunks > 0 and c >= self.n_chunks + self.world:
                raise RuntimeError(f"chunk counter {self.store_key!r} already stood at {c} before this rank's first draw "
                                   f"({self.n_chunks} chunks, {self.world} ranks): it belongs to another job")
        else:
            c = self._local
            self._local += 1
        self._taken += 1
        return c if c < self.n_chunks else None


def deal_chunks(process_chunk, n: int, chunk: int, key: str, dist=None):
    """every rank: draw chunks until none is left, process_chunk(c, lo, hi) for each; returns
    (chunks this rank took, seconds it was busy)"""
    bounds = chunk_bounds(n, chunk)
    dealer = ChunkDealer(len(bounds), key, dist)
    mine, busy = [], 0.0
    while True:
        c = dealer.take()
        if c is None:
            break
        t0 = time.perf_counter()
        process_chunk(c, *bounds[c])
        busy += time.perf_counter() - t0
        mine.append(c)
    return mine, busy


def imbalance(per_rank_seconds):
    """(max - mean) / max of the ranks' busy times: the share of the job's time its slowest rank adds"""
    v = [float(x) for x in per_rank_seconds]
    return (max(v) - sum(v) / len(v)) / max(v) if v and max(v) > 0 else 0.0


# ---- per-rank result files ---------------------------------------------------------------------------------
_MAGIC = b"GSCHUNK1"


def write_chunk(fh, chunk_id: int, lo: int, offsets: np.ndarray, hits: np.ndarray):
    """append one chunk's CSR hit list to a rank's file: header {chunk, first guide, guides, hits}, the offsets
    (relative to the chunk's first hit), the 16-byte hit records"""
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    hits = np.ascontiguousarray(hits)
    fh.write(_MAGIC + struct.pack("<QQQQ", chunk_id, lo, offsets.shape[0] - 1, hits.shape[0]))
    fh.write(offsets.tobytes())
    fh.write(hits.tobytes())


def read_chunks(path, hit_dtype):
    """the chunks of one rank's file, in file order: (chunk, first guide, offsets, hits)"""
    with open(path, "rb") as fh:
        while True:
            head = fh.read(8 + 32)
            if not head:
                return
            assert head[:8] == _MAGIC, path
            c, lo, n, nh = struct.unpack("<QQQQ", head[8:])
            off = np.frombuffer(fh.read(8 * (n + 1)), dtype=np.uint64)
            hits = np.frombuffer(fh.read(nh * np.dtype(hit_dtype).itemsize), dtype=hit_dtype)
            yield c, lo, off, hits


def merge_chunk_files(paths, n_guides: int, hit_dtype):
    """the whole job's (offsets[n+1], hits) in guide order from the ranks' files (manual.tex:551-583: the outputs
    concatenated) - for tests and small jobs; a large job streams the files chunk by chunk instead"""
    chunks = sorted((c for p in paths for c in read_chunks(p, hit_dtype)), key=lambda t: t[1])
    offsets = np.zeros(n_guides + 1, dtype=np.uint64)
    parts, base, at = [], 0, 0
    for _, lo, off, hits in chunks:
        assert lo == at, "chunks do not tile the guide range"
        n = off.shape[0] - 1
        offsets[lo + 1:lo + n + 1] = off[1:] + np.uint64(base)
        base += int(off[-1])
        at = lo + n
        parts.append(hits)
    assert at == n_guides
    return offsets, (np.concatenate(parts) if parts else np.zeros(0, dtype=hit_dtype))


def enumerate_dealt(enumerate_fn, seqs: np.ndarray, pams: np.ndarray, out_dir, chunk: int, key: str, dist=None, **kw):
    """Every rank passes the FULL batch and enumerates the chunks it draws; each chunk's hit list goes to the rank's own
    file under out_dir.  Returns (path of this rank's file, chunks taken, busy seconds)."""
    rank = dist.get_rank() if dist is not None else 0
    path = Path(out_dir) / f"hits.rank{rank}.gschunks"
    with open(path, "wb") as fh:
        def one(c, lo, hi):
            offsets, hits, _ = enumerate_fn(seqs[lo:hi], pams[lo:hi], **kw)
            write_chunk(fh, c, lo, offsets, hits)
        mine, busy = deal_chunks(one, seqs.shape[0], chunk, key, dist)
    return path, mine, busy


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


def gather_floats(value: float, dist=None):
    """every rank's value on every rank (a tensor all_gather: no pickling)"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


__all__ = ["shard_bounds", "chunk_bounds", "ChunkDealer", "deal_chunks", "imbalance", "write_chunk", "read_chunks",
           "merge_chunk_files", "enumerate_dealt", "timed_steps", "gather_floats"]
