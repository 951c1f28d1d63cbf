"""GS_TOOLS_STUB=1: rehearsal of the multi-rank plumbing of tools/genomewide_enumerate.py and tools/config5_stream.py
WITHOUT a GPU (tests/test_distributed_gloo.py): torch.distributed.run -> gloo rendezvous on 127.0.0.1 -> rank r takes
the batches b with b % world == r -> totals summed, times MAX-reduced -> one JSON line from rank 0.

The stand-ins below answer with numbers that depend on the guides alone (a hash of each guide's bytes), so the totals of
a two-rank run must equal a one-rank run's whatever the partition.  They search nothing: a line printed under the stub
says "stub": true and is not a measurement.  Nothing here is reachable without the environment variable."""
import ctypes as C
import zlib

import numpy as np


def host_bytes(ptr, n):
    return np.ctypeslib.as_array((C.c_uint8 * n).from_address(ptr)) if n else np.zeros(0, np.uint8)


def guide_hits(seqs_ptr, n, L):
    """a made-up hit count per guide: 1 + crc32(guide) % 7"""
    rows = host_bytes(seqs_ptr, n * L).reshape(n, L)
    return np.array([1 + zlib.crc32(r.tobytes()) % 7 for r in rows], dtype=np.int64)


class StubKmers:
    def __init__(self, seqs, pams):
        self.seqs, self.pams, self.n = seqs, pams, seqs.shape[0]
        self.seqs_ptr, self.pams_ptr = seqs.ctypes.data, pams.ctypes.data

    def close(self):
        self.seqs = self.pams = None


def generate_kmers(chrm, L=20):
    """NGG sites of the + strand whose 20-mer is all A,C,G,T (the device generator also takes the - strand)"""
    G = ord("G")
    n = chrm.shape[0] - (L + 3) + 1
    if n <= 0:
        return StubKmers(np.zeros((0, L), np.uint8), np.zeros((0, 3), np.uint8))
    acgt = np.zeros(256, dtype=bool)
    acgt[list(b"ACGT")] = True
    bad = np.concatenate([[0], np.cumsum(~acgt[chrm])])
    at = np.nonzero((chrm[L + 1:L + 1 + n] == G) & (chrm[L + 2:L + 2 + n] == G) & (bad[L + 3:L + 3 + n] - bad[:n] == 0))[0]
    seqs = np.ascontiguousarray(chrm[at[:, None] + np.arange(L)[None, :]])
    pams = np.ascontiguousarray(chrm[at[:, None] + L + np.arange(3)[None, :]])
    return StubKmers(seqs, pams)


class StubIndex:
    device_bytes = 0

    def __init__(self, text):
        self.n = int(text.shape[0])
        self._last = None

    def enumerate_device(self, seqs_ptr, n, L, pams_ptr, P, mismatches=3):
        h = guide_hits(seqs_ptr, n, L)
        self._last = h
        return 0, 0, {"n_hits": int(h.sum()), "ms_search": 0.0}

    def score_device(self, gs, seqs_ptr, n, L, P, d_off, d_hits, cfd_ptr, spec_ptr):
        spec = (1.0 / self._last).astype(np.float32)
        C.memmove(spec_ptr, spec.ctypes.data, 4 * n)

    def last_counters(self):
        return {"slots_per_item": 0, "guides_redone": 0, "overflow_from_arena": 0, "ordered_device_wide": 0, "ordered_in_tiles": False}

    def close(self):
        pass
