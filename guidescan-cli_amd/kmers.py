"""Candidate-guide generation: the kmers CSV that `enumerate` reads.

Vectorised restatement of the reference's helper script (scripts/generate_kmers.py:49-125):
every occurrence of each concrete expansion of the PAM on the + strand (protospacer = the k
bases before it; position = protospacer start, 1-based) in PAM-expansion order, then every
occurrence of each reverse-complemented PAM on the - strand (protospacer = reverse complement
of the k bases after it; position = PAM start, 1-based); kmers containing non-ACGT are dropped;
id = prefix + chromosome:position:sense; the `pam` column is the PATTERN (e.g. NGG).
"""
from __future__ import annotations

import numpy as np

_NUCS = "ACTG"  # expansion order of the reference (scripts/generate_kmers.py:49)
_COMP = {"A": "T", "T": "A", "C": "G", "G": "C"}


def pam_set(pam: str):
    """scripts/generate_kmers.py:55-69: breadth-first replacement of the first N"""
    stack = [pam]
    while any("N" in p for p in stack):
        p = stack.pop(0)
        if "N" not in p:
            stack.append(p)
            continue
        for n in _NUCS:
            stack.append(p.replace("N", n, 1))
    return stack


def _revcom(s: str) -> str:
    return "".join(_COMP[c] for c in reversed(s))


def _find_all(chrm: np.ndarray, pat: bytes) -> np.ndarray:
    """start offsets of every (overlapping) occurrence of pat"""
    n, m = chrm.shape[0], len(pat)
    if n < m:
        return np.empty(0, dtype=np.int64)
    ok = np.ones(n - m + 1, dtype=bool)
    for j, c in enumerate(pat):
        ok &= chrm[j:n - m + 1 + j] == c
    return np.nonzero(ok)[0].astype(np.int64)


_ACGT = np.zeros(256, dtype=bool)
_ACGT[list(b"ACGT")] = True
_RC = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGT", b"TGCA"):
    _RC[_a] = _b


def find_all_kmers(chrm, pam="NGG", k=20, start=False):
    """-> list of (sequence str, position 1-based, sense) in the reference script's order.
    chrm: bytes / uint8 array of ONE chromosome (any case)."""
    c = np.frombuffer(bytes(chrm), dtype=np.uint8) if not isinstance(chrm, np.ndarray) else chrm
    c = np.frombuffer(c.tobytes().upper(), dtype=np.uint8)
    n, P = c.shape[0], len(pam)
    end = not start
    out = []
    ar = np.arange(k)
    for forward, pats in ((True, pam_set(pam)), (False, [_revcom(p) for p in pam_set(pam)])):
        for p in pats:
            idx = _find_all(c, p.encode())
            if idx.size == 0:
                continue
            # scripts/generate_kmers.py:79-93: which side of the PAM the protospacer is on
            before = (end and forward) or (not end and not forward)
            if before:
                s0 = idx - k
                pos = idx - k
            else:
                s0 = idx + P
                pos = idx
            keep = (pos >= 0) & (s0 >= 0) & (s0 + k <= n)
            s0, pos = s0[keep], pos[keep]
            if s0.size == 0:
                continue
            win = c[s0[:, None] + ar[None, :]]
            good = _ACGT[win].all(axis=1)
            win, pos = win[good], pos[good]
            if not forward:
                win = _RC[win[:, ::-1]]
            sense = "+" if forward else "-"
            for w, q in zip(win, pos):
                out.append((w.tobytes().decode(), int(q) + 1, sense))
    return out


def find_all_kmers_device(chrm, pam="NGG", k=20, start=False, device=0):
    """find_all_kmers through the GPU scan (gs_kmers_generate, csrc/gs_kmers.hip); same records,
    same order.  Raises when the HIP library or a GPU is missing (no CPU fallback here)."""
    from importlib import import_module
    api = import_module("guidescan-cli_amd.api")
    km = api.generate_kmers(chrm, pam, k, start, device)
    try:
        seqs, _, pos, sense = km.to_host()
    finally:
        km.close()
    return [(seqs[i].tobytes().decode(), int(pos[i]), chr(sense[i])) for i in range(seqs.shape[0])]


def write_kmers_csv(fh, records, pam="NGG", k=20, start=False, prefix="", min_chr_length=0, device=None):
    """records: iterable of (name, sequence bytes).  Writes the kmers CSV (header included).
    device: GPU ordinal to scan on (None = the numpy restatement)."""
    fh.write("id,sequence,pam,chromosome,position,sense\n")
    n = 0
    for name, seq in records:
        if len(seq) < min_chr_length:
            continue
        found = find_all_kmers(seq, pam, k, start) if device is None else \
            find_all_kmers_device(seq, pam, k, start, device)
        for kmer, pos, sense in found:
            fh.write(f"{prefix}{name}:{pos}:{sense},{kmer},{pam},{name},{pos},{sense}\n")
            n += 1
    return n


def fasta_records(path):
    """(name, sequence) per record; name = first whitespace-delimited token after '>'"""
    name, chunks = None, []
    with open(path, "rb") as f:
        for raw in f:
            line = raw.strip()
            if line.startswith(b">"):
                if name is not None:
                    yield name, b"".join(chunks)
                tok = line[1:].split()
                name, chunks = (tok[0].decode() if tok else ""), []
            elif name is not None:
                chunks.append(line)
    if name is not None:
        yield name, b"".join(chunks)
