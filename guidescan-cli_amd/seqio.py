"""Input formats either side of the hot path.

Host-side parsing only; mirrors the reference's rules so coordinates agree:
  * FASTA -> raw text: lines starting with '>' skipped, other lines whitespace
    trimmed and upper-cased, every remaining byte kept (src/genomics/seq_io.cxx:47-63).
  * genome structure: name = first space-delimited token after '>', length = sum of
    UNTRIMMED line lengths (seq_io.cxx:83-107); `.gs` file = "name\\nlength\\n" per
    chromosome (seq_io.cxx:112-144).
  * kmers CSV: header id,sequence,pam,chromosome,position,sense; fields trimmed of
    ' ' and '\\t', no quoting (src/genomics/kmer.cxx:9-25, include/csv.hpp:1110-1116).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


def parse_fasta(path):
    """-> (text uint8[L], names, lengths)"""
    chunks = []
    names = []
    lengths = []
    with open(path, "rb") as f:
        for raw in f:
            line = raw[:-1] if raw.endswith(b"\n") else raw
            if line.startswith(b">"):
                tok = line[1:].split(b" ")[0]
                names.append(tok.decode())
                lengths.append(0)
                continue
            if lengths:
                lengths[-1] += len(line)  # untrimmed, seq_io.cxx:100-104
            s = line.strip().upper()
            if s:
                chunks.append(s)
    text = np.frombuffer(b"".join(chunks), dtype=np.uint8).copy()
    return text, names, lengths


def read_gs(path):
    names, lengths = [], []
    with open(path) as f:
        lines = [ln.rstrip("\n") for ln in f]
    for i in range(0, len(lines) - 1, 2):
        names.append(lines[i])
        lengths.append(int(lines[i + 1]))
    return names, lengths


@dataclass
class Kmer:
    id: str
    sequence: str
    pam: str
    chromosome: str
    position: int  # 0-based, kmer.cxx:20
    positive: bool


def read_kmers(path):
    out = []
    with open(path) as f:
        header = [h.strip(" \t") for h in f.readline().rstrip("\n").split(",")]
        want = ["id", "sequence", "pam", "chromosome", "position", "sense"]
        idx = [header.index(w) for w in want]
        for line in f:
            line = line.rstrip("\n").rstrip("\r")
            if not line:
                continue
            parts = [p.strip(" \t") for p in line.split(",")]
            v = [parts[i] for i in idx]
            out.append(Kmer(v[0], v[1], v[2], v[3], int(v[4]) - 1, v[5] == "+"))
    return out
