"""Seeded synthetic genomes and guide sets.

No genome ships with either box and there is no network, so every input is
generated here from a fixed numpy PCG64 seed; the same image runs on both boxes,
so bytes agree.  Formats follow the reference: a genome is the FASTA-order
concatenation of upper-cased chromosome sequences with no separators
(src/genomics/seq_io.cxx:57-63); the reverse text is its reverse complement with
non-ACGT bytes unchanged (src/genomics/sequences.cxx:14-46).
"""
from __future__ import annotations

import numpy as np

SACCER3_LENGTHS = [230218, 813184, 316620, 1531933, 576874, 270161, 1090940, 562643,
                   439888, 745751, 666816, 1078177, 924431, 784333, 1091291, 948066]
CHR1_LENGTH = 248956422
GRCH38_LENGTHS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979,
                  159345973, 145138636, 138394717, 133797422, 135086622, 133275309,
                  114364328, 107043718, 101991189, 90338345, 83257441, 80373285,
                  58617616, 64444167, 46709983, 50818468, 156040895, 57227415]

_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTacgt", b"TGCAtgca"):
    _COMP[_a] = _b


def reverse_complement_bytes(text: np.ndarray) -> np.ndarray:
    """sequences.cxx:14-46 applied to a whole text (seq_io.cxx:65-72)."""
    return _COMP[text[::-1]]


def _fill_bases(text, s, e, seed, cum):
    """bytes [s, e) of the i.i.d. stream: draw i of the PCG64(seed) stream decides base i, so any
    chunking (and any number of threads) gives the same text"""
    bg = np.random.PCG64(seed)
    bg.advance(s)
    u = np.random.Generator(bg).random(e - s)
    idx = (u >= cum[0]).view(np.uint8) + (u >= cum[1]).view(np.uint8) + (u >= cum[2]).view(np.uint8)
    text[s:e] = _ACGT[idx]


_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_genome(lengths, seed=1, probs=(0.29, 0.21, 0.21, 0.29), n_blocks=True, out=None, threads=None):
    """Return (text uint8[sum(lengths)], names, lengths).

    i.i.d. bases; when n_blocks, each chromosome longer than 100 kb gets 10 kb
    telomeric N runs and one centromeric N block of ~1.2 % of its length
    (SURVEY.md section 8d, C2/C3).  Chunks are filled by a thread pool, each from the
    seed's stream advanced to its own offset: the bytes do not depend on the thread count.
    `out`: a preallocated uint8 array (e.g. an np.memmap in /dev/shm shared by the ranks)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    total = int(sum(lengths))
    cum = np.cumsum(np.asarray(probs, dtype=np.float64))
    text = np.empty(total, dtype=np.uint8) if out is None else out
    assert text.shape[0] == total and text.dtype == np.uint8
    chunk = 1 << 24
    spans = [(s, min(total, s + chunk)) for s in range(0, total, chunk)]
    nt = threads or min(16, os.cpu_count() or 1)
    if nt > 1 and len(spans) > 1:
        with ThreadPoolExecutor(nt) as ex:
            list(ex.map(lambda se: _fill_bases(text, se[0], se[1], seed, cum), spans))
    else:
        for s, e in spans:
            _fill_bases(text, s, e, seed, cum)
    off = 0
    for ln in lengths:
        if n_blocks and ln > 100_000:
            tel = 10_000
            text[off:off + tel] = ord("N")
            text[off + ln - tel:off + ln] = ord("N")
            cen = max(1000, int(ln * 0.012))
            c0 = off + ln // 3
            text[c0:c0 + cen] = ord("N")
        off += ln
    names = [f"chr{i + 1}" for i in range(len(lengths))]
    return text, names, [int(x) for x in lengths]


def _mutate(rng, block, div):
    """substitute each base of uint8[n, w] independently with probability div (to one of the other three)"""
    hit = rng.random(block.shape) < div
    if hit.any():
        code = np.searchsorted(_ACGT, block[hit])            # A,C,G,T -> 0..3 (sorted bytes)
        block[hit] = _ACGT[(code + rng.integers(1, 4, size=code.shape[0])) & 3]
    return block


def plant_repeats(text, lengths, seed=1, fraction=0.45, chunk_bytes=1 << 28, sine_div=(0.10, 0.15), sine_copies=None):
    """Overwrite ~`fraction` of an i.i.d. genome with repeat families, both strands (a real assembly is
    about half interspersed repeats; guides drawn from them have 10^4..10^5 near-copies):
      SINE-like   : 300 bp unit, divergence 10-15 % from the unit per copy      (22 % of the repeat bases)
      LINE-like   : 6 kb unit (copies 5'-truncated to 0.5-6 kb), divergence 5 %  (45 %)
      tandem      : arrays of a 2-170 bp motif, 1-4 kb long, divergence 2 %       (8 %)
      segmental   : duplications of 100 kb stretches of the genome itself at 1 %  (25 %)
    At 3.09 Gbp and fraction 0.45 that is ~1.0e6 SINE-like and ~1.6e5 LINE-like copies.  Seeded;
    copies may overlap (later ones win); N blocks are not written over.
    sine_div / sine_copies: the SINE-like family's divergence range and copy number - (0.02, 0.15) and 1.2e6 copies
    make it Alu-like (young copies 2 % from the unit: a guide drawn from it has several 10^5 sites within three
    mismatches, bench.py's hg38alu workload)."""
    rng = np.random.Generator(np.random.PCG64(seed + 77))
    total = int(text.shape[0])
    rep = fraction * total
    ar = np.arange
    keep_n = text == ord("N") if total < (1 << 28) else None   # small genomes: exact; large: N blocks re-applied by the caller

    def place(unit, n_copies, div_lo, div_hi, min_len=None):
        w = unit.shape[0]
        per = max(1, chunk_bytes // w)
        for c0 in range(0, n_copies, per):
            n = min(per, n_copies - c0)
            block = np.tile(unit, (n, 1))
            div = rng.uniform(div_lo, div_hi, size=(n, 1))
            block = _mutate(rng, block, div)
            minus = rng.random(n) < 0.5
            block[minus] = _COMP[block[minus][:, ::-1]]
            starts = rng.integers(0, total - w, size=n)
            if min_len is None:
                text[starts[:, None] + ar(w)[None, :]] = block
            else:   # truncated copies: keep the last `ln` bases of each
                ln = rng.integers(min_len, w + 1, size=n)
                cols = ar(w)[None, :]
                m = cols >= (w - ln)[:, None]
                idx = (starts[:, None] + cols)[m]
                text[idx] = block[m]

    sine = _ACGT[rng.integers(0, 4, size=300)]
    place(sine, int(0.22 * rep / 300) if sine_copies is None else int(sine_copies), sine_div[0], sine_div[1])
    line = _ACGT[rng.integers(0, 4, size=6000)]
    place(line, int(0.45 * rep / 3250), 0.05, 0.05, min_len=500)
    n_tandem = int(0.08 * rep / 2500)
    for _ in range(n_tandem):
        motif = _ACGT[rng.integers(0, 4, size=int(rng.integers(2, 171)))]
        ln = int(rng.integers(1000, 4001))
        arr = np.tile(motif, ln // motif.shape[0] + 1)[:ln][None, :].copy()
        arr = _mutate(rng, arr, 0.02)[0]
        at = int(rng.integers(0, total - ln))
        text[at:at + ln] = arr
    n_seg = int(0.25 * rep / 100_000)
    for _ in range(n_seg):
        src = int(rng.integers(0, total - 100_000))
        dup = _mutate(rng, text[src:src + 100_000][None, :].copy(), 0.01)[0]
        ok = np.isin(dup, _ACGT)
        dup[~ok] = _ACGT[rng.integers(0, 4, size=int((~ok).sum()))]   # a source stretch inside an N block
        if rng.random() < 0.5:
            dup = _COMP[dup[::-1]]
        at = int(rng.integers(0, total - 100_000))
        text[at:at + 100_000] = dup
    if keep_n is not None:
        text[keep_n] = ord("N")
    return text


def make_repeat_genome(lengths, seed=1, probs=(0.29, 0.21, 0.21, 0.29), fraction=0.45, out=None, **family):
    """make_genome + plant_repeats (family: its sine_div / sine_copies), N blocks as make_genome lays them"""
    text, names, lengths = make_genome(lengths, seed=seed, probs=probs, n_blocks=False, out=out)
    if not text.flags.writeable:
        text = text.copy()
    plant_repeats(text, lengths, seed=seed, fraction=fraction, **family)
    off = 0
    for ln in lengths:
        if ln > 100_000:
            tel = 10_000
            text[off:off + tel] = ord("N")
            text[off + ln - tel:off + ln] = ord("N")
            cen = max(1000, int(ln * 0.012))
            c0 = off + ln // 3
            text[c0:c0 + cen] = ord("N")
        off += ln
    return text, names, lengths


def sample_guides(text: np.ndarray, n: int, seed=7, L=20, pam=b"NGG", minus_fraction=0.5):
    """Sample on-target guides: + strand sites whose next P bases match `pam`
    (N = any of ACGT) and - strand sites (reverse complement), protospacer ACGT only.
    Returns (seqs uint8[n,L], pams uint8[n,P] = the pattern, positions, strands).
    Candidates are drawn in bounded chunks and pre-filtered on the fixed PAM bases, so
    millions of guides on a 3 Gbp text stay cheap."""
    rng = np.random.Generator(np.random.PCG64(seed))
    P = len(pam)
    total = text.shape[0]
    seqs = np.empty((n, L), dtype=np.uint8)
    strands = np.empty(n, dtype=np.uint8)
    positions = np.empty(n, dtype=np.int64)
    got = 0
    acgt = np.zeros(256, dtype=bool)
    acgt[list(b"ACGT")] = True
    pam_arr = np.frombuffer(pam, dtype=np.uint8)
    ar = np.arange(L + P)[None, :]
    while got < n:
        m = int(min(max(65536, (n - got) * 32), 8_000_000))
        cand = rng.integers(0, total - (L + P), size=m)
        minus = rng.random(m) < minus_fraction
        ok = np.ones(m, dtype=bool)
        for j in range(P):
            if pam_arr[j] == ord("N"):
                continue
            # + strand: PAM base j sits at c+L+j ; - strand: its complement at c+P-1-j
            at = np.where(minus, cand + (P - 1 - j), cand + L + j)
            want = np.where(minus, _COMP[pam_arr[j]], pam_arr[j])
            ok &= text[at] == want
        idx = np.nonzero(ok)[0]
        if idx.size == 0:
            continue
        c = cand[idx]
        mi = minus[idx]
        win = text[c[:, None] + ar]
        win[mi] = _COMP[win[mi][:, ::-1]]
        good = acgt[win].all(axis=1)
        sel = np.nonzero(good)[0]
        take = min(sel.size, n - got)
        sel = sel[:take]
        seqs[got:got + take] = win[sel, :L]
        strands[got:got + take] = np.where(mi[sel], ord("-"), ord("+"))
        positions[got:got + take] = c[sel]
        got += take
    pams = np.tile(pam_arr, (n, 1))
    return seqs, pams, positions, strands


def write_fasta(path, text: np.ndarray, names, lengths, width=60, lowercase_chr=None):
    with open(path, "wb") as f:
        off = 0
        for i, (nm, ln) in enumerate(zip(names, lengths)):
            f.write(b">" + nm.encode() + b" synthetic\n")
            seq = text[off:off + ln].tobytes()
            if lowercase_chr is not None and i == lowercase_chr:
                seq = seq.lower()
            for s in range(0, ln, width):
                f.write(seq[s:s + width] + b"\n")
            off += ln


def write_kmers_csv(path, ids, seqs, pams, chroms, positions, senses):
    """kmers file: src/genomics/kmer.cxx:12-18 header and columns."""
    with open(path, "w") as f:
        f.write("id,sequence,pam,chromosome,position,sense\n")
        for i in range(len(ids)):
            f.write(f"{ids[i]},{seqs[i]},{pams[i]},{chroms[i]},{positions[i]},{senses[i]}\n")
