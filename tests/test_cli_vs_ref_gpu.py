"""Product vs the reference itself, on the GPU box: the C++ host (`guidescan enumerate`, HIP path
through the C-ABI) must write the same file as oracle/_ref/gs_ref_enumerate — the reference's own
index.hpp / process.hpp / printer.hpp compiled in place by oracle/Makefile — on fresh seeded
genomes with planted repeat families, N runs and indel copies, over the option sets of
test_oracle_vs_ref_pipeline.py (mismatches 0..4, CSV/SAM, succinct, alt PAMs, --start,
--max-off-targets, --threshold, RNA/DNA bulges).  The prebuilt oracle/_ref binaries travel with the
snapshot; nothing reads /root/reference here.  GPU only."""
import subprocess

import numpy as np

import pytest

import oracle_lib as ol
import test_oracle_vs_ref_pipeline as pipe

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(pipe.ref is None or not pipe.SHIM.exists(), reason="oracle/_ref not built")]
CLI = ol.ROOT / "guidescan-cli_amd" / "bin" / "guidescan"


def cli_args(m=3, fmt="csv", complete=True, alt=(), maxo=-1, thr=-1, start=False, rna=0, dna=0):
    a = ["-m", str(m), "--format", fmt, "--mode", "complete" if complete else "succinct"]
    for p in alt:
        a += ["-a", p]
    if maxo >= 0:
        a += ["--max-off-targets", str(maxo)]
    if thr >= 0:
        a += ["-t", str(thr)]
    if start:
        a.append("--start")
    if rna:
        a += ["--rna-bulges", str(rna)]
    if dna:
        a += ["--dna-bulges", str(dna)]
    return a


@pytest.mark.parametrize("seed", [201, 202, 203, 204])
def test_cli_equals_compiled_reference_on_random_genomes(seed, tmp_path):
    text, names, lengths, rows = pipe.random_case(seed)
    kcsv = tmp_path / "kmers.csv"
    pipe.synth.write_kmers_csv(kcsv, [r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows],
                               [names[0]] * len(rows), [1] * len(rows), [r[3] for r in rows])
    # reference side: SDSL index files written through the compiled reference containers
    oidx = ol.OracleIndex(text)
    try:
        pipe.write_reference_index(oidx, text.shape[0] + 1, tmp_path / "r.idx", names, lengths)
    finally:
        oidx.close()
    # product side: genome text + structure, index built on the GPU by `enumerate` itself
    text.tofile(tmp_path / "g.dna")
    (tmp_path / "g.gs").write_text("".join(f"{a}\n{b}\n" for a, b in zip(names, lengths)))
    for cfg in pipe.OPTION_SETS:
        want = pipe.run_shim(tmp_path / "r.idx", kcsv, tmp_path / "want", **cfg)
        out = tmp_path / "got"
        subprocess.run([str(CLI), "enumerate", str(tmp_path / "g"), "-f", str(kcsv), "-o", str(out), "-n", "1"]
                       + cli_args(**cfg), check=True, timeout=120)
        assert out.read_bytes() == want, (seed, cfg)


def test_cli_reading_reference_index_equals_compiled_reference(tmp_path):
    """same, with the product importing the reference's own index files (gs_index_open_sdsl)"""
    text, names, lengths, rows = pipe.random_case(301)
    kcsv = tmp_path / "kmers.csv"
    pipe.synth.write_kmers_csv(kcsv, [r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows],
                               [names[0]] * len(rows), [1] * len(rows), [r[3] for r in rows])
    oidx = ol.OracleIndex(text)
    try:
        pipe.write_reference_index(oidx, text.shape[0] + 1, tmp_path / "r.idx", names, lengths)
    finally:
        oidx.close()
    for cfg in (dict(m=3), dict(m=2, fmt="sam", alt=("NAG",)), dict(m=1, rna=1, dna=1)):
        want = pipe.run_shim(tmp_path / "r.idx", kcsv, tmp_path / "want", **cfg)
        out = tmp_path / "got"
        subprocess.run([str(CLI), "enumerate", str(tmp_path / "r.idx"), "-f", str(kcsv), "-o", str(out), "-n", "1"]
                       + cli_args(**cfg), check=True, timeout=120)
        assert out.read_bytes() == want, cfg


def test_cli_accepts_what_the_reference_accepts(tmp_path):
    """a kmers file holding guides with N / lower case / IUPAC symbols next to ordinary ones, five alt
    PAMs (one of them with a literal the genome lacks), many small batches: the CLI answers every
    row - the odd guides through the general path, the others through the fast path, four PAM
    patterns per pass - and its files equal the compiled reference's byte for byte"""
    text, names, lengths, rows = pipe.random_case(501)
    t = text
    isn = t == ord("N")
    at = int(np.nonzero(isn)[0][0]) if isn.any() else 100
    odd = ["ACGTNCGTACGTACGTACGT", "acgtACGTACGTACGTACGT", "ACGTACGTACGTACGTACGR",
           t[at - 7:at + 13].tobytes().decode(), rows[0][1][:9] + "N" + rows[0][1][10:]]
    L = len(rows[0][1])
    odd = [g for g in odd if len(g) == L]
    rows = rows[:10] + [(f"odd{i}", g, rows[0][2], "+") for i, g in enumerate(odd)] + rows[10:]
    kcsv = tmp_path / "kmers.csv"
    pipe.synth.write_kmers_csv(kcsv, [r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows],
                               [names[0]] * len(rows), [1] * len(rows), [r[3] for r in rows])
    oidx = ol.OracleIndex(text)
    try:
        pipe.write_reference_index(oidx, text.shape[0] + 1, tmp_path / "r.idx", names, lengths)
    finally:
        oidx.close()
    text.tofile(tmp_path / "g.dna")
    (tmp_path / "g.gs").write_text("".join(f"{a}\n{b}\n" for a, b in zip(names, lengths)))
    P = len(rows[0][2])
    alts = [p for p in ("NAG", "NGA", "RGG", "NGT", "NCG") if len(p) == P]
    for cfg in (dict(m=2, alt=tuple(alts)), dict(m=3, fmt="sam", alt=tuple(alts)), dict(m=1, start=True, alt=tuple(alts[:4])),
                dict(m=2, thr=1, alt=("NAG",)), dict(m=2, maxo=2, complete=False),
                # the threshold filter counts per PAM pattern (process.hpp:25-27): with the guides' own
                # pattern given again as an alt PAM every site counts twice and every guide is dropped
                dict(m=2, thr=1, alt=("NGG",)), dict(m=3, thr=2, alt=("NGG", "NAG"), fmt="sam")):
        want = pipe.run_shim(tmp_path / "r.idx", kcsv, tmp_path / "want", **cfg)
        out = tmp_path / "got"
        subprocess.run([str(CLI), "enumerate", str(tmp_path / "g"), "-f", str(kcsv), "-o", str(out), "-n", "2",
                        "--batch-size", "7"] + cli_args(**cfg), check=True, timeout=300)
        assert out.read_bytes() == want, cfg


def test_cli_alt_pams_of_other_lengths_equal_the_compiled_reference(tmp_path):
    """`-a` patterns shorter or longer than the guides' PAM: the reference searches each at its own length
    next to the guides' own (process.hpp:51-56), so rows of one guide carry match sequences of different
    lengths; the CLI routes such batches through the general path (gs_enumerate_general_pams) and writes the
    same files - CSV and SAM, with --threshold (counted per pattern), --start, a bulge, --max-off-targets"""
    text, names, lengths, rows = pipe.random_case(601)
    kcsv = tmp_path / "kmers.csv"
    pipe.synth.write_kmers_csv(kcsv, [r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows],
                               [names[0]] * len(rows), [1] * len(rows), [r[3] for r in rows])
    oidx = ol.OracleIndex(text)
    try:
        pipe.write_reference_index(oidx, text.shape[0] + 1, tmp_path / "r.idx", names, lengths)
    finally:
        oidx.close()
    text.tofile(tmp_path / "g.dna")
    (tmp_path / "g.gs").write_text("".join(f"{a}\n{b}\n" for a, b in zip(names, lengths)))
    for cfg in (dict(m=2, alt=("NG",)), dict(m=2, alt=("NGGN", "NAG")), dict(m=3, fmt="sam", alt=("G", "NNGRRT")),
                dict(m=1, start=True, alt=("TTTN",)), dict(m=2, thr=1, alt=("NG",)), dict(m=2, maxo=3, alt=("NGGNG",), complete=False),
                dict(m=1, rna=1, alt=("NGAN",))):
        want = pipe.run_shim(tmp_path / "r.idx", kcsv, tmp_path / "want", **cfg)
        out = tmp_path / "got"
        subprocess.run([str(CLI), "enumerate", str(tmp_path / "g"), "-f", str(kcsv), "-o", str(out), "-n", "2",
                        "--batch-size", "9"] + cli_args(**cfg), check=True, timeout=300)
        assert out.read_bytes() == want, cfg


def test_cli_guides_beyond_the_fast_key_equal_the_compiled_reference(tmp_path):
    """23-mers with a four-symbol PAM at the 5' end (Cas12a: TTTN + --start): 2L + 3P = 58 bits of match
    sequence do not fit the fast path's 52-bit key, so the CLI sends the batch through the general path; the
    files equal the compiled reference's.  Guides are read off the genome behind TTT sites, so they hit."""
    text, names, lengths, _ = pipe.random_case(701)
    t = text.tobytes()
    rows, at = [], 0
    while len(rows) < 10:
        at = t.find(b"TTT", at + 1)
        assert at >= 0
        w = t[at:at + 27]
        if len(w) == 27 and set(w) <= set(b"ACGT"):
            rows.append((f"c{len(rows)}", w[4:].decode(), "TTTN", "+"))
            at += 40
    kcsv = tmp_path / "kmers.csv"
    pipe.synth.write_kmers_csv(kcsv, [r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows],
                               [names[0]] * len(rows), [1] * len(rows), [r[3] for r in rows])
    oidx = ol.OracleIndex(text)
    try:
        pipe.write_reference_index(oidx, text.shape[0] + 1, tmp_path / "r.idx", names, lengths)
    finally:
        oidx.close()
    text.tofile(tmp_path / "g.dna")
    (tmp_path / "g.gs").write_text("".join(f"{a}\n{b}\n" for a, b in zip(names, lengths)))
    for cfg in (dict(m=2, start=True), dict(m=3, start=True, fmt="sam"), dict(m=1, start=True, alt=("TTN",))):
        want = pipe.run_shim(tmp_path / "r.idx", kcsv, tmp_path / "want", **cfg)
        assert want.count(b"\n") > len(rows)
        out = tmp_path / "got"
        subprocess.run([str(CLI), "enumerate", str(tmp_path / "g"), "-f", str(kcsv), "-o", str(out), "-n", "2",
                        "--batch-size", "4"] + cli_args(**cfg), check=True, timeout=300)
        assert out.read_bytes() == want, cfg
