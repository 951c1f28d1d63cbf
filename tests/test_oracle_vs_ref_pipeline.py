"""Pin the oracle's search recursion, set order/dedupe, expansion and CSV/SAM printers against the
REFERENCE'S OWN enumerate pipeline compiled in place: oracle/_ref/gs_ref_enumerate =
include/genomics/{index,process,printer}.hpp + src/genomics/*.cxx + sdsl csa_wt, built by
oracle/Makefile from /root/reference with g++ only (oracle/ref_enumerate.cpp says how).

Two kinds of test, CPU only, skipped when oracle/_ref was never built:
 * the committed toy goldens (tests/golden/toy/ref_*) are reproduced byte for byte by this
   repository's own build of the reference (they were first written by the survey phase's cmake
   build; from here on they are oracle/_ref outputs);
 * on fresh seeded genomes the oracle's whole output file equals the compiled reference's, over
   mismatch budgets 0..4, both formats and modes, alt PAMs, --start, --max-off-targets,
   --threshold and RNA/DNA bulges.
The index files the reference loads are written from the oracle's SA/BWT through the compiled
reference containers (ref_write_index_file; byte-identical to the reference's own `index` output on
the toy, test_oracle_vs_ref.py)."""
import hashlib
import subprocess
import tempfile
from importlib import import_module
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as ol
from test_oracle_golden import RUNS, reference_header

synth = import_module("guidescan-cli_amd.synth")
seqio = import_module("guidescan-cli_amd.seqio")

SHIM = ol.ORACLE_DIR / "_ref" / "gs_ref_enumerate"
ref = ol.ref()
pytestmark = pytest.mark.skipif(ref is None or not SHIM.exists(),
                                reason="oracle/_ref not built (no reference tree)")


def run_shim(prefix, kmers_csv, out, m=3, fmt="csv", complete=True, alt=(), maxo=-1, thr=-1, start=False,
             rna=0, dna=0):
    cmd = [str(SHIM), str(prefix), str(kmers_csv), str(out), fmt, "complete" if complete else "succinct",
           str(m), str(rna), str(dna), str(thr), str(maxo), "1" if start else "0", *alt]
    subprocess.run(cmd, check=True, timeout=900, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return Path(out).read_bytes()


def write_reference_index(oidx, n, prefix, names, lengths):
    """.forward / .reverse / .gs as `guidescan index` leaves them (src/guidescan.cxx:168-175)"""
    for which, strand_text, suffix in (("fwd", oidx.text, ".forward"), ("rev", oidx.rtext, ".reverse")):
        sa = oidx.sa(which)
        t0 = np.concatenate([strand_text, np.zeros(1, dtype=np.uint8)])  # sentinel, construct.hpp:133-135
        bwt = np.ascontiguousarray(t0[(sa.astype(np.int64) - 1) % n])
        tmp = tempfile.NamedTemporaryFile(delete=False, suffix=".sdsl")
        tmp.close()
        h = ref.ref_index_build(bwt.ctypes.data, sa.ctypes.data, n, tmp.name.encode())
        assert ref.ref_write_index_file(h, (str(prefix) + suffix).encode()) == 0
        ref.ref_index_free(h)
        Path(tmp.name).unlink(missing_ok=True)
    Path(str(prefix) + ".gs").write_text("".join(f"{a}\n{b}\n" for a, b in zip(names, lengths)))


def oracle_file(oidx, names, lengths, kmers, fmt="csv", complete=True, **kw):
    opts = ol.make_opts(mismatches=kw.get("m", 3), start=kw.get("start", False), alt_pams=kw.get("alt", ()),
                        max_off_targets=kw.get("maxo", -1), complete=complete, threshold=kw.get("thr", -1),
                        rna_bulges=kw.get("rna", 0), dna_bulges=kw.get("dna", 0))
    out = []
    for k in kmers:
        hits, ctr, raw = oidx.enumerate(k.sequence, k.pam, opts)
        if hits is None:
            continue
        out.append(ol.text_lines(fmt, names, lengths, k.id, k.sequence, k.pam, k.positive, opts, raw))
        ol.lib().gso_free(raw[0])
    header = reference_header(dict(names=names, lengths=lengths), fmt, complete)
    return (header + "".join(out)).encode()


@pytest.fixture(scope="module")
def toy_prefix(toy, tmp_path_factory):
    d = tmp_path_factory.mktemp("toyidx")
    for s in (".forward", ".reverse"):
        (d / ("toy.idx" + s)).write_bytes((toy["dir"] / ("toy.idx" + s)).read_bytes())
    (d / "toy.idx.gs").write_bytes((toy["dir"] / "toy.gs").read_bytes())
    return d / "toy.idx"


@pytest.mark.parametrize("name", sorted(RUNS))
def test_compiled_reference_reproduces_committed_goldens(toy, toy_prefix, tmp_path, name):
    cfg = RUNS[name]
    fmt = cfg.get("fmt", "csv")
    got = run_shim(toy_prefix, toy["dir"] / "kmers.csv", tmp_path / "out", **cfg)
    f = toy["dir"] / f"ref_{name}.{fmt}"
    if f.exists():
        assert got == f.read_bytes()
    else:
        digest, size = (toy["dir"] / f"ref_{name}.{fmt}.sha256").read_text().split()
        assert len(got) == int(size) and hashlib.sha256(got).hexdigest() == digest


def test_compiled_reference_reproduces_config1_goldens(tmp_path):
    """BASELINE config 1 (sacCer3-sized genome regenerated from its seed, 1,000 guides): the
    committed reference files of tests/golden/config1 come out of this repository's own build of
    the reference, from index files written through the compiled reference containers."""
    gold = ol.ROOT / "tests" / "golden" / "config1"
    text, names, lengths = synth.make_genome(synth.SACCER3_LENGTHS, seed=1, probs=(.31, .19, .19, .31))
    oidx = ol.OracleIndex(text)
    try:
        write_reference_index(oidx, text.shape[0] + 1, tmp_path / "g", names, lengths)
    finally:
        oidx.close()
    for name, cfg in (("m1_csv", dict(m=1)), ("m3_csv", dict(m=3)), ("m3_sam", dict(m=3, fmt="sam"))):
        fmt = cfg.get("fmt", "csv")
        got = run_shim(tmp_path / "g", gold / "kmers.csv", tmp_path / "out", **cfg)
        assert got == (gold / f"ref_{name}.{fmt}").read_bytes(), name


def random_case(seed):
    """a small genome with repeats, N runs and skewed composition + guides around its sites"""
    rng = np.random.Generator(np.random.PCG64(seed))
    lengths = [int(x) for x in rng.integers(3000, 12000, size=int(rng.integers(1, 4)))]
    p = rng.dirichlet([6, 4, 4, 6])
    text, names, lengths = synth.make_genome(lengths, seed=seed, probs=tuple(p), n_blocks=False)
    text = text.copy()
    n = text.shape[0]
    # repeat family with a few substitutions/indels so that every distance class is populated
    fam = bytearray(text[100:123].tobytes())
    fam[21:23] = b"GG"
    for _ in range(int(rng.integers(25, 45))):
        s = bytearray(fam)
        for i in rng.choice(20, size=int(rng.integers(0, 5)), replace=False):
            s[i] = rng.choice([c for c in b"ACGT" if c != s[i]])
        if rng.random() < 0.3:  # another PAM: NAG / NGA / none of them
            s[20:23] = bytes(rng.choice(list(b"ACGT"), size=3).astype(np.uint8))
        kind = int(rng.integers(0, 6))
        if kind == 1:  # deletion in the genome copy (RNA bulge)
            del s[int(rng.integers(2, 18))]
            s.append(ord("G"))
        elif kind == 2:  # insertion in the genome copy (DNA bulge)
            s.insert(int(rng.integers(2, 18)), int(rng.choice(list(b"ACGT"))))
        b = bytes(s)
        if rng.random() < 0.5:
            b = synth.reverse_complement_bytes(np.frombuffer(b, dtype=np.uint8)).tobytes()
        pos = int(rng.integers(0, n - len(b)))
        text[pos:pos + len(b)] = np.frombuffer(b, dtype=np.uint8)
    for _ in range(int(rng.integers(0, 3))):
        a = int(rng.integers(0, n - 50))
        text[a:a + int(rng.integers(1, 40))] = ord("N")
    text[100:123] = np.frombuffer(bytes(fam), dtype=np.uint8)
    seqs, pams, positions, strands = synth.sample_guides(text, 10, seed=seed + 1)
    rows = [(f"s{i}", seqs[i].tobytes().decode(), "NGG", "+") for i in range(len(seqs))]
    rows.append(("fam", bytes(fam[:20]).decode(), "NGG", "+"))
    rows.append(("famrc", synth.reverse_complement_bytes(np.frombuffer(bytes(fam[:20]), dtype=np.uint8))
                 .tobytes().decode(), "NGG", "-"))
    rows.append(("nopam", bytes(fam[:20]).decode(), "", "+"))
    rows.append(("absent", "ACGTTGCAACGTTGCAACGT", "NGG", "+"))
    return text, names, lengths, rows


OPTION_SETS = [
    dict(m=0), dict(m=1, fmt="sam"), dict(m=2, complete=False), dict(m=3), dict(m=4, fmt="sam", complete=False),
    dict(m=3, alt=("NAG",)), dict(m=2, alt=("NAG", "NGA"), fmt="sam"), dict(m=3, maxo=1),
    dict(m=2, thr=1), dict(m=2, thr=0, fmt="sam"), dict(m=2, start=True), dict(m=3, start=True, alt=("NAG",)),
    dict(m=1, rna=1), dict(m=1, dna=1), dict(m=1, rna=1, dna=1, fmt="sam"), dict(m=0, rna=2, dna=1),
    dict(m=1, dna=2, start=True), dict(m=2, rna=1, dna=1, alt=("NAG",), maxo=3),
]


@pytest.mark.parametrize("seed", [101, 102, 103, 104, 105, 106, 107, 108])
def test_oracle_equals_compiled_reference_on_random_genomes(seed, tmp_path):
    text, names, lengths, rows = random_case(seed)
    prefix = tmp_path / "g.idx"
    kcsv = tmp_path / "kmers.csv"
    synth.write_kmers_csv(kcsv, [r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows],
                          [names[0]] * len(rows), [1] * len(rows), [r[3] for r in rows])
    kmers = seqio.read_kmers(kcsv)
    oidx = ol.OracleIndex(text)
    try:
        write_reference_index(oidx, text.shape[0] + 1, prefix, names, lengths)
        for cfg in OPTION_SETS:
            fmt = cfg.get("fmt", "csv")
            complete = cfg.get("complete", True)
            kw = {k: v for k, v in cfg.items() if k not in ("fmt", "complete")}
            want = run_shim(prefix, kcsv, tmp_path / "out", fmt=fmt, complete=complete, **kw)
            got = oracle_file(oidx, names, lengths, kmers, fmt=fmt, complete=complete, **kw)
            assert got == want, (seed, cfg)
    finally:
        oidx.close()
