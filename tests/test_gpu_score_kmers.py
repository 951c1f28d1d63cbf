"""GPU parity tests of the two device steps either side of the search: CFD/specificity scoring
(printer.hpp:98-170, 251-297) and candidate-guide generation (scripts/generate_kmers.py:70-118).

Scoring is checked bit for bit: every hit's CFD against the oracle's calculate_cfd, every guide's
specificity against the oracle's CSV/SAM lines AND against the files the reference binary itself
wrote (tests/golden/toy/ref_*).  Run on the GPU box with `pytest -m gpu`."""
import ctypes as C
from importlib import import_module

import numpy as np
import pytest

import oracle_lib as ol
from kmers_golden import check_properties, expected_rows, golden_cases

api = import_module("guidescan-cli_amd.api")
synth = import_module("guidescan-cli_amd.synth")
kmers = import_module("guidescan-cli_amd.kmers")

pytestmark = pytest.mark.gpu

_COMP = str.maketrans("ACGTacgt", "TGCAtgca")


def fmt_f(x) -> str:
    """std::to_string(float): printf("%f") of the value widened to double"""
    return "%f" % float(np.float32(x))


def golden_specificities(path, sam):
    """id -> specificity string of a reference output file"""
    out = {}
    with open(path) as f:
        for line in f:
            if sam:
                if line.startswith("@"):
                    continue
                cols = line.rstrip("\n").split("\t")
                out[cols[0]] = [c for c in cols if c.startswith("sp:f:")][0][5:]
            else:
                if line.startswith("id,"):
                    continue
                cols = line.rstrip("\n").split(",")
                out[cols[0]] = cols[-1]
    return out


@pytest.fixture(scope="module")
def toy_gpu(toy):
    oidx = ol.OracleIndex(toy["text"])
    gidx = api.GenomeIndex.build(toy["text"], device=0)
    gs = api.make_genome_structure(toy["names"], toy["lengths"])
    yield toy, oidx, gidx, gs
    gidx.close()
    oidx.close()


SCORE_CASES = [
    dict(m=3, golden="ref_m3_csv.csv"), dict(m=3, sam=True, golden="ref_m3_sam.sam"),
    dict(m=3, max_off=2, golden="ref_m3_csv_max2.csv"), dict(m=3, sam=True, max_off=2, golden="ref_m3_sam_max2.sam"),
    dict(m=3, alt=("NAG",), golden="ref_m3_csv_nag.csv"), dict(m=3, sam=True, alt=("NAG",), golden="ref_m3_sam_nag.sam"),
    dict(m=4, golden="ref_m4_csv.csv"), dict(m=2, start=True, golden="ref_m2_csv_start.csv"),
    dict(m=0, golden="ref_m0_csv.csv"), dict(m=6), dict(m=5, sam=True, max_off=1),
]


@pytest.mark.parametrize("cfg", SCORE_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_device_scores_bit_exact(toy_gpu, cfg):
    toy, oidx, gidx, gs = toy_gpu
    m, sam, alt = cfg["m"], cfg.get("sam", False), cfg.get("alt", ())
    start, max_off = cfg.get("start", False), cfg.get("max_off", -1)
    gold = golden_specificities(toy["dir"] / cfg["golden"], sam) if "golden" in cfg else None
    L = ol.lib()
    checked = 0
    for P, group in ((3, [k for k in toy["kmers"] if k.pam]), (0, [k for k in toy["kmers"] if not k.pam])):
        if not group:
            continue
        seqs = np.array([list(k.sequence.encode()) for k in group], dtype=np.uint8)
        pams = np.array([list(k.pam.encode()) for k in group], dtype=np.uint8).reshape(len(group), P)
        offsets, hits, _ = gidx.enumerate(seqs, pams, mismatches=m, alt_pams=alt if P else (), start=start)
        cfd, spec = gidx.score(gs, seqs, P, offsets, hits, sam=sam, start=start, max_off_targets=max_off)
        opts = ol.make_opts(mismatches=m, alt_pams=alt, start=start, max_off_targets=max_off)
        for i, k in enumerate(group):
            ohits, _, raw = oidx.enumerate(k.sequence, k.pam, opts)
            assert len(ohits) == offsets[i + 1] - offsets[i]
            # every hit's CFD, bit for bit (printer.hpp:98-113 through the oracle's restatement)
            for j, h in enumerate(ohits):
                ms = h[3].translate(_COMP)
                pam = ms[20:23] if len(ms) >= 20 else ""
                exp = np.float32(L.gso_calculate_cfd(k.sequence.encode(), ms.encode(), pam.encode()))
                got = cfd[int(offsets[i]) + j]
                assert got.view(np.uint32) == exp.view(np.uint32), (k.id, j, got, exp)
            # the guide's specificity as the oracle's printers write it
            text = ol.text_lines("sam" if sam else "csv", toy["names"], toy["lengths"], k.id, k.sequence, k.pam,
                                 k.positive, opts, raw)
            L.gso_free(raw[0])
            lines = text.splitlines()
            if sam:
                osp = {[c for c in ln.split("\t") if c.startswith("sp:f:")][0][5:] for ln in lines}
            else:
                osp = {ln.split(",")[-1] for ln in lines}
            if osp and osp != {"1.0"}:   # "1.0" is the literal of the no-hit CSV row (printer.hpp:189-199)
                assert osp == {fmt_f(spec[i])}, (k.id, osp, spec[i])
                checked += 1
            if len(ohits) == 0:
                assert spec[i] == np.float32(1.0)
            # and as the reference binary wrote it
            if gold is not None and k.id in gold and gold[k.id] != "1.0":
                assert gold[k.id] == fmt_f(spec[i]), (k.id, gold[k.id], spec[i])
    assert checked >= 1


def test_scores_on_a_multi_chromosome_genome_with_boundary_hits():
    """guides sampled across chromosome joins (their on-target windows straddle a boundary and are
    dropped by resolve_absolute, structures.cxx:46-48) plus ordinary guides, m=3, all four rule sets"""
    rng = np.random.default_rng(11)
    lengths = [40_000, 3_000, 25_000, 61, 30_000]
    text, names, lengths = synth.make_genome(lengths, seed=21, n_blocks=False)
    # plant copies of one 23-mer so that several guides have many hits, some across joins
    site = np.frombuffer(b"GACGTTACCGGATTACAGCATGG", dtype=np.uint8)
    for at in (100, 39_990, 43_020, 68_050, 90_000):
        text[at:at + 23] = site
    for j in range(40):   # near copies (1-3 substitutions, any xGG PAM): a spread of CFD values
        cp = site.copy()
        for p in rng.choice(20, size=1 + j % 3, replace=False):
            cp[p] = rng.choice([c for c in b"ACGT" if c != cp[p]])
        cp[20] = rng.choice(list(b"ACGT"))
        at = 1_000 + 2_300 * j
        text[at:at + 23] = cp if j % 2 == 0 else synth.reverse_complement_bytes(cp)
    seqs, pams, _, _ = synth.sample_guides(text, 60, seed=3)
    extra = []
    cum = np.cumsum(lengths)
    for b in cum[:-1]:
        for off in (-22, -12, -3):
            w = text[b + off:b + off + 23]
            if w.shape[0] == 23 and all(c in b"ACGT" for c in w.tobytes()):
                extra.append(w[:20].copy())
    extra.append(site[:20].copy())
    seqs = np.concatenate([seqs, np.array(extra, dtype=np.uint8)])
    pams = np.tile(np.frombuffer(b"NGG", dtype=np.uint8), (seqs.shape[0], 1))
    oidx = ol.OracleIndex(text)
    gidx = api.GenomeIndex.build(text, device=0)
    gs = api.make_genome_structure(names, lengths)
    L = ol.lib()
    try:
        offsets, hits, _ = gidx.enumerate(seqs, pams, mismatches=3)
        n_sentinel = 0
        for sam, max_off in ((False, -1), (True, -1), (False, 1), (True, 2)):
            cfd, spec = gidx.score(gs, seqs, 3, offsets, hits, sam=sam, max_off_targets=max_off)
            opts = ol.make_opts(mismatches=3, max_off_targets=max_off)
            for i in range(seqs.shape[0]):
                g = seqs[i].tobytes().decode()
                ohits, _, raw = oidx.enumerate(g, "NGG", opts)
                text_out = ol.text_lines("sam" if sam else "csv", names, lengths, f"g{i}", g, "NGG", True, opts, raw)
                L.gso_free(raw[0])
                lines = text_out.splitlines()
                if sam:
                    osp = {[c for c in ln.split("\t") if c.startswith("sp:f:")][0][5:] for ln in lines}
                else:
                    osp = {ln.split(",")[-1] for ln in lines}
                    if max_off == -1:   # one CSV row per hit that resolve_absolute keeps
                        n_sentinel += len(ohits) - len([ln for ln in lines if ",NA,NA,NA," not in ln])
                if osp and osp != {"1.0"}:
                    assert osp == {fmt_f(spec[i])}, (i, sam, max_off, osp, spec[i])
        assert n_sentinel > 0, "no boundary-straddling hit in this input: the sentinel rule went untested"
    finally:
        gidx.close()
        oidx.close()


def test_score_rejects_bad_arguments(toy_gpu):
    toy, oidx, gidx, gs = toy_gpu
    seqs = np.array([list(b"ACGTACGTACGTACGTACGTACGTACGTACGTA")], dtype=np.uint8)   # L = 33
    with pytest.raises(api.GsError) as e:
        gidx.score(gs, seqs, 3, np.zeros(2, np.uint64), np.empty(0, api.HIT_DTYPE))
    assert e.value.status == 3


# ---- candidate-guide generation ---------------------------------------------------------------

@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_device_kmers_equal_the_reference_scripts_rows(case):
    """gs_kmers_generate against the rows the reference's own script gave (tests/golden/kmers): PAMs at
    the record ends, lower case, N runs, --start, NNGAAT, NNN, records shorter than a site"""
    got = kmers.find_all_kmers_device(case["record"].encode(), case["pam"], case["k"], case["start"])
    assert got == expected_rows(case)


@pytest.mark.parametrize("pam,k,start", [("NGG", 20, False), ("NGN", 19, True), ("TTTN", 23, True),
                                         ("NNGAAT", 21, False), ("NNN", 5, False), ("AGG", 20, False)])
def test_device_kmers_have_the_site_properties(pam, k, start):
    """independent brute force: every reported site is a site, every site is reported once, order =
    strand, PAM expansion, position"""
    rng = np.random.default_rng(5 + k)
    seq = "".join(rng.choice(list("ACGTNacgt"), 6000, p=[.22, .22, .22, .22, .04, .02, .02, .02, .02]))
    seq = "GG" + seq + "CC"          # PAMs at the very ends
    got = kmers.find_all_kmers_device(seq.encode(), pam, k, start)
    check_properties(seq, pam, k, start, got)
    assert len(got) >= 3


def test_device_kmers_edge_inputs():
    assert kmers.find_all_kmers_device(b"", "NGG", 20) == []
    assert kmers.find_all_kmers_device(b"GG", "NGG", 20) == []
    assert kmers.find_all_kmers_device(b"A" * 20 + b"TGG", "NGG", 20) == [("A" * 20, 1, "+")]
    assert kmers.find_all_kmers_device(b"CCA" + b"T" * 20, "NGG", 20) == [("A" * 20, 1, "-")]
    with pytest.raises(api.GsError):
        kmers.find_all_kmers_device(b"ACGT" * 10, "NGR", 20)


def test_device_kmers_feed_enumerate_without_leaving_hbm():
    """3 Mbp chromosome: the device scan equals the numpy restatement, and its HBM arrays go
    straight into gs_enumerate_device; the hit lists equal the host-pointer path's"""
    text, names, lengths = synth.make_genome([3_000_000], seed=9)
    exp = kmers.find_all_kmers(text, "NGG", 20)
    km = api.generate_kmers(text, "NGG", 20)
    try:
        seqs, pams, pos, sense = km.to_host()
        assert km.n == len(exp) > 200_000
        assert [s.tobytes().decode() for s in seqs[:2000]] == [e[0] for e in exp[:2000]]
        assert np.array_equal(pos, np.array([e[1] for e in exp], dtype=np.uint32))
        assert np.array_equal(sense, np.array([ord(e[2]) for e in exp], dtype=np.uint8))
        assert np.array_equal(seqs[-1000:], np.array([list(e[0].encode()) for e in exp[-1000:]], dtype=np.uint8))
        assert (pams == np.frombuffer(b"NGG", dtype=np.uint8)).all()
        gidx = api.GenomeIndex.build(text, device=0)
        try:
            n = 4096
            first = km.n // 2   # a slice that holds - strand records too
            d_off, d_hits, st = gidx.enumerate_device(km.seqs_ptr + first * 20, n, 20, km.pams_ptr + first * 3, 3,
                                                      mismatches=2)
            hip = C.CDLL("libamdhip64.so")
            hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            off = np.empty(n + 1, dtype=np.uint64)
            assert hip.hipMemcpy(off.ctypes.data, d_off, 8 * (n + 1), 2) == 0
            got = np.empty(int(off[-1]), dtype=api.HIT_DTYPE)
            assert hip.hipMemcpy(got.ctypes.data, d_hits, 16 * got.shape[0], 2) == 0
            off2, hits2, _ = gidx.enumerate(seqs[first:first + n], pams[first:first + n], mismatches=2)
            assert np.array_equal(off, off2) and np.array_equal(got, hits2)
            assert off[-1] >= n   # every candidate finds at least itself
        finally:
            gidx.close()
    finally:
        km.close()
