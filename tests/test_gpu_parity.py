"""GPU parity tests: the HIP path through the C-ABI vs the CPU oracle (bit-exact).

Run on the GPU box with `pytest -m gpu`.  Nothing here reads /root/reference."""
from importlib import import_module

import numpy as np
import pytest

import oracle_lib as ol

api = import_module("guidescan-cli_amd.api")
synth = import_module("guidescan-cli_amd.synth")

pytestmark = pytest.mark.gpu


def oracle_hits_as_records(oidx, seq, pam, opts, P, start=False):
    hits, ctr, raw = oidx.enumerate(seq, pam, opts)
    ol.lib().gso_free(raw[0])
    return [(h[0], h[1], h[2], h[3]) for h in hits], ctr


def gpu_hits_as_records(offsets, hits, i, guide, P, start=False):
    out = []
    for h in hits[offsets[i]:offsets[i + 1]]:
        key = int(h["key"])
        out.append((int(h["pos"]), key >> 61, (key >> 60) & 1,
                    api.decode_sequence(guide, P, key, api.GS_FLAG_PAM_AT_START if start else 0)))
    return out


@pytest.fixture(scope="module")
def toy_gpu(toy):
    oidx = ol.OracleIndex(toy["text"])
    gidx = api.GenomeIndex.build(toy["text"], device=0)
    yield toy, oidx, gidx
    gidx.close()
    oidx.close()


def test_every_row_check_proves_a_suffix_array_and_catches_a_wrong_one(toy_gpu):
    """gs_index_verify_sa with GS_VERIFY_ALL_ROWS: zero findings on the index's own arrays; an index built on a suffix array
    with two neighbouring rows swapped - still a permutation, the sampled check would have to hit that very pair - is caught"""
    toy, oidx, gidx = toy_gpu
    text = toy["text"]
    for s in (0, 1):
        rep = gidx.verify_sa(text, strand=s, samples="all")
        assert rep["sampled"] == text.shape[0]
        assert rep["not_permutation"] == rep["out_of_order"] == rep["undecided"] == rep["bwt_mismatch"] == 0, rep
    sa_f, sa_r = gidx.suffix_array(0).copy(), gidx.suffix_array(1).copy()
    r = sa_f.shape[0] // 3
    sa_f[r], sa_f[r + 1] = sa_f[r + 1], sa_f[r]
    bad = api.GenomeIndex.build(text, device=0, sa_fwd=sa_f, sa_rev=sa_r)
    try:
        rep = bad.verify_sa(text, strand=0, samples="all")
        assert rep["not_permutation"] == 0 and rep["out_of_order"] >= 1, rep
        rep = bad.verify_sa(text, strand=1, samples="all")
        assert rep["out_of_order"] == rep["bwt_mismatch"] == rep["not_permutation"] == 0, rep
    finally:
        bad.close()


def test_suffix_array_matches_oracle(toy_gpu):
    toy, oidx, gidx = toy_gpu
    for s, which in ((0, "fwd"), (1, "rev")):
        assert np.array_equal(gidx.suffix_array(s), oidx.sa(which))


def test_rank_bwt_matches_oracle_everywhere(toy_gpu):
    """Occ(c,i) for every row i in [0,n] and c in ACGT (csa_wt.hpp:270-273)."""
    toy, oidx, gidx = toy_gpu
    L = ol.lib()
    n = toy["text"].shape[0] + 1
    rows = np.arange(0, n + 1, dtype=np.uint64)
    for s, h in ((0, oidx.fwd), (1, oidx.rev)):
        got = gidx.rank_bwt4(rows, strand=s)
        step = 7
        for i in list(range(0, n + 1, step)) + [n]:
            for j, c in enumerate(b"ACGT"):
                assert got[i, j] == L.gso_rank_bwt(h, i, c), (s, i, chr(c))
        C5, size = gidx.meta(s)
        assert size == n
        for j, c in enumerate(b"ACGTN"):
            assert C5[j] == L.gso_C(h, c)


def test_resolve_matches_oracle(toy_gpu):
    toy, oidx, gidx = toy_gpu
    n = toy["text"].shape[0] + 1
    rng = np.random.default_rng(3)
    rows = np.concatenate([rng.integers(0, n, 4000), [0, n - 1]]).astype(np.uint64)
    for s, h in ((0, oidx.fwd), (1, oidx.rev)):
        got = gidx.resolve(rows, strand=s)
        exp = [ol.lib().gso_locate(h, int(r)) for r in rows]
        assert got.tolist() == exp


CASES = [dict(m=0), dict(m=1), dict(m=2), dict(m=3), dict(m=4), dict(m=5), dict(m=6),
         dict(m=3, alt=("NAG",)), dict(m=2, start=True), dict(m=3, alt=("NAG", "NGA")),
         dict(m=5, alt=("NAG",), start=True)]


@pytest.mark.parametrize("cfg", CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_toy_hit_sets_bit_exact(toy_gpu, cfg):
    """every guide of the toy kmers file: same hits, same order, same counters"""
    toy, oidx, gidx = toy_gpu
    m = cfg["m"]
    alt = cfg.get("alt", ())
    start = cfg.get("start", False)
    for P, group in ((3, [k for k in toy["kmers"] if k.pam]), (0, [k for k in toy["kmers"] if not k.pam])):
        if not group:
            continue
        seqs = np.array([list(k.sequence.encode()) for k in group], dtype=np.uint8)
        pams = np.array([list(k.pam.encode()) for k in group], dtype=np.uint8).reshape(len(group), P)
        opts = ol.make_opts(mismatches=m, alt_pams=alt, start=start)
        expected = [oracle_hits_as_records(oidx, k.sequence, k.pam, opts, P, start) for k in group]
        for faithful in (True, False):   # reference-order walk, then the prefix-table shortcut
            offsets, hits, stats = gidx.enumerate(seqs, pams, mismatches=m, alt_pams=alt if P else (),
                                                  start=start, faithful=faithful)
            for i, k in enumerate(group):
                got = gpu_hits_as_records(offsets, hits, i, k.sequence, P, start)
                assert got == expected[i][0], (k.id, cfg, faithful)
            if faithful:
                assert stats["n_ext"] == sum(e[1].n_ext for e in expected)
            else:
                assert stats["n_ext"] <= sum(e[1].n_ext for e in expected)
            assert stats["n_hits"] == offsets[-1]


def general_hits_as_records(offsets, hits, i):
    return [(int(x["pos"]), int(x["mismatches"]), int(x["index"]), api.decode_sequence_ex(x), int(x["dna_bulges"]),
             int(x["rna_bulges"])) for x in hits[offsets[i]:offsets[i + 1]]]


def oracle_general_records(oidx, seq, pam, opts):
    _, ctr, raw = oidx.enumerate(seq, pam, opts)
    out, n = raw
    exp = [(out[j].pos, out[j].mismatches, out[j].index, out[j].sequence.decode(), out[j].dna_bulges,
            out[j].rna_bulges) for j in range(n)]
    ol.lib().gso_free(out)
    return exp


ODD_GUIDES = ["ACGTNCGTACGTACGTACGT", "acgtACGTACGTACGTACGT", "ACGTACGTACGTACGTACGR", "NNACGTACGTACGTACGTAC"]


@pytest.mark.parametrize("cfg", [dict(m=2), dict(m=3, alt=("NAG", "NGA", "NGT", "NTG", "NCG")),
                                 dict(m=1, alt=("NAG", "RGG", "nGG", "NGN", "NNG", "NGC"), start=True),
                                 dict(m=2, alt=("NAG",), own="NRG")],
                         ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_symbols_outside_acgt_and_long_pam_lists(toy_gpu, cfg):
    """what the reference accepts, the product accepts (index.hpp:125-170, 218-247; process.hpp:51-56):
    a guide with N / lower case / IUPAC symbols does not abort its batch - the fast path answers the
    other guides (any number of alt PAMs, four per pass) and flags the odd ones, which the general
    path then enumerates exactly; alt PAMs with literals the genome does not hold simply never match"""
    toy, oidx, gidx = toy_gpu
    m, alt, start, own = cfg["m"], cfg.get("alt", ()), cfg.get("start", False), cfg.get("own", "NGG")
    # take guides from the toy genome itself so that the odd ones sit next to real sites
    normal = [k.sequence for k in toy["kmers"] if k.pam and len(k.sequence) == 20][:12]
    t = toy["text"]
    isn = t == ord("N")
    lone = np.nonzero(isn[1:-1] & ~isn[:-2] & ~isn[2:])[0] + 1   # an isolated literal N of the genome
    at = int(lone[0]) if lone.size else int(np.nonzero(isn)[0][0])
    odd = list(ODD_GUIDES) + [t[at - 4:at + 16].tobytes().decode()]   # a site whose 5th base is that N
    guides = normal[:6] + odd + normal[6:]
    seqs = np.array([list(g.encode()) for g in guides], dtype=np.uint8)
    pams = np.tile(np.frombuffer(own.encode(), np.uint8), (len(guides), 1))
    opts = ol.make_opts(mismatches=m, alt_pams=alt, start=start)
    offsets, hits, stats = gidx.enumerate(seqs, pams, mismatches=m, alt_pams=alt, start=start)
    plain_own = all(c in "ACGTN" for c in own)
    want_flagged = [i for i, g in enumerate(guides) if not plain_own or any(c not in "ACGT" for c in g)]
    assert stats["needs_general"] == want_flagged
    for i, g in enumerate(guides):
        if i in want_flagged:
            assert offsets[i + 1] == offsets[i]
        else:
            exp, _ = oracle_hits_as_records(oidx, g, own, opts, 3, start)
            assert gpu_hits_as_records(offsets, hits, i, g, 3, start) == exp, (i, g, cfg)
    # the general path: every guide of the batch, odd or not, equals the oracle
    goff, ghits = gidx.enumerate_general(seqs, pams, mismatches=m, alt_pams=alt, start=start)
    n_lit = 0
    for i, g in enumerate(guides):
        exp = oracle_general_records(oidx, g, own, opts)
        got = general_hits_as_records(goff, ghits, i)
        assert got == exp, (i, g, cfg)


def test_literal_symbols_match_literally():
    """a guide's N against a literal N of the genome is an exact match, against a base a mismatch
    (index.hpp:218-247); an IUPAC letter in a PAM pattern matches only the same letter in the genome
    (index.hpp:130-137): planted sites on both strands, general path vs oracle"""
    rng = np.random.default_rng(3)
    text = rng.choice(np.frombuffer(b"ACGT", np.uint8), 30_000)
    site = np.frombuffer(b"ACGTNCGTACGATTGCATGC", np.uint8)
    text[1000:1023] = np.concatenate([site, np.frombuffer(b"TGG", np.uint8)])                 # N under the guide's N
    text[5000:5023] = synth.reverse_complement_bytes(np.concatenate([site, np.frombuffer(b"AGG", np.uint8)]))
    plain = site.copy()
    plain[4] = ord("A")
    text[9000:9023] = np.concatenate([plain, np.frombuffer(b"CGG", np.uint8)])                # a base under the guide's N
    text[12000:12023] = np.concatenate([plain, np.frombuffer(b"RGG", np.uint8)])              # literal R in the genome's PAM
    text[15000:15023] = np.concatenate([plain, np.frombuffer(b"NGG", np.uint8)])              # literal N in the genome's PAM
    oidx = ol.OracleIndex(text)
    gidx = api.GenomeIndex.build(text, device=0)
    try:
        guides = [site.tobytes().decode(), plain.tobytes().decode()]
        seqs = np.array([list(g.encode()) for g in guides], dtype=np.uint8)
        for own, alt in (("NGG", ()), ("NGG", ("RGG",)), ("RGG", ("NAG",))):
            pams = np.tile(np.frombuffer(own.encode(), np.uint8), (2, 1))
            opts = ol.make_opts(mismatches=2, alt_pams=alt)
            goff, ghits = gidx.enumerate_general(seqs, pams, mismatches=2, alt_pams=alt)
            seen = set()
            for i, g in enumerate(guides):
                exp = oracle_general_records(oidx, g, own, opts)
                assert general_hits_as_records(goff, ghits, i) == exp, (i, own, alt)
                seen |= {e[3] for e in exp}
            if own == "NGG" and not alt:
                assert any(sq[4] == "N" for sq in seen) and any(sq[4] in "acgt" for sq in seen) and \
                    any(sq[20] == "N" for sq in seen)
            else:
                assert any("R" in sq[20:] for sq in seen)
            # the fast path with an alt PAM whose literal the genome holds hands the whole batch over
            offsets, hits, stats = gidx.enumerate(seqs, pams, mismatches=2, alt_pams=alt)
            assert stats["needs_general"] == ([0, 1] if (own != "NGG" or "RGG" in alt) else [0])
    finally:
        gidx.close()
        oidx.close()


def test_reference_index_files_on_the_device_and_damaged_ones(toy_gpu, tmp_path):
    """gs_index_open_sdsl builds text and suffix arrays of both strands from the reference's files on the device, without
    a suffix sort (gs_sdsl_import.hip: wavelet-tree access, LF walks from the file's own samples): the index answers as the
    one built from the text.  A damaged file - eight bytes flipped anywhere in either file - is GS_ERR_FORMAT / _IO /
    _UNSUPPORTED or, where the flip hit bytes the path does not use (the inverse samples), an index with the same answers:
    never a fault, never other hits (the walks trust the file's samples: the arrays are checked as a permutation and,
    at sampled places, as ordering the text)."""
    toy, oidx, gidx = toy_gpu
    golden = ol.ROOT / "tests" / "golden" / "toy"
    group = [k for k in toy["kmers"] if len(k.pam) == 3][:16]
    seqs = np.array([list(k.sequence.encode()) for k in group], dtype=np.uint8)
    pams = np.array([list(k.pam.encode()) for k in group], dtype=np.uint8)
    off0, hits0, _ = gidx.enumerate(seqs, pams, mismatches=3)
    imp = api.GenomeIndex.open_sdsl(golden / "toy.idx", device=0)
    try:
        off1, hits1, _ = imp.enumerate(seqs, pams, mismatches=3)
    finally:
        imp.close()
    assert off1.tobytes() == off0.tobytes() and hits1.tobytes() == hits0.tobytes()
    good = {e: (golden / f"toy.idx.{e}").read_bytes() for e in ("forward", "reverse")}
    rng = np.random.default_rng(11)
    opened = refused = 0
    for trial in range(48):
        which = ("forward", "reverse")[trial & 1]
        b = bytearray(good[which])
        at = int(rng.integers(0, len(b) - 8))
        for j in range(8):
            b[at + j] ^= 0xFF
        for e in ("forward", "reverse"):
            (tmp_path / f"d.idx.{e}").write_bytes(bytes(b) if e == which else good[e])
        try:
            ix = api.GenomeIndex.open_sdsl(tmp_path / "d.idx", device=0)
        except api.GsError as e:
            assert e.status in (4, 5, 6), (trial, which, at, e.status)
            refused += 1
            continue
        try:
            off2, hits2, _ = ix.enumerate(seqs, pams, mismatches=3)
        finally:
            ix.close()
        assert off2.tobytes() == off0.tobytes() and hits2.tobytes() == hits0.tobytes(), (trial, which, at)
        opened += 1
    assert refused >= 8, (opened, refused)


def test_a_prepared_handle_answers_as_a_fresh_one(toy_gpu):
    """gs_index_prepare runs the first batch's one-off work (seed recipes, PAM-pair / deep tables, workspace) on generated
    guides and drops the result: the handle then answers a real batch with the bytes a fresh handle gives - for the shape it
    was prepared for and for another one"""
    toy, oidx, gidx = toy_gpu
    group = [k for k in toy["kmers"] if len(k.pam) == 3]
    seqs = np.array([list(k.sequence.encode()) for k in group], dtype=np.uint8)
    pams = np.array([list(k.pam.encode()) for k in group], dtype=np.uint8)
    other = api.GenomeIndex.build(toy["text"], device=0)
    try:
        other.prepare(5000, L=seqs.shape[1], pam="NGG", alt_pams=("NAG",), mismatches=3)
        other.prepare(0, L=seqs.shape[1], pam="NGG", mismatches=2)      # nothing to do
        for kw in (dict(mismatches=3, alt_pams=("NAG",)), dict(mismatches=4), dict(mismatches=2, start=True)):
            a_off, a_hits, _ = gidx.enumerate(seqs, pams, **kw)
            b_off, b_hits, _ = other.enumerate(seqs, pams, **kw)
            assert a_off.tobytes() == b_off.tobytes() and a_hits.tobytes() == b_hits.tobytes(), kw
        with pytest.raises(api.GsError):
            other.prepare(10, L=40, pam="NGG")                           # a guide length the path does not take
    finally:
        other.close()


def test_empty_batch(toy_gpu):
    toy, oidx, gidx = toy_gpu
    offsets, hits, stats = gidx.enumerate(np.empty((0, 20), np.uint8), np.empty((0, 3), np.uint8))
    assert offsets.tolist() == [0] and hits.shape[0] == 0


def test_medium_genome_random_guides_bit_exact():
    """2 Mbp genome with N blocks, 300 guides on both strands, m=3: hits, order, counters"""
    text, names, lengths = synth.make_genome([900_000, 700_000, 400_000], seed=5)
    oidx = ol.OracleIndex(text)
    gidx = api.GenomeIndex.build(text, device=0)
    try:
        assert np.array_equal(gidx.suffix_array(0), oidx.sa("fwd"))
        assert np.array_equal(gidx.suffix_array(1), oidx.sa("rev"))
        seqs, pams, pos, strands = synth.sample_guides(text, 300, seed=9)
        opts = ol.make_opts(mismatches=3)
        expected = [oracle_hits_as_records(oidx, seqs[i].tobytes().decode(), "NGG", opts, 3)
                    for i in range(seqs.shape[0])]
        for faithful in (True, False):
            offsets, hits, stats = gidx.enumerate(seqs, pams, mismatches=3, faithful=faithful)
            for i in range(seqs.shape[0]):
                g = seqs[i].tobytes().decode()
                got = gpu_hits_as_records(offsets, hits, i, g, 3)
                assert got == expected[i][0], (i, faithful)
                assert len(got) >= 1  # the sampled site itself
            if faithful:
                assert stats["n_ext"] == sum(e[1].n_ext for e in expected)
    finally:
        gidx.close()
        oidx.close()


def test_match_slot_overflow_retry():
    """a highly repetitive genome overflows the first slot capacity and is redone exactly"""
    rng = np.random.default_rng(1)
    unit = rng.choice(np.frombuffer(b"ACGT", np.uint8), 2000)
    site = np.frombuffer(b"GATTACAGATTACAGATTAC", np.uint8)
    chunks = []
    for i in range(300):
        s = site.copy()
        for j in rng.choice(20, size=i % 4, replace=False):
            s[j] = rng.choice([c for c in b"ACGT" if c != s[j]])
        pam = np.frombuffer(rng.choice([b"AGG", b"CGG", b"GGG", b"TGG"]), np.uint8)
        chunks += [unit[: 50 + (i % 7)], s, pam]
    text = np.concatenate(chunks)
    oidx = ol.OracleIndex(text)
    gidx = api.GenomeIndex.build(text, device=0)
    try:
        # the overflowing guide sits between ordinary ones: only it is redone, and its hits
        # must land at its own CSR position
        others, _, _, _ = synth.sample_guides(text, 6, seed=4)
        seqs = np.concatenate([others[:3], np.array([list(site)], dtype=np.uint8), others[3:],
                               np.array([list(site)], dtype=np.uint8)])
        pams = np.tile(np.frombuffer(b"NGG", np.uint8), (seqs.shape[0], 1))
        for faithful in (False, True):
            offsets, hits, stats = gidx.enumerate(seqs, pams, mismatches=3, faithful=faithful)
            n_matches = 0
            for i in range(seqs.shape[0]):
                g = seqs[i].tobytes().decode()
                exp, ctr = oracle_hits_as_records(oidx, g, "NGG", ol.make_opts(3), 3)
                got = gpu_hits_as_records(offsets, hits, i, g, 3)
                assert got == exp, (i, faithful)
                n_matches += len(set((e[2], e[3]) for e in exp))
            assert stats["n_matches"] == n_matches
            assert stats["n_matches"] > 2 * 64
    finally:
        gidx.close()
        oidx.close()


@pytest.mark.parametrize("pk", ["8", "10", "12", "13"])
def test_context_verification_paths_bit_exact(toy, pk, monkeypatch):
    """force deep prefix tables so that intervals are resolved through ctx[] (the hg38-size
    code path) on small genomes: toy (literal-N PAM, boundaries, repeats) and a 2 Mbp genome"""
    monkeypatch.setenv("GS_PREFIX_K", pk)
    oidx = ol.OracleIndex(toy["text"])
    gidx = api.GenomeIndex.build(toy["text"], device=0)
    try:
        for cfg in (dict(m=3), dict(m=4), dict(m=2, alt=("NAG",)), dict(m=1, start=True), dict(m=0)):
            m, alt, start = cfg["m"], cfg.get("alt", ()), cfg.get("start", False)
            for P, group in ((3, [k for k in toy["kmers"] if k.pam]), (0, [k for k in toy["kmers"] if not k.pam])):
                seqs = np.array([list(k.sequence.encode()) for k in group], dtype=np.uint8)
                pams = np.array([list(k.pam.encode()) for k in group], dtype=np.uint8).reshape(len(group), P)
                opts = ol.make_opts(mismatches=m, alt_pams=alt, start=start)
                offsets, hits, stats = gidx.enumerate(seqs, pams, mismatches=m, alt_pams=alt if P else (),
                                                      start=start)
                for i, k in enumerate(group):
                    exp, _ = oracle_hits_as_records(oidx, k.sequence, k.pam, opts, P, start)
                    assert gpu_hits_as_records(offsets, hits, i, k.sequence, P, start) == exp, (k.id, cfg, pk)
    finally:
        gidx.close()
        oidx.close()


@pytest.mark.parametrize("pk", ["9", "11", "13"])
def test_context_verification_medium_genome(pk, monkeypatch):
    monkeypatch.setenv("GS_PREFIX_K", pk)
    text, names, lengths = synth.make_genome([900_000, 700_000, 400_000], seed=5)
    # plant a few near-copies so that verification has multi-row intervals with hits
    rng = np.random.default_rng(0)
    seqs, pams, pos, strands = synth.sample_guides(text, 200, seed=9)
    for i in range(0, 40):
        s = np.concatenate([seqs[i], np.frombuffer(b"TGG", np.uint8)]).copy()
        for j in rng.choice(20, size=i % 4, replace=False):
            s[j] = rng.choice([c for c in b"ACGT" if c != s[j]])
        at = int(rng.integers(50_000, 1_900_000))
        text[at:at + 23] = s if i % 2 else synth.reverse_complement_bytes(s)
    oidx = ol.OracleIndex(text)
    gidx = api.GenomeIndex.build(text, device=0)
    try:
        opts = ol.make_opts(mismatches=3)
        offsets, hits, stats = gidx.enumerate(seqs, pams, mismatches=3)
        total = 0
        for i in range(seqs.shape[0]):
            g = seqs[i].tobytes().decode()
            exp, _ = oracle_hits_as_records(oidx, g, "NGG", opts, 3)
            assert gpu_hits_as_records(offsets, hits, i, g, 3) == exp, (i, pk)
            total += len(exp)
        assert total > 200
    finally:
        gidx.close()
        oidx.close()


def two_sided_genome(pk, L):
    """a genome with a repeat family whose copies carry 0..4 substitutions anywhere, on both
    strands, some of them with a literal N under the PAM's N, plus N runs"""
    rng = np.random.default_rng(int(pk) * 100 + L)
    text, names, lengths = synth.make_genome([150_000, 90_000], seed=int(pk), n_blocks=False)
    text = text.copy()
    fam = text[1000:1000 + L].copy()
    text[1000 + L:1003 + L] = np.frombuffer(b"AGG", np.uint8)
    for c in range(160):
        site = fam.copy()
        for j in rng.choice(L, size=int(rng.integers(0, 5)), replace=False):
            site[j] = rng.choice([x for x in b"ACGT" if x != site[j]])
        pam = bytes(rng.choice([b"AGG", b"CGG", b"TGG", b"GGG", b"NGG", b"NGG", b"AAG", b"NAG", b"ANG"]))
        w = np.concatenate([site, np.frombuffer(pam, np.uint8)])
        if c % 2:
            w = synth.reverse_complement_bytes(w)
        at = int(rng.integers(3000, text.shape[0] - 100))
        text[at:at + L + 3] = w
    for _ in range(6):
        at = int(rng.integers(3000, text.shape[0] - 3000))
        text[at:at + int(rng.integers(1, 60))] = ord("N")
    return text, fam


def test_two_sided_pieces_exceptions_and_windows(monkeypatch, capfd):
    """the corners of two-sided seeding: intervals larger than one queued descriptor holds are
    verified piece by piece (forced by shrinking GS_VERIFY_MAX down to one row per piece), rows next
    to N runs are decided from the exception list, sites with a literal N under the PAM that belong
    to the other strand's share are reported from the window list, and a PAM pattern with more than
    two N makes the item one-sided.  Hits stay bit-exact; the library's counters (GS_DEBUG) show
    that both kinds of item occurred."""
    import re
    monkeypatch.setenv("GS_PREFIX_K", "13")
    monkeypatch.setenv("GS_DEBUG", "1")
    text, fam = two_sided_genome("13", 20)
    # a second family: 60 copies on the + strand that share positions 0..12 and differ, at most
    # once, in 13..19, each with a concrete xGG PAM: seeds with 60-row intervals
    rng = np.random.default_rng(99)
    fam2 = np.frombuffer(b"TCAGGATCGTACCTGAAGTC", np.uint8)
    for c in range(60):
        site = fam2.copy()
        if c % 2:
            j = int(rng.integers(13, 20))
            site[j] = rng.choice([x for x in b"ACGT" if x != site[j]])
        w = np.concatenate([site, np.frombuffer(bytes(rng.choice([b"AGG", b"CGG", b"TGG", b"GGG"])), np.uint8)])
        at = 5000 + 1500 * c
        text[at:at + 23] = w
    oidx = ol.OracleIndex(text)
    gidx = api.GenomeIndex.build(text, device=0)
    try:
        sampled, _, _, _ = synth.sample_guides(text, 120, seed=8)
        guides = [fam.tobytes().decode(), synth.reverse_complement_bytes(fam).tobytes().decode(),
                  fam2.tobytes().decode()]
        swap = {65: 67, 67: 71, 71: 84, 84: 65}
        for pos in ((10, 11), (11,), (10, 12), (12,), (0, 1, 2), (3, 15), (17, 18, 19), (1, 10, 16)):
            for base in (fam, fam2):
                g2 = base.copy()
                for q in pos:
                    g2[q] = swap[int(g2[q])]
                guides.append(g2.tobytes().decode())
        guides += [sampled[i].tobytes().decode() for i in range(sampled.shape[0])]
        seqs = np.array([list(g.encode()) for g in guides], dtype=np.uint8)
        pams = np.tile(np.frombuffer(b"NGG", np.uint8), (len(guides), 1))
        seen = dict(both=0, one=0)
        for m, alt in ((3, ()), (2, ()), (4, ("NAG",)), (3, ("NNN",))):
            opts = ol.make_opts(mismatches=m, alt_pams=alt)
            expected = [oracle_hits_as_records(oidx, g, "NGG", opts, 3)[0] for g in guides]
            for vmax in ("1", "3", "9", "14", "1023"):
                gidx.set_option("GS_VERIFY_MAX", vmax)
                capfd.readouterr()
                offsets, hits, stats = gidx.enumerate(seqs, pams, mismatches=m, alt_pams=alt)
                err = capfd.readouterr().err
                mt = re.search(r"both strands (\d+), one-sided \(PAM with more than two N\) (\d+)", err)
                assert mt, err
                seen["both"] += int(mt.group(1))
                seen["one"] += int(mt.group(2))
                for i, g in enumerate(guides):
                    assert gpu_hits_as_records(offsets, hits, i, g, 3) == expected[i], (i, m, alt, vmax)
        assert seen["both"] > 0 and seen["one"] > 0, seen
    finally:
        gidx.close()
        oidx.close()


@pytest.mark.parametrize("buckets", [False, True], ids=["list", "buckets"])
def test_many_literal_n_windows(monkeypatch, capfd, buckets):
    """hundreds of windows with a literal N under the PAM (a scaffold-level assembly has thousands of N
    runs): the window list is scanned in several passes of 64, by both kinds of item - through the
    PAM-pair tables every such site comes from the list, with the strand tables only the other strand's
    share does - and every site is found exactly once.  buckets: the list indexed by the four 5-symbol chunks of
    the windows' guide part (what a list beyond 256 windows gets at <= 3 mismatches), forced on this short one"""
    import re
    monkeypatch.setenv("GS_PREFIX_K", "13")
    monkeypatch.setenv("GS_DEBUG", "1")
    if buckets:
        monkeypatch.setenv("GS_CAND_BUCKETS_FROM", "0")
    else:
        monkeypatch.setenv("GS_NO_CAND_BUCKETS", "1")
    rng = np.random.default_rng(77)
    text, names, lengths = synth.make_genome([260_000, 140_000], seed=21, n_blocks=False)
    text = text.copy()
    fam = text[2000:2020].copy()
    for c in range(500):
        site = fam.copy()
        for j in rng.choice(20, size=int(rng.integers(0, 5)), replace=False):
            site[j] = rng.choice([x for x in b"ACGT" if x != site[j]])
        pam = bytes(rng.choice([b"NGG", b"NGG", b"NGG", b"NAG", b"ANG", b"GNN", b"TGG"]))
        w = np.concatenate([site, np.frombuffer(pam, np.uint8)])
        if c % 2:
            w = synth.reverse_complement_bytes(w)
        at = 3000 + 780 * c + int(rng.integers(0, 700))
        text[at:at + 23] = w
    oidx = ol.OracleIndex(text)
    gidx = api.GenomeIndex.build(text, device=0)
    try:
        sampled, _, _, _ = synth.sample_guides(text, 30, seed=9)
        guides = [fam.tobytes().decode(), synth.reverse_complement_bytes(fam).tobytes().decode()]
        swap = {65: 67, 67: 71, 71: 84, 84: 65}
        for pos in ((0,), (5, 6), (12,), (2, 15), (18, 19), (1, 9, 17)):
            g2 = fam.copy()
            for q in pos:
                g2[q] = swap[int(g2[q])]
            guides.append(g2.tobytes().decode())
        guides += [sampled[i].tobytes().decode() for i in range(sampled.shape[0])]
        seqs = np.array([list(g.encode()) for g in guides], dtype=np.uint8)
        pams = np.tile(np.frombuffer(b"NGG", np.uint8), (len(guides), 1))
        for m, alt, no_tables in ((3, (), False), (4, ("NAG",), False), (3, (), True), (2, ("NGN",), False)):
            if no_tables:
                gidx.set_option("GS_NO_PAIRTAB", "1")
            capfd.readouterr()
            offsets, hits, stats = gidx.enumerate(seqs, pams, mismatches=m, alt_pams=alt)
            gidx.set_option("GS_NO_PAIRTAB", None)
            err = capfd.readouterr().err
            mt = re.search(r"literal-N windows (\d+) \+ (\d+)", err)
            assert mt and int(mt.group(1)) > 128 and int(mt.group(2)) > 128, mt   # more than two passes of 64 per strand
            assert ("bucketed" in err) == (buckets and m <= 3), (m, err)
            opts = ol.make_opts(mismatches=m, alt_pams=alt)
            n_lit = 0
            for i, g in enumerate(guides):
                exp, _ = oracle_hits_as_records(oidx, g, "NGG", opts, 3)
                assert gpu_hits_as_records(offsets, hits, i, g, 3) == exp, (i, m, alt, no_tables)
                n_lit += sum(1 for e in exp if e[3][20:].startswith("N"))
            assert n_lit > 100, n_lit   # sites whose PAM shows the genome's literal N
    finally:
        gidx.close()
        oidx.close()


@pytest.mark.parametrize("pk,L", [("13", 20), ("11", 16), ("12", 18), ("12", 20)])
def test_two_sided_seeding_bit_exact(pk, L, monkeypatch, capfd):
    """the table depth that makes k_search seed from both strands (sites with >= 2 substitutions
    among the first consumed symbols come from the other strand's table): a genome with a repeat
    family whose copies carry 0..4 substitutions anywhere, on both strands, some of them with a
    literal N under the PAM's N (only the one-sided walk sees those), plus N runs"""
    monkeypatch.setenv("GS_PREFIX_K", pk)
    monkeypatch.setenv("GS_DEBUG", "1")
    text, fam = two_sided_genome(pk, L)
    oidx = ol.OracleIndex(text)
    gidx = api.GenomeIndex.build(text, device=0)
    try:
        sampled, _, _, _ = synth.sample_guides(text, 40, seed=3, L=L)
        guides = [fam.tobytes().decode(), synth.reverse_complement_bytes(fam).tobytes().decode()]
        guides += [sampled[i].tobytes().decode() for i in range(sampled.shape[0])]
        seqs = np.array([list(g.encode()) for g in guides], dtype=np.uint8)
        for cfg in (dict(m=3), dict(m=2), dict(m=4), dict(m=3, alt=("NAG",)), dict(m=3, start=True),
                    dict(m=5, alt=("NAG", "NGA")), dict(m=6), dict(m=1), dict(m=4, alt=("NGN",)),
                    dict(m=3, start=True, pam="TTN"), dict(m=3, start=True, pam="TTTN"), dict(m=2, pam="NNGG"),
                    dict(m=3, pam="NAG", alt=("NGG",)), dict(m=3, one_table=True, alt=("NAG",)),
                    dict(m=4, no_tables=True), dict(m=3), dict(m=3, pam="NCG", frozen=True), dict(m=2, frozen=True),
                    # six patterns = two passes of the kernel, two pairs; concrete first symbols pick one of a deep line's entries
                    dict(m=3, alt=("AGG", "CGG", "TGG", "GGG", "NAG")), dict(m=2, pam="CGG", alt=("TAG",))):
            m, alt, start, own = cfg["m"], cfg.get("alt", ()), cfg.get("start", False), cfg.get("pam", "NGG")
            pams = np.tile(np.frombuffer(own.encode(), np.uint8), (len(guides), 1))
            opts = ol.make_opts(mismatches=m, alt_pams=alt, start=start)
            if cfg.get("one_table"):
                gidx.set_option("GS_PAIRTABS", "1")
            if cfg.get("no_tables"):
                gidx.set_option("GS_NO_PAIRTAB", "1")
            capfd.readouterr()
            # frozen: GS_FLAG_NO_NEW_TABLES - a pair the handle has no table for (NCG) goes without, one it has (NGG) uses it
            offsets, hits, stats = gidx.enumerate(seqs, pams, mismatches=m, alt_pams=alt, start=start,
                                                  no_new_tables=cfg.get("frozen", False))
            gidx.set_option("GS_PAIRTABS", None)
            gidx.set_option("GS_NO_PAIRTAB", None)
            err = capfd.readouterr().err
            # (12, 20): X = L + P - k = 11 symbols reaches step k - 2 of the table's two-symbol extension (round 4); a
            # four-symbol PAM there leaves 12 > k - 1 and the batch is seeded from one side
            two_sided = L + len(own) - int(pk) + 1 <= int(pk)
            assert ("two-sided seeding" in err) == two_sided, cfg
            if not two_sided:
                for i, g in enumerate(guides):
                    exp, _ = oracle_hits_as_records(oidx, g, own, opts, len(own), start)
                    assert gpu_hits_as_records(offsets, hits, i, g, len(own), start) == exp, (i, cfg, pk)
                continue
            # PAM-pair tables serve the items whose patterns all end in (at most two) pairs of concrete bases
            cnt = gidx.last_counters()
            pairs = {p[-2:] if not start else p[:2][::-1] for p in (own,) + tuple(alt)}
            concrete = all("N" not in p for p in pairs)
            want_tables = (concrete and len(pairs) <= (1 if cfg.get("one_table") else 2) and not cfg.get("no_tables")
                           and not (cfg.get("frozen") and own != "NGG"))
            assert (cnt["items_pair_tables"] > 0) == want_tables, (cfg, cnt)
            # ... and when every pattern of the batch has one (3-symbol PAMs), the other strand's side uses the deep tables
            assert ("with deep tables" in err) == (want_tables and len(own) == 3), (cfg, err)
            total = 0
            for i, g in enumerate(guides):
                exp, _ = oracle_hits_as_records(oidx, g, own, opts, len(own), start)
                assert gpu_hits_as_records(offsets, hits, i, g, len(own), start) == exp, (i, cfg, pk)
                total += len(exp)
            assert total > 100 or start or m < 2 or own != "NGG"
    finally:
        gidx.close()
        oidx.close()


def test_config1_saccer3_sized_1k_guides_m1():
    """BASELINE config 1 as a parity case: sacCer3-sized genome (16 chromosomes, 12.07 Mbp),
    1,000 NGG guides, <= 1 mismatch: every guide's hits vs the oracle"""
    text, names, lengths = synth.make_genome(synth.SACCER3_LENGTHS, seed=1, probs=(.31, .19, .19, .31))
    gidx = api.GenomeIndex.build(text, device=0)
    # the oracle's own suffix-array builder is checked against the GPU builder in the tests
    # above; at this size it is handed the arrays to keep the test short
    oidx = ol.OracleIndex(text, sa_provider=lambda s: gidx.suffix_array(s), nthreads=8)
    try:
        seqs, pams, pos, strands = synth.sample_guides(text, 1000, seed=11, minus_fraction=0.0)
        offsets, hits, stats = gidx.enumerate(seqs, pams, mismatches=1)
        opts = ol.make_opts(mismatches=1)
        for i in range(seqs.shape[0]):
            g = seqs[i].tobytes().decode()
            exp, _ = oracle_hits_as_records(oidx, g, "NGG", opts, 3)
            assert gpu_hits_as_records(offsets, hits, i, g, 3) == exp, i
    finally:
        gidx.close()
        oidx.close()


ARENA_MODES = [None, ("GS_NO_ARENA", "1"), ("GS_ARENA_CHUNKS", "2"),
               # the tiles' waves order (word, key) records instead of packed words (what serves sequence words of 2^23 and more)
               ("GS_TILE_NO_PACK", "1"),
               # the device-wide ordering (what serves batches the per-guide tile ordering does not take):
               ("GS_NO_TILE_ORDER", "1"),
               # the one-word form as one sort + rows ordered inside the runs of equal words, whatever their length
               # (the default gives up on runs beyond 32 records and sorts by row first), and the two sorts from the start
               ("GS_NO_TILE_ORDER", "1", "GS_BIG2_SHORT", "1000000"),
               # long runs from the start: one sort of (word, row bits) - all 32 row bits on these small genomes -,
               # and the two stable sorts it replaces
               ("GS_NO_TILE_ORDER", "1", "GS_BIG2_TWO_SORTS", "1"),
               ("GS_NO_TILE_ORDER", "1", "GS_BIG2_TWO_SORTS", "1", "GS_BIG2_NO_COMPOSITE", "1"),
               # heavy items shared among waves (k_search's package queue; GS_HEAVY=1: the instantiation a handle otherwise
               # picks after a batch that showed an item of 4,096 records): never; every verification pass, in the smallest
               # packages; the same with an arena that runs out under the helpers; with a queue of four packages
               ("GS_SHARE_MIN", "0"),
               ("GS_HEAVY", "1", "GS_SHARE_MIN", "1", "GS_SHARE_MAX", "128"),
               ("GS_ARENA_CHUNKS", "2", "GS_HEAVY", "1", "GS_SHARE_MIN", "1", "GS_SHARE_MAX", "128"),
               ("GS_HEAVY", "1", "GS_SHARE_MIN", "1", "GS_SHARE_MAX", "128", "GS_SHARE_QUEUE", "4"),
               ("GS_HEAVY", "1", "GS_SHARE_MIN", "1", "GS_SHARE_MAX", "128", "GS_NO_TILE_ORDER", "1"),
               # two launches: the plain form publishes every pass and leaves, the heavy form - no items of its own - runs the
               # packages beside it on a stream of the lowest priority; the same with an arena that runs out; and with the
               # second launch BEFORE the first (its waves leave at once: everything is left for a launch behind)
               ("GS_SPLIT_SHARE", "2", "GS_SHARE_MIN", "1", "GS_SHARE_MAX", "128"),
               ("GS_ARENA_CHUNKS", "2", "GS_SPLIT_SHARE", "2", "GS_SHARE_MIN", "1", "GS_SHARE_MAX", "128"),
               ("GS_SPLIT_SHARE", "3", "GS_SHARE_MIN", "1", "GS_SHARE_MAX", "128")]
ARENA_IDS = ["arena-tiles", "second-pass", "arena-exhausted", "tiles-unpacked", "device-wide", "one-sort-and-runs", "composite-sort",
             "two-sorts", "never-shared", "every-pass-shared", "shared-arena-exhausted", "shared-queue-of-four", "shared-device-wide",
             "published-run-beside", "published-arena-exhausted", "published-run-behind"]


def set_mode(monkeypatch, arena):
    for i in range(0, len(arena or ()), 2):
        monkeypatch.setenv(arena[i], arena[i + 1])


@pytest.mark.parametrize("arena", ARENA_MODES, ids=ARENA_IDS)
def test_repeat_guide_with_thousands_of_matches(monkeypatch, arena):
    """a guide whose (guide, strand) match count exceeds the LDS sort (2048): its records beyond the slots
    come out of the overflow arena (or, arena off / too small, from the exact-size second pass), then the
    device-wide sort and the per-record locate - still bit-exact and in CSR order"""
    set_mode(monkeypatch, arena)
    rng = np.random.default_rng(7)
    site = np.frombuffer(b"GATTACAGATTACAGATTAC", np.uint8)
    chunks = []
    for i in range(7000):
        s = site.copy()
        for j in rng.choice(20, size=int(rng.integers(0, 4)), replace=False):
            s[j] = rng.choice([c for c in b"ACGT" if c != s[j]])
        pam = np.frombuffer(rng.choice([b"AGG", b"CGG", b"GGG", b"TGG"]), np.uint8)
        filler = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(30, 60)))
        chunks += [filler, s, pam]
    text = np.concatenate(chunks)
    oidx = ol.OracleIndex(text)
    gidx = api.GenomeIndex.build(text, device=0)
    try:
        others, _, _, _ = synth.sample_guides(text, 5, seed=4)
        seqs = np.concatenate([others[:2], np.array([list(site)], dtype=np.uint8), others[2:]])
        pams = np.tile(np.frombuffer(b"NGG", np.uint8), (seqs.shape[0], 1))
        # m = 6 starts at the largest slot capacity: the redo list is counted by a pass of its own
        for m, faithful in ((3, False), (3, True), (6, False)):
            offsets, hits, stats = gidx.enumerate(seqs, pams, mismatches=m, faithful=faithful)
            big = 0
            for i in range(seqs.shape[0]):
                g = seqs[i].tobytes().decode()
                exp, ctr = oracle_hits_as_records(oidx, g, "NGG", ol.make_opts(m), 3)
                got = gpu_hits_as_records(offsets, hits, i, g, 3)
                assert got == exp, (i, m, faithful)
                big = max(big, len(set((e[2], e[3]) for e in exp)))
            assert big > 2048
            ctr = gidx.last_counters()
            assert ctr["guides_redone"] >= 1 and ctr["overflow_from_arena"] == (arena is None or "ARENA" not in arena[0]), (ctr, m, faithful)
            sh = gidx.last_sharing()
            if arena and "GS_SHARE_MIN" in arena and arena[arena.index("GS_SHARE_MIN") + 1] != "0" and not faithful:
                # every item with a pass to verify was handed out; each package was run by a wave that drew its ticket
                assert sh["shared_items"] >= 2 and sh["packages"] >= 8 and sh["tickets"] >= min(sh["packages"], sh["queue_packages"]), (sh, m)
            if faithful or (arena and ("GS_NO_ARENA" in arena or ("GS_SHARE_MIN", "0") == tuple(arena[:2]))):
                assert sh["shared_items"] == 0, (sh, m, faithful)
            # the default: ordered per guide in LDS tiles (the walk's interval records are not its business)
            if arena is None:
                assert ctr["ordered_in_tiles"] == (not faithful), (ctr, m, faithful)
            elif "GS_NO_TILE_ORDER" in arena or "GS_NO_ARENA" in arena:
                assert not ctr["ordered_in_tiles"], (ctr, m, faithful)
            assert not ctr["tile_ordering_gave_up"], (ctr, m, faithful)
    finally:
        gidx.close()
        oidx.close()


@pytest.mark.parametrize("unpacked", [False, True], ids=["packed-words", "word-key-records"])
def test_tile_sizes_either_side_of_every_kernel_boundary(monkeypatch, unpacked):
    """One batch whose guides have exactly 300, 511, 512, 513, 1,023, 1,024, 1,025, 4,095, 4,096, 4,097 and 9,000 copies of
    their site (a third of them on the - strand): the items on the overflow list are tiles of one wave (<= 512 records),
    of the 128- and 512-thread workgroup kernels (<= 1,024, <= 4,096) and, beyond, dealt into buckets - every boundary
    from both sides, with equal sequences throughout (the row alone orders a family).  Hit lists equal the oracle's."""
    if unpacked:
        monkeypatch.setenv("GS_TILE_NO_PACK", "1")
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    counts = [300, 511, 512, 513, 1023, 1024, 1025, 4095, 4096, 4097, 9000]
    sites = [rng.choice(acgt, 20) for _ in counts]
    chunks = []
    for site, n in zip(sites, counts):
        for i in range(n):
            unit = np.concatenate([site, np.frombuffer(rng.choice([b"AGG", b"CGG", b"GGG", b"TGG"]), np.uint8)])
            if i % 3 == 2:
                unit = synth.reverse_complement_bytes(unit)
            chunks += [rng.choice(acgt, int(rng.integers(9, 19))), unit]
    order = rng.permutation(len(chunks) // 2)
    text = np.concatenate([np.concatenate(chunks[2 * i:2 * i + 2]) for i in order] + [rng.choice(acgt, 64)])
    oidx = ol.OracleIndex(text)
    gidx = api.GenomeIndex.build(text, device=0)
    try:
        seqs = np.array([list(s) for s in sites], dtype=np.uint8)
        pams = np.tile(np.frombuffer(b"NGG", np.uint8), (len(sites), 1))
        for m in (0, 2):
            offsets, hits, _ = gidx.enumerate(seqs, pams, mismatches=m)
            ctr = gidx.last_counters()
            assert ctr["ordered_in_tiles"] and not ctr["tile_ordering_gave_up"] and ctr["guides_redone"] >= len(counts) - 1, ctr
            opts = ol.make_opts(m)
            for i, n in enumerate(counts):
                g = seqs[i].tobytes().decode()
                exp, _ = oracle_hits_as_records(oidx, g, "NGG", opts, 3)
                assert len(exp) >= n
                assert gpu_hits_as_records(offsets, hits, i, g, 3) == exp, (n, m)
    finally:
        gidx.close()
        oidx.close()


def test_an_item_of_more_than_half_a_million_records(monkeypatch, capfd):
    """700,000 exact copies of one site on the + strand (and 1,000 on the -): the item is dealt into 684 buckets that
    aim at 1,024 records, ordered by the workgroup kernels (128, 512 and 1,024 threads) - the plan's last size class,
    beyond which (10^6 records) a batch is ordered device-wide.  Every hit in row order, as the oracle has them; a
    second time with every bucket beyond 1,024 records handed to the 1,024-thread kernel (GS_TILE_BIG_FROM: its tiles
    proper, beyond 4,096 records, take a splitter's bad luck)."""
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    site = rng.choice(acgt, 20)
    n_copies = 700_000
    unit = 32
    text = rng.choice(acgt, n_copies * unit + 40_000).astype(np.uint8)
    body = text[:n_copies * unit].reshape(n_copies, unit)
    body[:, 4:24] = site
    body[:, 24] = rng.choice(acgt, n_copies)
    body[:, 25:27] = ord("G")
    tail = text[n_copies * unit:n_copies * unit + 1000 * unit].reshape(1000, unit)
    tail[:, 2:25] = synth.reverse_complement_bytes(np.concatenate([site, np.frombuffer(b"AGG", np.uint8)]))
    oidx = ol.OracleIndex(text)
    gidx = api.GenomeIndex.build(text, device=0)
    try:
        other, _, _, _ = synth.sample_guides(text[n_copies * unit + 1000 * unit:], 2, seed=9)
        seqs = np.concatenate([other[:1], np.array([list(site)], dtype=np.uint8), other[1:]])
        pams = np.tile(np.frombuffer(b"NGG", np.uint8), (3, 1))
        import re
        gidx.set_option("GS_DEBUG", "1")
        opts = ol.make_opts(1)
        want = [oracle_hits_as_records(oidx, seqs[i].tobytes().decode(), "NGG", opts, 3)[0] for i in range(3)]
        assert len(want[1]) >= n_copies + 1000
        for per in (None, "1024"):
            if per:
                gidx.set_option("GS_TILE_BIG_FROM", per)
            capfd.readouterr()
            offsets, hits, _ = gidx.enumerate(seqs, pams, mismatches=1)
            err = capfd.readouterr().err
            ctr = gidx.last_counters()
            assert ctr["ordered_in_tiles"] and not ctr["tile_ordering_gave_up"] and ctr["matches_max_per_item"] >= n_copies, ctr
            beyond = [int(x) for x in re.findall(r"tiles beyond one wave: (\d+) of up to 1024 records, (\d+) of up to 4096, (\d+) beyond", err)[-1]]
            assert beyond[0] + beyond[1] > 300 and (per is None or beyond[2] > 0), (per, beyond)
            for i in range(3):
                got = gpu_hits_as_records(offsets, hits, i, seqs[i].tobytes().decode(), 3)
                assert len(got) == len(want[i]) and got == want[i], (i, per)
    finally:
        gidx.close()
        oidx.close()


def test_a_guide_beyond_the_tiles_reach_goes_alone_to_the_device_wide_ordering():
    """1,600,000 copies of one site on the + strand (a guide inside the largest repeat family of a real genome): its
    item holds more than 2^20 match records, beyond what the per-guide tile ordering deals into buckets.  That GUIDE
    alone is ordered device-wide (gs_enumerate.hip, big_order on the list gs_tileorder.hip's k_to_fill leaves); the rest
    of the batch - a guide of 300,000 copies that is dealt into buckets, guides of a few hits - stays in tiles.  Until
    round 5 one such item sent the whole batch through the device-wide ordering.  Every guide's hit list equals the
    oracle's (process.hpp:100-115 order; the std::set's dedupe, structures.hpp:33-43, has nothing to drop here)."""
    rng = np.random.default_rng(31)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    site, site2 = rng.choice(acgt, 20), rng.choice(acgt, 20)
    n1, n2, unit = 1_600_000, 300_000, 32
    text = rng.choice(acgt, (n1 + n2) * unit + 40_000).astype(np.uint8)
    body = text[:(n1 + n2) * unit].reshape(n1 + n2, unit)
    which = rng.permutation(n1 + n2) < n1
    body[which, 4:24] = site
    body[~which, 4:24] = site2
    body[:, 24] = rng.choice(acgt, n1 + n2)
    body[:, 25:27] = ord("G")
    gidx = api.GenomeIndex.build(text, device=0)
    oidx = ol.OracleIndex(text, sa_provider=lambda s: gidx.suffix_array(s), nthreads=8)
    try:
        other, _, _, _ = synth.sample_guides(text[(n1 + n2) * unit:], 3, seed=9)
        seqs = np.concatenate([other[:1], np.array([list(site)], dtype=np.uint8), other[1:2], np.array([list(site2)], dtype=np.uint8),
                               other[2:]])
        pams = np.tile(np.frombuffer(b"NGG", np.uint8), (seqs.shape[0], 1))
        offsets, hits, _ = gidx.enumerate(seqs, pams, mismatches=1)
        ctr, sh = gidx.last_counters(), gidx.last_sharing()
        assert ctr["ordered_in_tiles"] and not ctr["tile_ordering_gave_up"] and ctr["matches_max_per_item"] >= n1, ctr
        assert sh["guides_ordered_device_wide_alone"] == 1, sh
        opts = ol.make_opts(1)
        for i in range(seqs.shape[0]):
            g = seqs[i].tobytes().decode()
            exp, _ = oracle_hits_as_records(oidx, g, "NGG", opts, 3)
            got = gpu_hits_as_records(offsets, hits, i, g, 3)
            assert len(got) == len(exp) and got == exp, i
            assert (i != 1 or len(exp) >= n1) and (i != 3 or len(exp) >= n2)
        # a second batch on the handle (the heavy instantiation of the search by now): the same lists
        off2, hits2, _ = gidx.enumerate(seqs, pams, mismatches=1)
        assert np.array_equal(offsets, off2) and hits.tobytes() == hits2.tobytes()
        assert gidx.last_sharing()["guides_ordered_device_wide_alone"] == 1
    finally:
        gidx.close()
        oidx.close()


def test_a_heavy_item_is_run_by_many_waves():
    """One (guide, strand) item of 2 x 10^5 records - 190,000 near-copies of one site with 0..3 substitutions on the +
    strand, 10,000 on the - strand - next to five ordinary guides: k_search hands its verification passes to the waves
    that have run out of items (packages of at most 2,048 row groups in a queue in memory; gs_search.hip, `shq`), their
    records land in arena chunks of their own and k_share_fix closes the gaps.  The hit bytes equal those of the run
    in which the item stays with its wave (the plain instantiation), and the oracle's lists (process.hpp:100-115 order)."""
    rng = np.random.default_rng(23)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    site = rng.choice(acgt, 20)
    n_plus, n_minus, unit = 190_000, 10_000, 40
    n_copies = n_plus + n_minus
    text = rng.choice(acgt, n_copies * unit + 60_000).astype(np.uint8)
    body = text[:n_copies * unit].reshape(n_copies, unit)
    copies = np.tile(site, (n_copies, 1))
    nsub = rng.integers(0, 4, n_copies)
    for c in np.nonzero(nsub)[0]:
        for q in rng.choice(20, size=int(nsub[c]), replace=False):
            copies[c, q] = rng.choice([x for x in b"ACGT" if x != copies[c, q]])
    body[:n_plus, 6:26] = copies[:n_plus]
    body[:n_plus, 27:29] = ord("G")
    rc = synth.reverse_complement_bytes(np.concatenate([copies[n_plus:], rng.choice(acgt, (n_minus, 1)),
                                                        np.full((n_minus, 2), ord("G"), np.uint8)], axis=1).reshape(-1))
    body[n_plus:, 6:29] = rc.reshape(n_minus, 23)[::-1]
    oidx = ol.OracleIndex(text)
    gidx = api.GenomeIndex.build(text, device=0)
    try:
        other, _, _, _ = synth.sample_guides(text[n_copies * unit:], 5, seed=9)
        seqs = np.concatenate([other[:2], np.array([list(site)], dtype=np.uint8), other[2:]])
        pams = np.tile(np.frombuffer(b"NGG", np.uint8), (seqs.shape[0], 1))
        # the form is chosen from the batch at hand: the repeat-derived guide's own k-mer heads an interval of thousands of rows
        # (k_estimate_heavy), so already the handle's FIRST batch shares its heavy passes; without the estimate a first batch
        # runs every item on its own wave and the second one learns from its count of heavy passes
        gidx.set_option("GS_NO_FORM_ESTIMATE", "1")
        off0, hits0, _ = gidx.enumerate(seqs, pams, mismatches=3)
        assert gidx.last_sharing()["shared_items"] == 0
        gidx.set_option("GS_NO_FORM_ESTIMATE", None)
        fresh = api.GenomeIndex.build(text, device=0)
        try:
            off_f, hits_f, _ = fresh.enumerate(seqs, pams, mismatches=3)
            shf = fresh.last_sharing()
            assert shf["guides_with_heavy_kmer"] >= 1 and shf["shared_items"] >= 1 and shf["form"] == 1, shf
            assert np.array_equal(off0, off_f) and hits0.tobytes() == hits_f.tobytes()
        finally:
            fresh.close()
        off1, hits1, _ = gidx.enumerate(seqs, pams, mismatches=3)
        sh, ctr = gidx.last_sharing(), gidx.last_counters()
        assert sh["shared_items"] >= 1 and sh["packages"] >= 8 and sh["tickets"] >= sh["packages"], sh
        assert sh["form"] == 1, sh     # six guides: a batch this small publishes AND helps in one launch
        assert ctr["matches_max_per_item"] >= n_plus and ctr["ordered_in_tiles"] and not ctr["tile_ordering_gave_up"], ctr
        assert np.array_equal(off0, off1) and hits0.tobytes() == hits1.tobytes()
        # smaller packages, more of them: the same bytes again; and with the plain instantiation asked for
        gidx.set_options(GS_SHARE_MIN="64", GS_SHARE_MAX="256")
        off2, hits2, _ = gidx.enumerate(seqs, pams, mismatches=3)
        assert gidx.last_sharing()["packages"] > sh["packages"]
        assert np.array_equal(off0, off2) and hits0.tobytes() == hits2.tobytes()
        gidx.set_option("GS_HEAVY", "0")
        off3, hits3, _ = gidx.enumerate(seqs, pams, mismatches=3)
        assert gidx.last_sharing()["shared_items"] == 0
        assert np.array_equal(off0, off3) and hits0.tobytes() == hits3.tobytes()
        # a helping wave that gives up waiting (the launch was not on the chip as a whole) fails the launch, not the call: the
        # batch is redone with every item on its own wave, and the handle shares no more
        gidx.set_options(GS_SHARE_MIN=None, GS_SHARE_MAX=None, GS_HEAVY="1", GS_DBG_SHARE_TIMEOUT="1")
        off5, hits5, _ = gidx.enumerate(seqs, pams, mismatches=3)
        assert gidx.last_sharing()["shared_items"] == 0 and gidx.last_sharing()["form"] in (0, 3)
        assert np.array_equal(off0, off5) and hits0.tobytes() == hits5.tobytes()
        off6, hits6, _ = gidx.enumerate(seqs, pams, mismatches=3)   # ... and keeps off it for a while (a back-off, not for good)
        assert gidx.last_sharing()["shared_items"] == 0
        assert np.array_equal(off0, off6) and hits0.tobytes() == hits6.tobytes()
        gidx.set_options(GS_HEAVY=None, GS_DBG_SHARE_TIMEOUT=None)
        gidx.close()
        gidx = api.GenomeIndex.build(text, device=0)      # (a fresh handle: this one has given sharing up)
        gidx.set_options(GS_SHARE_MIN="64", GS_SHARE_MAX="256")
        # two launches (what a handle picks for a large batch with FEW heavy items): the plain form publishes and leaves,
        # the heavy form runs the packages beside it (2), behind it (1), or - launched first, its waves leave at once -
        # in a launch the host adds behind both (3)
        gidx.set_option("GS_HEAVY", None)
        for mode, behind in (("2", None), ("1", 0), ("3", 1)):
            gidx.set_option("GS_SPLIT_SHARE", mode)
            off4, hits4, _ = gidx.enumerate(seqs, pams, mismatches=3)
            sh4 = gidx.last_sharing()
            assert sh4["form"] == 2 and sh4["shared_items"] >= 1 and sh4["tickets"] >= sh4["packages"] > sh["packages"], (mode, sh4)
            assert behind is None or sh4["launches_behind"] == behind, (mode, sh4)
            assert np.array_equal(off0, off4) and hits0.tobytes() == hits4.tobytes(), mode
        opts = ol.make_opts(3)
        for i in range(seqs.shape[0]):
            g = seqs[i].tobytes().decode()
            exp, _ = oracle_hits_as_records(oidx, g, "NGG", opts, 3)
            got = gpu_hits_as_records(off1, hits1, i, g, 3)
            assert len(got) == len(exp) and got == exp, i
            assert i != 2 or len(exp) >= n_copies
        # the same item in a LARGE batch of light guides (more than 64 items per wave slot of the chip, heavy passes in far
        # fewer than one item of sixteen): the first batch runs plain and counts the passes, the second takes the two
        # launches by itself - the same bytes, and the heavy guide's list is the one checked against the oracle above
        gidx.set_option("GS_SPLIT_SHARE", None)
        gidx.set_options(GS_SHARE_MIN=None, GS_SHARE_MAX=None)
        big_s = np.concatenate([np.tile(other, (60_000, 1)), np.array([list(site)], dtype=np.uint8)])
        big_p = np.tile(np.frombuffer(b"NGG", np.uint8), (big_s.shape[0], 1))
        # ... when the heavy item is too large to hide behind the batch on one wave (from GS_SPLIT_FROM rows under its k-mer on:
        # 2^19 by default; the planted family has thousands): below that every item stays with its wave in the seeding launches
        boff9, bhits9, _ = gidx.enumerate(big_s, big_p, mismatches=2)
        shb9 = gidx.last_sharing()
        # (form 3 - the two seeding launches - where the batch's patterns have their deep tables; this small genome's table is too
        # shallow for them: the one launch, form 0)
        assert shb9["guides_with_heavy_kmer"] >= 1 and shb9["form"] in (0, 3) and shb9["shared_items"] == 0, shb9
        gidx.set_option("GS_SPLIT_FROM", "1000")
        boff0, bhits0, _ = gidx.enumerate(big_s, big_p, mismatches=2)     # (another budget: a shape this handle has not seen)
        shb0 = gidx.last_sharing()   # the FIRST batch of the shape: one guide of 300,001 has a heavy k-mer - two launches at once
        assert shb0["guides_with_heavy_kmer"] >= 1 and shb0["form"] == 2 and shb0["shared_items"] >= 1, shb0
        assert np.array_equal(boff0, boff9) and bhits0.tobytes() == bhits9.tobytes()
        boff1, bhits1, _ = gidx.enumerate(big_s, big_p, mismatches=2)
        shb = gidx.last_sharing()
        assert shb["form"] == 2 and shb["shared_items"] >= 1, shb
        assert np.array_equal(boff0, boff1) and bhits0.tobytes() == bhits1.tobytes()
        g = bytes(site).decode()
        exp, _ = oracle_hits_as_records(oidx, g, "NGG", ol.make_opts(2), 3)
        assert gpu_hits_as_records(boff1, bhits1, big_s.shape[0] - 1, g, 3) == exp
    finally:
        gidx.close()
        oidx.close()


def test_tile_ordering_gives_up_and_the_device_wide_form_takes_over(monkeypatch):
    """The per-guide tile ordering writes final hits on two assumptions and checks them tile by tile: no (sequence, row)
    twice in an item - overlapping PAM patterns break it: NGG listed again as an alt PAM, every site found twice, the
    per-distance std::set keeps one (process.hpp:21-23) - and no bucket beyond its slots (forced here by a sample of
    one word per splitter).  A tile that sees either raises a flag and the call orders the batch with the device-wide
    form: the counters say so and the hit lists still equal the oracle's.  A handle that met duplicates does not try
    the tiles again for that batch shape; another shape it does."""
    rng = np.random.default_rng(7)
    site = np.frombuffer(b"GATTACAGATTACAGATTAC", np.uint8)
    chunks = []
    for i in range(16000):
        s = site.copy()
        for j in rng.choice(20, size=int(rng.integers(0, 4)), replace=False):
            s[j] = rng.choice([c for c in b"ACGT" if c != s[j]])
        pam = np.frombuffer(rng.choice([b"AGG", b"CGG", b"GGG", b"TGG"]), np.uint8)
        filler = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(30, 60)))
        chunks += [filler, s, pam]
    text = np.concatenate(chunks)
    oidx = ol.OracleIndex(text)
    gidx = api.GenomeIndex.build(text, device=0)
    try:
        others, _, _, _ = synth.sample_guides(text, 3, seed=4)
        seqs = np.concatenate([others[:1], np.array([list(site)], dtype=np.uint8), others[1:]])
        pams = np.tile(np.frombuffer(b"NGG", np.uint8), (seqs.shape[0], 1))

        def check(alt, want_tiles, want_gave_up):
            offsets, hits, _ = gidx.enumerate(seqs, pams, mismatches=3, alt_pams=alt)
            ctr = gidx.last_counters()
            pick = {k: ctr[k] for k in ("ordered_in_tiles", "tile_ordering_gave_up", "matches_max_per_item", "guides_redone")}
            assert ctr["matches_max_per_item"] > 4096 and ctr["ordered_in_tiles"] == want_tiles and \
                ctr["tile_ordering_gave_up"] == want_gave_up, (alt, pick)
            opts = ol.make_opts(3, alt_pams=alt)
            for i in range(seqs.shape[0]):
                g = seqs[i].tobytes().decode()
                exp, _ = oracle_hits_as_records(oidx, g, "NGG", opts, 3)
                assert gpu_hits_as_records(offsets, hits, i, g, 3) == exp, (i, alt)

        check((), True, False)
        # a sample of one word per splitter and a quarter of the slots: most buckets outgrow theirs - what they receive
        # beyond goes to the spill list and the buckets move to slots of their real size (k_to_respill): same hits, in tiles
        gidx.set_option("GS_TILE_SAMPLE_PER", "1")
        check((), True, False)
        gidx.set_option("GS_TILE_NO_SPILL", "1")       # ... without the spill list: gives up, same hits
        check((), False, True)
        gidx.set_option("GS_TILE_NO_SPILL", None)
        gidx.set_option("GS_TILE_SAMPLE_PER", None)
        check(("NGG",), False, True)                       # every site twice: gives up, the device-wide form drops the copies
        check(("NGG",), False, False)                      # ... and this shape is not tried again on this handle
        check(("NAG",), True, False)                       # another shape is
    finally:
        gidx.close()
        oidx.close()


def test_composite_ordering_puts_runs_right(monkeypatch):
    """the device-wide ordering as ONE sort of (sort word, low bits of the first row): the rows of a run of equal
    words are scattered over their k-mer's suffix array interval, and where that reaches across a multiple of
    2^bits the run comes out of the sort out of order; k_big2_wraps finds such runs, k_big2_fixruns orders them by
    the rows' high part.  Few row bits (13, 9, 5) on the genome of 7,000 near-copies, the multiples moved through
    the runs by an offset: every result equals the oracle-checked one of the two stable sorts, and runs do get
    put right."""
    rng = np.random.default_rng(7)
    site = np.frombuffer(b"GATTACAGATTACAGATTAC", np.uint8)
    chunks = []
    for i in range(7000):
        s = site.copy()
        for j in rng.choice(20, size=int(rng.integers(0, 4)), replace=False):
            s[j] = rng.choice([c for c in b"ACGT" if c != s[j]])
        pam = np.frombuffer(rng.choice([b"AGG", b"CGG", b"GGG", b"TGG"]), np.uint8)
        filler = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(30, 60)))
        chunks += [filler, s, pam]
    text = np.concatenate(chunks)
    oidx = ol.OracleIndex(text)
    gidx = api.GenomeIndex.build(text, device=0)
    try:
        seqs = np.array([list(site)], dtype=np.uint8)
        pams = np.frombuffer(b"NGG", np.uint8).reshape(1, 3)
        gidx.set_option("GS_NO_TILE_ORDER", "1")
        gidx.set_option("GS_BIG2_TWO_SORTS", "1")
        gidx.set_option("GS_BIG2_NO_COMPOSITE", "1")
        ref_off, ref_hits, _ = gidx.enumerate(seqs, pams, mismatches=3)
        exp, _ = oracle_hits_as_records(oidx, site.tobytes().decode(), "NGG", ol.make_opts(3), 3)
        assert gpu_hits_as_records(ref_off, ref_hits, 0, site.tobytes().decode(), 3) == exp
        assert not gidx.last_counters()["ordered_by_one_composite_sort"]
        gidx.set_option("GS_BIG2_NO_COMPOSITE", None)
        for bits, step in ((13, 512), (9, 64), (5, 8)):
            gidx.set_option("GS_BIG2_ROWBITS", str(bits))
            fixed = 0
            for off in range(0, 1 << bits, step):
                gidx.set_option("GS_BIG2_ROWOFF", str(off))
                offsets, hits, _ = gidx.enumerate(seqs, pams, mismatches=3)
                ctr = gidx.last_counters()
                assert ctr["ordered_by_one_composite_sort"], (bits, off, ctr)
                assert np.array_equal(offsets, ref_off) and hits.tobytes() == ref_hits.tobytes(), (bits, off)
                fixed += bool(ctr["runs_turned_round"])
            assert fixed >= 2, (bits, fixed)
    finally:
        gidx.close()
        oidx.close()


BULGE_CASES = [dict(m=1, rna=1, dna=0), dict(m=1, rna=0, dna=1), dict(m=2, rna=1, dna=1),
               dict(m=0, rna=2, dna=2), dict(m=1, rna=0, dna=1, alt=("NAG",), start=True)]


@pytest.mark.parametrize("cfg", BULGE_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_bulge_aware_search_bit_exact(toy_gpu, cfg):
    """index.hpp:250-375 on the GPU: per guide the ordered list of
    (pos, mismatches, index, match.sequence, dna_bulges, rna_bulges) equals the oracle's"""
    toy, oidx, gidx = toy_gpu
    m, rna, dna = cfg["m"], cfg["rna"], cfg["dna"]
    alt, start = cfg.get("alt", ()), cfg.get("start", False)
    for P, group in ((3, [k for k in toy["kmers"] if k.pam]), (0, [k for k in toy["kmers"] if not k.pam])):
        seqs = np.array([list(k.sequence.encode()) for k in group], dtype=np.uint8)
        pams = np.array([list(k.pam.encode()) for k in group], dtype=np.uint8).reshape(len(group), P)
        offsets, hits = gidx.enumerate_bulges(seqs, pams, mismatches=m, rna_bulges=rna, dna_bulges=dna,
                                              alt_pams=alt if P else (), start=start)
        opts = ol.make_opts(mismatches=m, alt_pams=alt, start=start, rna_bulges=rna, dna_bulges=dna)
        for i, k in enumerate(group):
            _, ctr, raw = oidx.enumerate(k.sequence, k.pam, opts)
            out, n = raw
            exp = [(out[j].pos, out[j].mismatches, out[j].index, out[j].sequence.decode(),
                    out[j].dna_bulges, out[j].rna_bulges) for j in range(n)]
            ol.lib().gso_free(out)
            assert general_hits_as_records(offsets, hits, i) == exp, (k.id, cfg)


@pytest.mark.parametrize("arena", ARENA_MODES, ids=ARENA_IDS)
def test_repeat_family_genome_bit_exact(monkeypatch, arena):
    """a genome with 45 % of its bases in repeat families (synth.plant_repeats: SINE-like, LINE-like,
    tandem arrays, segmental duplications, both strands), table depth forced to the hg38 code path:
    guides drawn from the families have hundreds to thousands of near-copies - large intervals
    verified in pieces, slot overflow, LDS and device-wide ordering - and stay bit-exact"""
    monkeypatch.setenv("GS_PREFIX_K", "13")
    if arena:
        set_mode(monkeypatch, (arena[0], "1") if "ARENA" in arena[0] else arena)   # one chunk: the first family guide exhausts it
    text, names, lengths = synth.make_repeat_genome([2_000_000, 1_000_000], seed=4)
    oidx = ol.OracleIndex(text)
    gidx = api.GenomeIndex.build(text, device=0)
    try:
        seqs, pams, pos, strands = synth.sample_guides(text, 160, seed=12)
        for m, n in ((3, 160), (4, 48), (2, 160)):
            opts = ol.make_opts(mismatches=m)
            offsets, hits, stats = gidx.enumerate(seqs[:n], pams[:n], mismatches=m)
            big = 0
            for i in range(n):
                g = seqs[i].tobytes().decode()
                exp, _ = oracle_hits_as_records(oidx, g, "NGG", opts, 3)
                assert gpu_hits_as_records(offsets, hits, i, g, 3) == exp, (i, m)
                big = max(big, len(exp))
            assert m < 3 or big > 100, big   # some guide really sits in a family
    finally:
        gidx.close()
        oidx.close()


def test_raw_hit_counts_before_dedupe(toy_gpu):
    """GS_FLAG_RAW_COUNTS: hits per guide before duplicate sequences collapse - with the guides' own PAM
    pattern listed again as an alt PAM every hit counts twice (off_target_counter, process.hpp:25-27)"""
    toy, oidx, gidx = toy_gpu
    group = [k for k in toy["kmers"] if k.pam][:24]
    seqs = np.array([list(k.sequence.encode()) for k in group], dtype=np.uint8)
    pams = np.array([list(k.pam.encode()) for k in group], dtype=np.uint8)
    off1, hits1, st1 = gidx.enumerate(seqs, pams, mismatches=2, raw_counts=True)
    assert np.array_equal(st1["raw_hits"], np.diff(off1).astype(np.uint32))
    off2, hits2, st2 = gidx.enumerate(seqs, pams, mismatches=2, alt_pams=("NGG",), raw_counts=True)
    assert np.array_equal(off2, off1) and hits2.tobytes() == hits1.tobytes()      # the sets drop the duplicates
    assert np.array_equal(st2["raw_hits"], 2 * st1["raw_hits"])                   # the counter does not
    assert st1["raw_hits"].sum() > 0


def test_search_iteration_bound_fails_cleanly(toy_gpu, monkeypatch):
    """every loop of a search item counts against an iteration bound: with a tiny bound the call returns
    GS_ERR_DEVICE (no hang, the grid drains), and the handle serves the next call as before"""
    toy, oidx, gidx = toy_gpu
    group = [k for k in toy["kmers"] if k.pam][:16]
    seqs = np.array([list(k.sequence.encode()) for k in group], dtype=np.uint8)
    pams = np.array([list(k.pam.encode()) for k in group], dtype=np.uint8)
    want = gidx.enumerate(seqs, pams, mismatches=3)
    for faithful in (False, True):
        gidx.set_option("GS_SEARCH_MAX_ITER", "2")
        with pytest.raises(api.GsError) as e:
            gidx.enumerate(seqs, pams, mismatches=3, faithful=faithful)
        assert e.value.status == 2
        gidx.set_option("GS_SEARCH_MAX_ITER", None)
        got = gidx.enumerate(seqs, pams, mismatches=3, faithful=faithful)
        assert np.array_equal(got[0], want[0]) and got[1].tobytes() == want[1].tobytes()


def cas12a_genome(seed, n_copies, size):
    """PAM at the 5' end: sites are TTTN + a 23-nt protospacer.  A family of near-copies of one site (0..4 substitutions
    in the protospacer, the PAM's N any base, a few with another PAM), on both strands"""
    rng = np.random.default_rng(seed)
    text, names, lengths = synth.make_genome(size, seed=seed, n_blocks=False)
    text = text.copy()
    fam = text[2000:2023].copy()
    text[1996:2000] = np.frombuffer(b"TTTA", np.uint8)
    for c in range(n_copies):
        site = fam.copy()
        for j in rng.choice(23, size=int(rng.integers(0, 5)), replace=False):
            site[j] = rng.choice([x for x in b"ACGT" if x != site[j]])
        pam = bytes(rng.choice([b"TTTA", b"TTTC", b"TTTG", b"TTTT", b"TTTA", b"TCTA", b"CTTA", b"TTTN"]))
        w = np.concatenate([np.frombuffer(pam, np.uint8), site])
        if c % 2:
            w = synth.reverse_complement_bytes(w)
        at = int(rng.integers(3000, text.shape[0] - 100))
        text[at:at + 27] = w
    for _ in range(4):
        at = int(rng.integers(3000, text.shape[0] - 3000))
        text[at:at + int(rng.integers(1, 40))] = ord("N")
    return text, fam


@pytest.mark.parametrize("table_k", [13, 14])
def test_wide_keys_23mers_with_a_four_symbol_pam(monkeypatch, table_k):
    """Cas12a: 23-mers behind TTTN (--start).  2L + 3P = 58 bits of match sequence: beyond the 52 bits the hit key
    carried until round 4, inside the 59 it carries now (key bits 59:1) - the table-seeded kernels take the batch
    (one-sided seeding with 13-symbol tables; two-sided with 14: X, 13 symbols, then reaches step k-2 of the table's
    two-symbol extension), LDS orders it, and on the second genome - 30,000 near-copies of one site - the
    guide's thousands of records overflow into the arena and are ordered per guide in LDS tiles.  Every hit list
    equals the oracle's: positions, distances, index, match.sequence as gs_decode_sequence rebuilds it from the key."""
    monkeypatch.setenv("GS_PREFIX_K", str(table_k))
    for seed, n_copies, size, ms in ((21, 150, [150_000, 90_000], (1, 2, 3, 4)), (22, 30000, [3_000_000], (3,))):
        text, fam = cas12a_genome(seed, n_copies, size)
        oidx = ol.OracleIndex(text)
        gidx = api.GenomeIndex.build(text, device=0)
        try:
            t = text.tobytes()
            guides, at = [fam.tobytes().decode(), synth.reverse_complement_bytes(fam).tobytes().decode()], 0
            while len(guides) < 26:   # guides read off the genome behind TTT sites: they hit
                at = t.find(b"TTT", at + 1)
                assert at >= 0
                w = t[at:at + 27]
                if len(w) == 27 and set(w) <= set(b"ACGT"):
                    guides.append(w[4:].decode())
                    at += 5000
            seqs = np.array([list(g.encode()) for g in guides], dtype=np.uint8)
            pams = np.tile(np.frombuffer(b"TTTN", np.uint8), (len(guides), 1))
            for m in ms:
                for alt in ((), ("TCTN",)):
                    opts = ol.make_opts(mismatches=m, alt_pams=alt, start=True)
                    offsets, hits, stats = gidx.enumerate(seqs, pams, mismatches=m, alt_pams=alt, start=True)
                    ctr = gidx.last_counters()
                    total = 0
                    for i, g in enumerate(guides):
                        exp, _ = oracle_hits_as_records(oidx, g, "TTTN", opts, 4, True)
                        assert gpu_hits_as_records(offsets, hits, i, g, 4, True) == exp, (seed, i, m, alt)
                        total += len(exp)
                    assert total > len(guides)
                    assert (ctr["items_two_sided"] > 0) == (table_k == 14 and m >= 1), (m, ctr["items_two_sided"])
                    if n_copies >= 5000:
                        pick = {k: ctr[k] for k in ("guides_redone", "ordered_in_tiles", "tile_ordering_gave_up", "matches_max_per_item",
                                                    "overflow_from_arena", "redo_ordered_device_wide")}
                        assert ctr["guides_redone"] >= 1 and ctr["ordered_in_tiles"] and ctr["matches_max_per_item"] > 4096, pick
            # the reference-order walk carries 52 path bits: it says so instead of returning a truncated key
            with pytest.raises(api.GsError) as e:
                gidx.enumerate(seqs[:2], pams[:2], mismatches=1, start=True, faithful=True)
            assert e.value.status == 3
        finally:
            gidx.close()
            oidx.close()
