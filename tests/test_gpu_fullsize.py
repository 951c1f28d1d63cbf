"""BASELINE configs 2-5 at their real sizes, on the GPU box (`pytest -m gpu`).

 * config 2: chr1-sized index (249 Mbp), 100,000 guides, <= 3 mismatches
 * config 3: hg38-sized index (3.09 Gbp, both strands in HBM), 1,000,000 guides, <= 3 mismatches
 * config 4: genome-wide NGG candidates of a chromosome, scanned on the device and enumerated from HBM
 * config 5: hg38-sized index at <= 6 mismatches with CFD / specificity

What is compared with what:
 * the timed fast path (two-sided seeding) vs one-sided seeding (GS_NO_BIDIR) vs the reference-order
   walk from the root (GS_FLAG_FAITHFUL_WALK): same bytes; keys ascending; every sampled guide's own
   site reported at distance 0 at its own coordinate and strand, and its text equals guide + PAM;
 * the suffix arrays against the genome text alone (gs_index_verify_sa: permutation test on every
   row, suffix order and BWT symbol on >= 10^7 sampled rows) - nothing borrowed from the GPU builder;
 * the product's CSV lines (device search + device scoring + gs_format_guide_scored) against the
   lines the REFERENCE ITSELF writes for the same guides: oracle/_ref/gs_ref_enumerate (the
   reference's index.hpp / process.hpp / printer.hpp compiled in place) run on this host over index
   files written through the compiled SDSL containers.  Done at hg38 size for sampled guides and
   genome-wide candidates at m = 3 and for config 5's depth at m = 6.

 * the repeat-rich path at size: tests/test_gpu_fullsize_rep.py (a module of its own: its hg38-sized
   index needs the HBM this module's holds).

The reference legs need oracle/_ref (prebuilt, travels with the snapshot); nothing reads
/root/reference.  The SDSL index files are written by background threads while the GPU tests run."""
import ctypes as C
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time
from importlib import import_module

import numpy as np
import pytest

import oracle_lib as ol

api = import_module("guidescan-cli_amd.api")
synth = import_module("guidescan-cli_amd.synth")

pytestmark = pytest.mark.gpu
SHIM = ol.ORACLE_DIR / "_ref" / "gs_ref_enumerate"
NGG = np.frombuffer(b"NGG", np.uint8)


def _hip():
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return hip


def device_result_to_torch(torch, hip, d_off, d_hits, n, n_hits):
    off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    hits = torch.empty((max(n_hits, 1), 2), dtype=torch.int64, device="cuda")
    assert hip.hipMemcpy(off.data_ptr(), d_off, 8 * (n + 1), 3) == 0
    if n_hits:
        assert hip.hipMemcpy(hits.data_ptr(), d_hits, 16 * n_hits, 3) == 0
    return off, hits[:n_hits]


def check_batch_properties(text, seqs, positions, strands, off, pos, key):
    """size-independent properties of one batch's CSR hit list (numpy arrays on the host)"""
    n = seqs.shape[0]
    assert off[0] == 0 and off[-1] == pos.shape[0] and np.all(np.diff(off) >= 0)
    seg = np.repeat(np.arange(n), np.diff(off))
    same = seg[1:] == seg[:-1]
    assert np.all(key[1:][same] >= key[:-1][same]), "keys not ascending within a guide"
    assert np.all((key & np.uint64(0xFF)) == 0)
    # every sampled site is reported at distance 0 at its own coordinate and strand: + strand sites
    # come from the reverse index with pos = p + 22 (inclusive end), - strand sites from the forward
    # index with pos = -p (process.hpp:104,111)
    want = np.where(strands == ord("+"), positions + 22, -positions)
    d0 = (key >> np.uint64(61)) == 0
    hit_is_own = d0 & (pos == want[seg])
    found = np.zeros(n, dtype=bool)
    found[seg[hit_is_own]] = True
    assert found.all(), f"{int((~found).sum())} guides lack their own site at distance 0"
    # the index bit agrees with the sign convention, and the text under every distance-0 hit of a
    # sample equals guide + xGG
    idx = np.nonzero(d0)[0][:: max(1, int(d0.sum()) // 20000)]
    rev = ((key[idx] >> np.uint64(60)) & np.uint64(1)).astype(bool)
    for h, r in zip(idx, rev):
        g = seqs[seg[h]]
        if r:
            w = text[pos[h] - 22:pos[h] + 1]
        else:
            w = synth.reverse_complement_bytes(text[-pos[h]:-pos[h] + 23])
        assert np.array_equal(w[:20], g) and w[21] == ord("G") and w[22] == ord("G")
    return int(idx.size)


def device_result_to_host(hip, d_off, d_hits, n, n_hits):
    off = np.empty(n + 1, dtype=np.int64)
    hits = np.empty((max(n_hits, 1), 2), dtype=np.int64)
    assert hip.hipMemcpy(off.ctypes.data, d_off, 8 * (n + 1), 2) == 0
    if n_hits:
        assert hip.hipMemcpy(hits.ctypes.data, d_hits, 16 * n_hits, 2) == 0
    return off, hits[:n_hits]


def three_paths_same_bytes(torch, gidx, d_seqs, d_pams, n, m, n_walk, on_host=False):
    """two-sided seeding == one-sided seeding (whole batch) == reference-order walk (first n_walk).
    on_host: the copies that are compared live in host memory (hit lists of several GB next to an index
    that fills the HBM)"""
    hip = _hip()
    if on_host:
        d_off, d_hits, st = gidx.enumerate_device(d_seqs.data_ptr(), n, 20, d_pams.data_ptr(), 3, mismatches=m)
        off, hits = device_result_to_host(hip, d_off, d_hits, n, st["n_hits"])
        gidx.set_option("GS_NO_BIDIR", "1")
        try:
            d_o, d_h, st2 = gidx.enumerate_device(d_seqs.data_ptr(), n, 20, d_pams.data_ptr(), 3, mismatches=m)
            o2, h2 = device_result_to_host(hip, d_o, d_h, n, st2["n_hits"])
        finally:
            gidx.set_option("GS_NO_BIDIR", None)
        assert np.array_equal(o2, off) and h2.tobytes() == hits.tobytes(), "one-sided and two-sided seeding differ"
        del o2, h2
        d_o, d_h, st3 = gidx.enumerate_device(d_seqs.data_ptr(), n_walk, 20, d_pams.data_ptr(), 3, mismatches=m,
                                              faithful=True)
        o3, h3 = device_result_to_host(hip, d_o, d_h, n_walk, st3["n_hits"])
        nh = int(off[n_walk])
        assert np.array_equal(o3, off[:n_walk + 1]) and h3.tobytes() == hits[:nh].tobytes(), "walk and fast path differ"
        return off, hits, st, st3
    d_off, d_hits, st = gidx.enumerate_device(d_seqs.data_ptr(), n, 20, d_pams.data_ptr(), 3, mismatches=m)
    off, hits = device_result_to_torch(torch, hip, d_off, d_hits, n, st["n_hits"])
    gidx.set_option("GS_NO_BIDIR", "1")
    try:
        d_o, d_h, st2 = gidx.enumerate_device(d_seqs.data_ptr(), n, 20, d_pams.data_ptr(), 3, mismatches=m)
        o2, h2 = device_result_to_torch(torch, hip, d_o, d_h, n, st2["n_hits"])
    finally:
        gidx.set_option("GS_NO_BIDIR", None)
    assert torch.equal(o2, off) and torch.equal(h2, hits), "one-sided and two-sided seeding differ"
    d_o, d_h, st3 = gidx.enumerate_device(d_seqs.data_ptr(), n_walk, 20, d_pams.data_ptr(), 3, mismatches=m,
                                          faithful=True)
    o3, h3 = device_result_to_torch(torch, hip, d_o, d_h, n_walk, st3["n_hits"])
    nh = int(off[n_walk].item())
    assert torch.equal(o3, off[:n_walk + 1]) and torch.equal(h3, hits[:nh]), "walk and fast path differ"
    return off.cpu().numpy(), hits.cpu().numpy(), st, st3


# ---- config 2 ------------------------------------------------------------------------------------

def test_config2_chr1_sized_100k_guides_m3():
    """BASELINE config 2 at its real size: 249 Mbp index, 100,000 guides, <= 3 mismatches"""
    import torch
    text, names, lengths = synth.make_genome([synth.CHR1_LENGTH], seed=1)
    gidx = api.GenomeIndex.build(text, device=0)
    try:
        for s in (0, 1):   # 2.5e8 rows: suffix order / BWT symbol on 4M sampled rows, permutation on all
            rep = gidx.verify_sa(text, strand=s, samples=1 << 22, seed=5 + s)
            assert rep["rows"] == text.shape[0] + 1 and rep["sampled"] == 1 << 22
            assert rep["not_permutation"] == rep["out_of_order"] == rep["undecided"] == rep["bwt_mismatch"] == 0, rep
            rep = gidx.verify_sa(text, strand=s, samples="all")   # ... and every adjacent pair by the linear-time rule
            assert rep["sampled"] == text.shape[0], rep
            assert rep["not_permutation"] == rep["out_of_order"] == rep["undecided"] == rep["bwt_mismatch"] == 0, rep
        n = 100_000
        seqs, pams, pos, strands = synth.sample_guides(text, n, seed=21)
        d_seqs, d_pams = torch.from_numpy(seqs).cuda(), torch.from_numpy(pams).cuda()
        off, hits, st, st_walk = three_paths_same_bytes(torch, gidx, d_seqs, d_pams, n, 3, n)
        assert st["n_hits"] == off[-1] >= n
        check_batch_properties(text, seqs, pos, strands, off, hits[:, 0], hits[:, 1].view(np.uint64))
        # host-pointer entry point: same bytes as the device-pointer one
        off_h, hits_h, _ = gidx.enumerate(seqs, pams, mismatches=3)
        assert np.array_equal(off_h.astype(np.int64), off)
        assert hits_h.tobytes() == np.ascontiguousarray(hits).tobytes()
        # oracle spot check on 64 guides (its index is built from the suffix arrays verified above)
        oidx = ol.OracleIndex(text, sa_provider=lambda s: gidx.suffix_array(s), nthreads=16)
        try:
            opts = ol.make_opts(mismatches=3)
            for i in range(0, n, n // 64):
                g = seqs[i].tobytes().decode()
                exp, ctr, raw = oidx.enumerate(g, "NGG", opts)
                ol.lib().gso_free(raw[0])
                got = [(int(h["pos"]), int(h["key"]) >> 61, (int(h["key"]) >> 60) & 1,
                        api.decode_sequence(g, 3, int(h["key"]))) for h in hits_h[off_h[i]:off_h[i + 1]]]
                assert got == [(e[0], e[1], e[2], e[3]) for e in exp], i
        finally:
            oidx.close()
    finally:
        gidx.close()


def test_chr1_sized_assembly_with_5000_n_runs(monkeypatch, capfd):
    """A scaffold-level assembly: 5,000 N runs in a chr1-sized genome (3 x 10^4 literal-N windows per strand), 3,000 of
    them single Ns planted where an NGG site has its N on them (the reference matches a PAM N against the genome's
    literal N, index.hpp:139-149).  The window list is indexed by 5-symbol chunks (one bucket read per chunk and
    item instead of 470 passes over the list): 4,000 guides - half of them at the planted sites - return the same
    bytes as the plain scan of the list, and 96 of them the oracle's hit lists."""
    import torch
    rng = np.random.default_rng(19)
    text, names, lengths = synth.make_genome([synth.CHR1_LENGTH], seed=1)
    text = text.copy()
    n_total = text.shape[0]
    ok = np.frombuffer(b"ACGT", np.uint8)
    planted = []
    at = 2_000_000
    for r in range(5000):
        at += int(rng.integers(20_000, 45_000))
        if r < 3000:   # 20-mer N GG on the + strand, or CC N 20-mer (a - strand site)
            text[at - 30:at + 30] = rng.choice(ok, 60)
            text[at] = ord("N")
            if r % 2 == 0:
                text[at + 1:at + 3] = np.frombuffer(b"GG", np.uint8)
                planted.append(text[at - 20:at].copy())
            else:
                text[at - 2:at] = np.frombuffer(b"CC", np.uint8)
                planted.append(synth.reverse_complement_bytes(text[at + 1:at + 21]).copy())
        else:
            text[at:at + int(rng.integers(2, 3000))] = ord("N")
    assert at < n_total - 10_000
    monkeypatch.setenv("GS_DEBUG", "1")
    gidx = api.GenomeIndex.build(text, device=0)
    try:
        sampled, _, _, _ = synth.sample_guides(text, 2000, seed=4)
        pick = rng.choice(len(planted), 2000, replace=False)
        seqs = np.concatenate([np.stack([planted[i] for i in pick]), sampled])
        for i in range(0, 2000, 3):   # some of the planted guides one or two substitutions away from their site
            for q in rng.choice(20, size=int(rng.integers(1, 3)), replace=False):
                seqs[i, q] = rng.choice([x for x in b"ACGT" if x != seqs[i, q]])
        pams = np.tile(NGG, (seqs.shape[0], 1))
        capfd.readouterr()
        off, hits, st = gidx.enumerate(seqs, pams, mismatches=3)
        err = capfd.readouterr().err
        assert "bucketed by 5-symbol chunks" in err, err[-400:]
        gidx.set_option("GS_NO_CAND_BUCKETS", "1")
        off2, hits2, st2 = gidx.enumerate(seqs, pams, mismatches=3)
        gidx.set_option("GS_NO_CAND_BUCKETS", None)
        assert "bucketed" not in capfd.readouterr().err
        assert np.array_equal(off, off2) and hits.tobytes() == hits2.tobytes()
        lit = 0
        oidx = ol.OracleIndex(text, sa_provider=lambda s: gidx.suffix_array(s), nthreads=16)
        try:
            opts = ol.make_opts(mismatches=3)
            for i in list(range(0, 2000, 42)) + list(range(2000, 4000, 42)):
                g = seqs[i].tobytes().decode()
                exp, ctr, raw = oidx.enumerate(g, "NGG", opts)
                ol.lib().gso_free(raw[0])
                got = [(int(h["pos"]), int(h["key"]) >> 61, (int(h["key"]) >> 60) & 1,
                        api.decode_sequence(g, 3, int(h["key"]))) for h in hits[off[i]:off[i + 1]]]
                assert got == [(e[0], e[1], e[2], e[3]) for e in exp], i
                lit += sum(1 for e in exp if e[3][20:].startswith("N"))
        finally:
            oidx.close()
        assert lit >= 40, lit   # the planted sites are found with the genome's N in their match sequence
    finally:
        gidx.close()


# ---- configs 3, 4, 5: one hg38-sized index for the whole group -------------------------------------

class Hg38:
    """the hg38-sized genome, its device index, and (in the background) the reference's index files"""

    def __init__(self, lengths=None, repeats=False):
        # torch's runtime first: initialised only after an index has filled the HBM it reported
        # "No HIP GPUs are available" on the GPU box
        import torch
        torch.zeros(1, device="cuda")
        t0 = time.time()
        lengths = synth.GRCH38_LENGTHS if lengths is None else lengths
        make = synth.make_repeat_genome if repeats else synth.make_genome
        self.text, self.names, self.lengths = make(lengths, seed=1)
        self.t_gen = time.time() - t0
        t0 = time.time()
        self.gidx = api.GenomeIndex.build(self.text, device=0)
        self.t_build = time.time() - t0
        # The reference's index files below are written from THIS suffix array, and the large oracle tests borrow it: it is
        # proved first, from the text alone - every adjacent pair of rows by the linear-time rule (text[SA[r]] < text[SA[r+1]],
        # or equal symbols and ISA[SA[r]+1] < ISA[SA[r+1]+1]), ISA the inverse of SA, every row's BWT symbol
        # (gs_index_verify_sa with GS_VERIFY_ALL_ROWS; csa_wt::operator[] presumes exactly this, csa_wt.hpp:333-346)
        t0 = time.time()
        self.sa_reports = [self.gidx.verify_sa(self.text, strand=s, samples="all") for s in (0, 1)]
        self.t_verify = time.time() - t0
        for rep in self.sa_reports:
            assert rep["rows"] == self.text.shape[0] + 1 and rep["sampled"] == self.text.shape[0], rep
            assert rep["not_permutation"] == rep["out_of_order"] == rep["undecided"] == rep["bwt_mismatch"] == 0, rep
        self.gs = api.make_genome_structure(self.names, self.lengths)
        self.dir = tempfile.mkdtemp(prefix="gs_full_")
        self.ref_ok = ol.ref() is not None and SHIM.exists()
        self.ref_err = []
        self.threads = []
        self.t_files = None
        if self.ref_ok:
            self._t0 = time.time()
            for strand, suffix in ((0, ".forward"), (1, ".reverse")):
                th = threading.Thread(target=self._write_strand, args=(strand, suffix))
                th.start()
                self.threads.append(th)
            with open(os.path.join(self.dir, "g.gs"), "w") as f:
                f.write("".join(f"{a}\n{b}\n" for a, b in zip(self.names, self.lengths)))

    def _write_strand(self, strand, suffix):
        """<prefix>.forward / .reverse through the compiled reference containers (ctypes releases the GIL)"""
        try:
            ref = ol.ref()
            n = self.text.shape[0] + 1
            sa = self.gidx.suffix_array(strand)
            st = np.ascontiguousarray(self.text if strand == 0 else synth.reverse_complement_bytes(self.text))
            h = ref.ref_index_build_text(st.ctypes.data, sa.ctypes.data, n,
                                         os.path.join(self.dir, f"tmp{strand}.sdsl").encode())
            assert ref.ref_write_index_file(h, os.path.join(self.dir, "g" + suffix).encode()) == 0
            ref.ref_index_free(h)
        except Exception as e:  # surfaced by reference_prefix()
            self.ref_err.append(repr(e))

    def reference_prefix(self):
        if not self.ref_ok:
            pytest.skip("oracle/_ref not built")
        for th in self.threads:
            th.join()
        if self.threads:
            self.t_files = time.time() - self._t0
            self.threads = []
        assert not self.ref_err, self.ref_err
        return os.path.join(self.dir, "g")

    def run_reference(self, tag, ids, seqs, m, threads, alt=(), in_order=False):
        """kmers CSV -> the compiled reference (CSV, complete) -> sorted data lines (in_order: as written;
        with one thread the reference writes guide after guide, each guide's rows in its own order)"""
        prefix = self.reference_prefix()
        kcsv, out = os.path.join(self.dir, tag + ".kmers.csv"), os.path.join(self.dir, tag + ".out.csv")
        synth.write_kmers_csv(kcsv, ids, [s.tobytes().decode() for s in seqs], ["NGG"] * len(ids),
                              [self.names[0]] * len(ids), [1] * len(ids), ["+"] * len(ids))
        env = dict(os.environ, GS_REF_THREADS=str(threads))
        subprocess.run([str(SHIM), prefix, kcsv, out, "csv", "complete", str(m), "0", "0", "-1", "-1", "0"] + list(alt),
                       env=env, check=True, timeout=1500, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        with open(out) as f:
            lines = f.read().splitlines()
        os.unlink(out)
        return lines[0], (lines[1:] if in_order else sorted(lines[1:]))

    def product_lines(self, ids, seqs, m, alt=(), in_order=False):
        """the same guides through the product: search, device scoring, text lines -> sorted data lines"""
        n = len(ids)
        pams = np.tile(NGG, (n, 1))
        off, hits, st = self.gidx.enumerate(seqs, pams, mismatches=m, alt_pams=alt)
        _, spec = self.gidx.score(self.gs, seqs, 3, off, hits, want_cfd=False)
        out = []
        for i in range(n):
            txt = api.format_guide(self.gs, ids[i], seqs[i].tobytes().decode(), "NGG", True,
                                   hits[off[i]:off[i + 1]], m, specificity=spec[i])
            out += txt.splitlines()
        return (out if in_order else sorted(out)), int(off[-1])

    def close(self):
        for th in self.threads:
            th.join()
        self.gidx.close()
        shutil.rmtree(self.dir, ignore_errors=True)


@pytest.fixture(scope="module")
def hg38():
    h = Hg38()
    yield h
    h.close()


def test_hg38_suffix_arrays_against_the_text_alone(hg38):
    """n = 3.09e9 > 2^31: both suffix arrays are permutations, 2^24 (1.7e7) evenly spread adjacent
    row pairs per strand are in suffix order by direct text comparison, and the Occ blocks hold the
    BWT symbol text[SA[r]-1] on those rows"""
    assert len(hg38.sa_reports) == 2   # (the fixture has checked EVERY pair of rows by the linear-time rule already)
    for s in (0, 1):
        rep = hg38.gidx.verify_sa(hg38.text, strand=s, samples=1 << 24, seed=11 + s)
        assert rep["rows"] == hg38.text.shape[0] + 1 and rep["sampled"] == 1 << 24
        assert rep["not_permutation"] == 0 and rep["out_of_order"] == 0 and rep["bwt_mismatch"] == 0, rep
        assert rep["undecided"] == 0, rep


def test_config3_hg38_sized_1M_guides_m3(hg38):
    """BASELINE config 3: 1,000,000 guides, <= 3 mismatches: the timed path, one-sided seeding and
    the reference-order walk (first 20,000 guides) return the same bytes; order and own-site properties"""
    import torch
    n = 1_000_000
    seqs, pams, pos, strands = synth.sample_guides(hg38.text, n, seed=1000)
    d_seqs, d_pams = torch.from_numpy(seqs).cuda(), torch.from_numpy(pams).cuda()
    off, hits, st, st_walk = three_paths_same_bytes(torch, hg38.gidx, d_seqs, d_pams, n, 3, 20_000)
    assert st["n_hits"] == off[-1] >= n
    checked = check_batch_properties(hg38.text, seqs, pos, strands, off, hits[:, 0], hits[:, 1].view(np.uint64))
    assert checked >= 10_000
    hg38.sampled = (seqs, pos, strands)


def test_config3_and_4_hg38_lines_equal_the_compiled_reference(hg38):
    """hg38-sized parity against process.hpp:35-128 itself: 4,096 sampled guides (config 3) and 512
    genome-wide candidates (config 4: NGG sites of a 46.7 Mbp chromosome scanned on the device and
    enumerated from HBM), <= 3 mismatches: every CSV line the reference writes, specificity included"""
    import torch
    text = hg38.text
    # config 4 on one GPU: candidates of chr21 (index 20), from HBM, every one finds itself
    c = 20
    c_off = int(sum(hg38.lengths[:c]))
    d_chr = torch.from_numpy(np.ascontiguousarray(text[c_off:c_off + hg38.lengths[c]])).cuda()
    km = api.generate_kmers(None, "NGG", 20, device=0, chrm_device_ptr=d_chr.data_ptr(), chrm_len=hg38.lengths[c])
    try:
        assert km.n > 4_000_000
        kseqs, kpams, kpos, ksense = km.to_host()
        hip = _hip()
        total_hits, batch = 0, 1 << 20
        for b0 in range(0, km.n, batch):
            nb = min(batch, km.n - b0)
            d_off, d_hits, st = hg38.gidx.enumerate_device(km.seqs_ptr + b0 * 20, nb, 20, km.pams_ptr + b0 * 3, 3,
                                                           mismatches=3)
            off, hits = device_result_to_torch(torch, hip, d_off, d_hits, nb, st["n_hits"])
            assert bool((off[1:] > off[:-1]).all()), "a candidate without any hit (not even itself)"
            total_hits += st["n_hits"]
            if b0 == 0:   # own-site property of the first batch (positions are 1-based, chromosome-relative)
                p0 = c_off + kpos[:nb].astype(np.int64) - 1
                check_batch_properties(text, kseqs[:nb], p0, ksense[:nb], off.cpu().numpy(),
                                       hits[:, 0].cpu().numpy(), hits[:, 1].cpu().numpy().view(np.uint64))
        assert total_hits >= km.n
    finally:
        km.close()
    del d_chr
    rng = np.random.default_rng(4)
    pick = np.sort(rng.choice(kseqs.shape[0], 512, replace=False))
    seqs_s, _, _, _ = synth.sample_guides(text, 4096, seed=1001)
    seqs = np.concatenate([seqs_s, kseqs[pick]])
    ids = [f"s{i}" for i in range(4096)] + [f"chr21:{int(kpos[j])}:{chr(ksense[j])}" for j in pick]
    got, n_hits = hg38.product_lines(ids, seqs, 3)
    header, want = hg38.run_reference("m3", ids, seqs, 3, os.cpu_count() or 8)
    assert header.startswith("id,sequence,")
    assert len(want) >= len(ids) and len(got) == len(want)
    assert got == want
    # row ORDER at this size, not only the multiset: with one thread the reference writes the guides in input
    # order and each guide's rows as its std::set / resolve loops yield them (process.hpp:100-115) - the
    # product's file for the same 128 guides is that file, byte for byte
    got1, _ = hg38.product_lines(ids[:128], seqs[:128], 3, in_order=True)
    header1, want1 = hg38.run_reference("m3n1", ids[:128], seqs[:128], 3, 1, in_order=True)
    assert got1 == want1


def test_config5_hg38_depth_m6_lines_equal_the_compiled_reference(hg38):
    """BASELINE config 5's depth: <= 6 mismatches + CFD at hg38 size, 64 guides: ~1e4 hits per guide;
    every CSV line (coordinates, match sequences, distances, specificity from k_score) equals the
    reference's"""
    seqs, _, _, _ = synth.sample_guides(hg38.text, 64, seed=1000)
    ids = [f"d{i}" for i in range(64)]
    got, n_hits = hg38.product_lines(ids, seqs, 6)
    assert n_hits > 64 * 5000
    header, want = hg38.run_reference("m6", ids, seqs, 6, 64)
    assert len(got) == len(want)
    assert got == want


def test_config5_hg38_m6_2048_guides_properties_and_paths(hg38):
    """BASELINE config 5 beyond the 64 guides the reference leg can afford: 2,048 guides at <= 6 mismatches
    (2.2 x 10^7 hits) + CFD.  Every guide through the device-wide ordering (12 k slots per item) and - the
    same bytes - through one-sided seeding; own site at distance 0, keys ascending, a specificity in (0, 1]
    for every guide, and the first 64 guides' lists equal the lists of the 64-guide batch the reference
    leg compares line by line (a batch's composition must not change a guide's result)"""
    import torch
    n = 2048
    seqs, pams, pos, strands = synth.sample_guides(hg38.text, n, seed=1000)
    d_seqs, d_pams = torch.from_numpy(seqs).cuda(), torch.from_numpy(pams).cuda()
    hip = _hip()
    d_off, d_hits, st = hg38.gidx.enumerate_device(d_seqs.data_ptr(), n, 20, d_pams.data_ptr(), 3, mismatches=6)
    ctr = hg38.gidx.last_counters()
    assert ctr["ordered_device_wide"] and ctr["ordered_in_tiles"] and not ctr["tile_ordering_gave_up"], ctr
    off, hits = device_result_to_host(hip, d_off, d_hits, n, st["n_hits"])
    assert st["n_hits"] > n * 5000
    check_batch_properties(hg38.text, seqs, pos, strands, off, hits[:, 0], hits[:, 1].view(np.uint64))
    d_spec = torch.empty(n, dtype=torch.float32, device="cuda")
    hg38.gidx.score_device(hg38.gs, d_seqs.data_ptr(), n, 20, 3, d_off, d_hits, None, d_spec.data_ptr())
    spec = d_spec.cpu().numpy()
    assert np.all(np.isfinite(spec)) and np.all(spec > 0) and np.all(spec <= 1.0)
    hg38.gidx.set_option("GS_NO_BIDIR", "1")
    try:
        d_o, d_h, st2 = hg38.gidx.enumerate_device(d_seqs.data_ptr(), n, 20, d_pams.data_ptr(), 3, mismatches=6)
        o2, h2 = device_result_to_host(hip, d_o, d_h, n, st2["n_hits"])
    finally:
        hg38.gidx.set_option("GS_NO_BIDIR", None)
    assert np.array_equal(o2, off) and h2.tobytes() == hits.tobytes(), "one-sided and two-sided seeding differ"
    # the 64-guide batch of the reference leg (same seed: the same first guides)
    off64, hits64, _ = hg38.gidx.enumerate(seqs[:64], pams[:64], mismatches=6)
    assert np.array_equal(off64.astype(np.int64), off[:65])
    assert hits64.tobytes() == np.ascontiguousarray(hits[:int(off[64])]).tobytes()
    _, spec64 = hg38.gidx.score(hg38.gs, seqs[:64], 3, off64, hits64, want_cfd=False)
    assert np.array_equal(np.asarray(spec64, dtype=np.float32), spec[:64])


def test_hg38_cas12a_and_the_general_path_equal_the_compiled_reference(hg38):
    """What runs beside the NGG headline, pinned at hg38 size against the compiled reference (tools/general_bench.py
    is the same comparison with larger batches and rates): (1) Cas12a - TTTN at the 5' end (--start) - 20-mers
    (two-sided seeding through a four-symbol PAM's pair table) and 23-mers (58 key bits: the wide key, one-sided), 1,024
    guides each; (2) the general path: 256 guides at -m 1 --rna-bulges 1 --dna-bulges 1 (index.hpp:250-375) and 256
    guides with an N in them at -m 3 (index.hpp:218-247).  Every CSV line equals the reference's."""
    sys.path.insert(0, str(ol.ROOT / "tools"))
    gb = import_module("general_bench")
    for L in (20, 23):
        n = 1024
        seqs, pams = gb.cas_guides(synth, hg38.text, n, L, 60 + L)
        ids = [f"c{L}_{i}" for i in range(n)]
        want, _ = gb.ref_lines(hg38, f"tcas{L}", [(ids[i], seqs[i].tobytes().decode(), "TTTN") for i in range(n)], 3, start=True)
        off, hits, st = hg38.gidx.enumerate(seqs, pams, mismatches=3, start=True)
        _, spec = hg38.gidx.score(hg38.gs, seqs, 4, off, hits, want_cfd=False, start=True)
        got = []
        for i in range(n):
            got += api.format_guide(hg38.gs, ids[i], seqs[i].tobytes().decode(), "TTTN", True, hits[off[i]:off[i + 1]], 3,
                                    specificity=spec[i], start=True).splitlines()
        assert len(want) >= n and sorted(got) == want, L
    n = 256
    seqs, pams, _, _ = synth.sample_guides(hg38.text, n, seed=77)
    odd = seqs.copy()
    rng = np.random.default_rng(5)
    for i in range(n):
        odd[i, int(rng.integers(0, 20))] = ord("N")
    for tag, s, m, rna, dna in (("tbul", seqs, 1, 1, 1), ("todd", odd, 3, 0, 0)):
        ids = [f"{tag}{i}" for i in range(n)]
        want, _ = gb.ref_lines(hg38, tag, [(ids[i], s[i].tobytes().decode(), "NGG") for i in range(n)], m, rna, dna)
        off, hx = hg38.gidx.enumerate_general(s, pams, mismatches=m, rna_bulges=rna, dna_bulges=dna)
        got = []
        for i in range(n):
            got += api.format_guide_ex(hg38.gs, ids[i], s[i].tobytes().decode(), "NGG", True, hx[off[i]:off[i + 1]], m).splitlines()
        assert len(want) >= n // 2 and sorted(got) == want, tag


def test_hg38_alt_pam_through_two_pair_tables_equals_the_compiled_reference(hg38):
    """`-a NAG` at hg38 size: the guides' own NGG and the alt pattern end in different pairs of bases, so
    the batch needs two PAM-pair tables (with their deep tables) next to the 184 GB of strand tables -
    they share what is free; every CSV line equals the reference's"""
    seqs, _, _, _ = synth.sample_guides(hg38.text, 1024, seed=1002)
    ids = [f"a{i}" for i in range(1024)]
    got, n_hits = hg38.product_lines(ids, seqs, 3, alt=("NAG",))
    cnt = hg38.gidx.last_counters()
    assert cnt["items_pair_tables"] >= 2 * 1024, cnt   # every (guide, strand) item went through the tables
    header, want = hg38.run_reference("m3nag", ids, seqs, 3, os.cpu_count() or 8, alt=("NAG",))
    assert len(want) >= len(ids) and len(got) == len(want)
    assert got == want



def test_hg38_out_of_memory_drops_derived_tables_and_redoes_the_batch(hg38):
    """the strand tables' rotated copies (86 GB, built by the one-sided runs above) and the PAM-pair / deep
    tables (26 GB) are derived data: a batch whose workspace no longer fits drops them - the rotated copies
    first, the pair tables next - and is redone without them: same bytes, no error.  Forced here by ballast
    allocations that leave 1 GB.  (Last test of the module: the handle goes on without them afterwards.)"""
    import torch
    n = 260_000
    seqs, pams, pos, strands = synth.sample_guides(hg38.text, n, seed=1003)
    d_seqs, d_pams = torch.from_numpy(seqs).cuda(), torch.from_numpy(pams).cuda()
    hip = _hip()
    n4 = 200_000
    d_off, d_hits, st = hg38.gidx.enumerate_device(d_seqs.data_ptr(), n4, 20, d_pams.data_ptr(), 3, mismatches=4)
    assert hg38.gidx.last_counters()["items_pair_tables"] > 0
    off, hits = device_result_to_host(hip, d_off, d_hits, n4, st["n_hits"])

    def starved(n_guides):
        """one m <= 5 call with 1 GB of HBM left; returns the bytes the handle gave back"""
        before = hg38.gidx.device_bytes
        free, total = torch.cuda.mem_get_info()
        ballast = torch.empty(max(0, free - (1 << 30)), dtype=torch.uint8, device="cuda")
        try:
            _, _, st5 = hg38.gidx.enumerate_device(d_seqs.data_ptr(), n_guides, 20, d_pams.data_ptr(), 3, mismatches=5)
            assert st5["n_hits"] > 1000 * n_guides
        finally:
            del ballast
            torch.cuda.empty_cache()
        return before - hg38.gidx.device_bytes

    freed1 = starved(200_000)          # its slots alone are 11 GB: something has to go
    assert freed1 > 20e9, freed1
    freed2 = starved(260_000)          # a larger batch again: the next kind of table goes
    assert freed1 + freed2 > 100e9, (freed1, freed2)   # both kinds are gone: 86 GB + 26 GB
    # and the m <= 4 batch, now through the strand tables alone, returns the same bytes as before
    d_o, d_h, st3 = hg38.gidx.enumerate_device(d_seqs.data_ptr(), n4, 20, d_pams.data_ptr(), 3, mismatches=4)
    assert hg38.gidx.last_counters()["items_pair_tables"] == 0
    o3, h3 = device_result_to_host(hip, d_o, d_h, n4, st3["n_hits"])
    assert np.array_equal(o3, off) and h3.tobytes() == hits.tobytes()


# ---- configs 4 and 5 beyond the reference leg's reach: committed checksums -------------------------------------
# The sums below were written by this test's own first run on an MI355X box (round 5) - the same batches whose sampled
# guides the tests above compare with the compiled reference line by line - and pin every offset, every hit record and
# every specificity bit of the runs: a change in any kernel that moves one hit shows here.
CONFIG4_TWO_CHROMOSOMES = {"candidates": 8_493_342, "hits": 110_009_853, "checksum": "9de46e73bf38e5e5"}
CONFIG5_STREAM_100K = {"hits": 1_079_231_985, "checksum": "1d26cdb4bdf31a49"}


def fold_batch(torch, hip, csum, d_off, d_hits, n, n_hits, d_spec=None):
    """tools/config5_stream.py's checksum: hit records folded with their place, then the offsets and specificity bits"""
    off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    assert hip.hipMemcpy(off.data_ptr(), d_off, 8 * (n + 1), 3) == 0
    step = 1 << 26
    for h0 in range(0, n_hits, step):
        m = min(step, n_hits - h0)
        piece = torch.empty((m, 2), dtype=torch.int64, device="cuda")
        assert hip.hipMemcpy(piece.data_ptr(), d_hits + 16 * h0, 16 * m, 3) == 0
        w = torch.arange(h0, h0 + m, dtype=torch.int64, device="cuda") * 0x9E3779B1 + 12345
        csum = (csum + int(((piece[:, 0] ^ piece[:, 1]) * w).sum().item())) & 0xFFFFFFFFFFFFFFFF
        del piece, w
    tail = int(d_spec.view(torch.int32).long().sum().item()) if d_spec is not None else 0
    csum = (csum * 1099511628211 + int((off * torch.arange(1, n + 2, device="cuda")).sum().item()) + tail) & 0xFFFFFFFFFFFFFFFF
    del off
    return csum


def test_config4_two_chromosomes_and_config5_stream_checksums(hg38):
    """BASELINE config 4 on two chromosomes (every NGG candidate of chr21 and chr22 scanned on the device, 8.5 x 10^6
    guides, enumerated from HBM at <= 3 mismatches in batches of 2^20) and config 5 as a stream (100,000 sampled guides at
    <= 6 mismatches + CFD in batches of 20,000): hit totals and checksums equal the committed ones, every candidate finds
    itself, and 128 candidates of chr22 give the compiled reference's CSV lines (chr21's and the sampled guides' lines are
    compared above)."""
    import torch
    hip = _hip()
    text = hg38.text
    csum, total_hits, n_cand = 0, 0, 0
    sub_seqs, sub_ids = None, None
    for c in (20, 21):
        c_off = int(sum(hg38.lengths[:c]))
        d_chr = torch.from_numpy(np.ascontiguousarray(text[c_off:c_off + hg38.lengths[c]])).cuda()
        km = api.generate_kmers(None, "NGG", 20, device=0, chrm_device_ptr=d_chr.data_ptr(), chrm_len=hg38.lengths[c])
        try:
            n_cand += km.n
            for b0 in range(0, km.n, 1 << 20):
                nb = min(1 << 20, km.n - b0)
                d_off, d_hits, st = hg38.gidx.enumerate_device(km.seqs_ptr + b0 * 20, nb, 20, km.pams_ptr + b0 * 3, 3, mismatches=3)
                assert st["n_hits"] >= nb
                csum = fold_batch(torch, hip, csum, d_off, d_hits, nb, st["n_hits"])
                total_hits += st["n_hits"]
            if c == 21:
                kseqs, kpams, kpos, ksense = km.to_host()
                pick = np.sort(np.random.default_rng(8).choice(kseqs.shape[0], 128, replace=False))
                sub_seqs = kseqs[pick]
                sub_ids = [f"chr22:{int(kpos[j])}:{chr(ksense[j])}" for j in pick]
        finally:
            km.close()
        del d_chr
    got4 = {"candidates": n_cand, "hits": total_hits, "checksum": f"{csum:016x}"}
    # config 5's stream
    n, batch = 100_000, 20_000
    seqs, pams, _, _ = synth.sample_guides(text, n, seed=1000)
    d_s, d_p = torch.from_numpy(seqs).cuda(), torch.from_numpy(pams).cuda()
    csum5, hits5 = 0, 0
    for lo in range(0, n, batch):
        d_off, d_hits, st = hg38.gidx.enumerate_device(d_s.data_ptr() + lo * 20, batch, 20, d_p.data_ptr() + lo * 3, 3, mismatches=6)
        d_spec = torch.empty(batch, dtype=torch.float32, device="cuda")
        hg38.gidx.score_device(hg38.gs, d_s.data_ptr() + lo * 20, batch, 20, 3, d_off, d_hits, None, d_spec.data_ptr())
        csum5 = fold_batch(torch, hip, csum5, d_off, d_hits, batch, st["n_hits"], d_spec)
        hits5 += st["n_hits"]
        del d_spec
    got5 = {"hits": hits5, "checksum": f"{csum5:016x}"}
    print("config 4 (chr21 + chr22):", got4, " config 5 stream (100 k):", got5)
    assert got4 == CONFIG4_TWO_CHROMOSOMES, (got4, got5)
    assert got5 == CONFIG5_STREAM_100K, (got4, got5)
    got, _ = hg38.product_lines(sub_ids, sub_seqs, 3)
    header, want = hg38.run_reference("c22", sub_ids, sub_seqs, 3, os.cpu_count() or 8)
    assert len(got) == len(want) and got == want


CONFIG4_WHOLE_GENOME = {"candidates": 269_049_665, "hits": 3_485_210_395, "checksum": "08fc836fc587650b"}


def test_config4_every_candidate_of_the_genome_on_one_gpu(hg38):
    """BASELINE config 4 AT ITS SIZE on one GPU (15 s of wall time in all since the index builds in 8): every NGG candidate
    of all 24 chromosomes scanned on the device (gs_kmers_generate), enumerated from HBM at <= 3 mismatches in batches of
    2^20 and scored (CFD + specificity) - 2.69 x 10^8 guides, 3.49 x 10^9 hits.  Candidates and hits equal what
    tools/genomewide_enumerate.py has printed since round 4 (profiles/r0*_hg38_genomewide.json.log); the checksum over every
    offset, hit record and specificity is this path's own first run - a tripwire at the stated size, not a second opinion:
    correctness at size rests on the lines compared with the compiled reference above (4,096 + 128 candidates, 512 + 64
    guides at depth 6) and on the properties checked here: every candidate finds its own site, hits per guide >= 1."""
    import torch
    hip = _hip()
    text = hg38.text
    csum, total_hits, n_cand = 0, 0, 0
    spec_sum = 0.0
    t0 = time.time()
    for c in range(len(hg38.lengths)):
        c_off = int(sum(hg38.lengths[:c]))
        d_chr = torch.from_numpy(np.ascontiguousarray(text[c_off:c_off + hg38.lengths[c]])).cuda()
        km = api.generate_kmers(None, "NGG", 20, device=0, chrm_device_ptr=d_chr.data_ptr(), chrm_len=hg38.lengths[c])
        try:
            n_cand += km.n
            for b0 in range(0, km.n, 1 << 20):
                nb = min(1 << 20, km.n - b0)
                d_off, d_hits, st = hg38.gidx.enumerate_device(km.seqs_ptr + b0 * 20, nb, 20, km.pams_ptr + b0 * 3, 3, mismatches=3)
                assert st["n_hits"] >= nb
                d_spec = torch.empty(nb, dtype=torch.float32, device="cuda")
                hg38.gidx.score_device(hg38.gs, km.seqs_ptr + b0 * 20, nb, 20, 3, d_off, d_hits, None, d_spec.data_ptr())
                csum = fold_batch(torch, hip, csum, d_off, d_hits, nb, st["n_hits"], d_spec)
                spec_sum += float(d_spec.double().sum().item())
                assert float(d_spec.min().item()) > 0.0 and float(d_spec.max().item()) <= 1.0
                total_hits += st["n_hits"]
                del d_spec
        finally:
            km.close()
        del d_chr
    got = {"candidates": n_cand, "hits": total_hits, "checksum": f"{csum:016x}"}
    print("config 4, the whole genome:", got, f"mean specificity {spec_sum / n_cand:.10f}, {time.time() - t0:.1f} s")
    assert abs(spec_sum / n_cand - 0.3665231549) < 1e-6   # (tools/genomewide_enumerate.py --score, rounds 4 and 6)
    assert got == CONFIG4_WHOLE_GENOME, got


CONFIG5_ONE_MILLION = {"hits": 10_806_215_010, "checksum": "6817fb7783584b19"}


def test_config5_one_million_guides_at_depth_6_with_cfd(hg38):
    """BASELINE config 5 AT ITS SIZE on one GPU: 1,000,000 sampled guides at <= 6 mismatches with CFD + specificity, as a
    stream of batches of 20,000 whose hit lists (2 x 10^8 hits = 3.5 GB each) stay in HBM - 1.08 x 10^10 hits in ~4 s.  The hit
    total equals what tools/config5_stream.py has printed (profiles/r06_hg38_config5_stream.json.log, same seed); the checksum
    over every offset, hit record and specificity is this path's own first run - a tripwire at the stated size: correctness
    at depth 6 rests on the 512 + 64 guides compared with the compiled reference above and on the properties checked here."""
    import torch
    hip = _hip()
    n, batch = 1_000_000, 20_000
    seqs, pams, _, _ = synth.sample_guides(hg38.text, n, seed=1000)
    d_s, d_p = torch.from_numpy(seqs).cuda(), torch.from_numpy(pams).cuda()
    csum, hits = 0, 0
    t0 = time.time()
    for lo in range(0, n, batch):
        d_off, d_hits, st = hg38.gidx.enumerate_device(d_s.data_ptr() + lo * 20, batch, 20, d_p.data_ptr() + lo * 3, 3, mismatches=6)
        assert st["n_hits"] >= batch   # every sampled guide finds its own site
        d_spec = torch.empty(batch, dtype=torch.float32, device="cuda")
        hg38.gidx.score_device(hg38.gs, d_s.data_ptr() + lo * 20, batch, 20, 3, d_off, d_hits, None, d_spec.data_ptr())
        assert float(d_spec.min().item()) > 0.0 and float(d_spec.max().item()) <= 1.0
        csum = fold_batch(torch, hip, csum, d_off, d_hits, batch, st["n_hits"], d_spec)
        hits += st["n_hits"]
        del d_spec
    got = {"hits": hits, "checksum": f"{csum:016x}"}
    print("config 5, one million guides:", got, f"{time.time() - t0:.1f} s")
    assert got == CONFIG5_ONE_MILLION, got


def test_hg38_reference_index_files_open_through_the_importer(hg38):
    """SURVEY 8a row a12 at the size users download it: the <prefix>.forward / .reverse files the reference leg reads
    (csa_wt<wt_huff<>,64,8192>::serialize, sdsl/include/sdsl/csa_wt.hpp:372-391; 1.48 GB per strand, n > 2^31, 32-bit-wide
    samples) opened through gs_index_open_sdsl - wavelet tree expanded to the BWT, inverted to the text on the host's
    threads, device layout rebuilt - give the SAME hit bytes as the index built from the text, for 4,096 guides.
    LAST in this module: the built index is closed first (two hg38-sized indexes do not share the HBM with their
    tables and workspace); the fixture's teardown closes the imported one."""
    prefix = hg38.reference_prefix()
    seqs, pams, _, _ = synth.sample_guides(hg38.text, 4096, seed=1001)
    off, hits, _ = hg38.gidx.enumerate(seqs, pams, mismatches=3)
    n_bytes = hg38.gidx.device_bytes
    hg38.gidx.close()
    t0 = time.time()
    hg38.gidx = api.GenomeIndex.open_sdsl(prefix, device=0)
    t_import = time.time() - t0
    print(f"gs_index_open_sdsl at hg38 size: {t_import:.1f} s ({os.path.getsize(prefix + '.forward') / 1e9:.2f} GB per strand)")
    assert hg38.gidx.genome_length == hg38.text.shape[0]
    off2, hits2, _ = hg38.gidx.enumerate(seqs, pams, mismatches=3)
    assert np.array_equal(off, off2) and hits.tobytes() == hits2.tobytes()
    assert t_import < 600 and n_bytes > 0
