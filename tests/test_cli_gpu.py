"""End to end through the C++ host that keeps the reference's command line: `guidescan index` +
`guidescan enumerate` on the toy genome must write the reference's output files byte for byte
(tests/golden/toy/ref_*, -n 1 order).  GPU only."""
import os
import subprocess
from importlib import import_module

import numpy as np
import pytest

import oracle_lib as ol

api = import_module("guidescan-cli_amd.api")

pytestmark = pytest.mark.gpu
CLI = ol.ROOT / "guidescan-cli_amd" / "bin" / "guidescan"

RUNS = {
    "m0_csv": ["-m", "0"], "m1_csv": ["-m", "1"], "m2_csv": ["-m", "2"], "m3_csv": ["-m", "3"],
    "m4_csv": ["-m", "4"],
    "m3_sam": ["-m", "3", "--format", "sam"],
    "m2_sam_succinct": ["-m", "2", "--format", "sam", "--mode", "succinct"],
    "m3_csv_succinct": ["-m", "3", "--mode", "succinct"],
    "m3_csv_nag": ["-m", "3", "-a", "NAG"],
    "m3_sam_nag": ["-m", "3", "-a", "NAG", "--format", "sam"],
    "m3_csv_max2": ["-m", "3", "--max-off-targets", "2"],
    "m3_sam_max2": ["-m", "3", "--max-off-targets", "2", "--format", "sam"],
    "m2_csv_start": ["-m", "2", "--start"],
    "m2_csv_t1": ["-m", "2", "-t", "1"],
    "m1_csv_rna1": ["-m", "1", "--rna-bulges", "1"],
    "m1_csv_dna1": ["-m", "1", "--dna-bulges", "1"],
    "m2_csv_rna1_dna1": ["-m", "2", "--rna-bulges", "1", "--dna-bulges", "1"],
    "m1_sam_rna1_dna1": ["-m", "1", "--rna-bulges", "1", "--dna-bulges", "1", "--format", "sam"],
    "m0_csv_rna2_dna2": ["-m", "0", "--rna-bulges", "2", "--dna-bulges", "2"],
    "m1_csv_dna1_nag_start": ["-m", "1", "--dna-bulges", "1", "-a", "NAG", "--start"],
}


@pytest.fixture(scope="module")
def indexed(toy, tmp_path_factory):
    d = tmp_path_factory.mktemp("cli")
    subprocess.run([str(CLI), "index", "--index", str(d / "toy"), str(toy["dir"] / "toy.fa")], check=True,
                   timeout=120)
    assert (d / "toy.gs").read_text() == (toy["dir"] / "toy.gs").read_text()
    return d


@pytest.mark.parametrize("name", sorted(RUNS))
def test_cli_output_is_byte_identical(toy, indexed, name):
    ext = "sam" if "sam" in name else "csv"
    out = indexed / f"{name}.{ext}"
    subprocess.run([str(CLI), "enumerate", str(indexed / "toy"), "-f", str(toy["dir"] / "kmers.csv"),
                    "-o", str(out), "-n", "1"] + RUNS[name], check=True, timeout=300)
    ref = toy["dir"] / f"ref_{name}.{ext}"
    if ref.exists():
        assert out.read_bytes() == ref.read_bytes()
    else:  # large reference outputs are committed as a digest
        import hashlib
        digest, size = (toy["dir"] / f"ref_{name}.{ext}.sha256").read_text().split()
        data = out.read_bytes()
        assert len(data) == int(size) and hashlib.sha256(data).hexdigest() == digest


def test_cli_rejects_bad_usage(toy, indexed):
    r = subprocess.run([str(CLI), "enumerate", str(indexed / "toy"), "-f", str(toy["dir"] / "kmers.csv"),
                        "-o", str(indexed / "x.csv"), "--format", "vcf"], timeout=60)
    assert r.returncode == 2


def test_cli_reads_reference_index_files(toy, tmp_path):
    """an index made by the reference itself (toy.idx.forward + .gs) works as is"""
    import shutil
    for f in ("toy.idx.forward", "toy.idx.reverse"):
        shutil.copy(toy["dir"] / f, tmp_path / f)
    shutil.copy(toy["dir"] / "toy.gs", tmp_path / "toy.idx.gs")
    out = tmp_path / "o.csv"
    subprocess.run([str(CLI), "enumerate", str(tmp_path / "toy.idx"), "-f", str(toy["dir"] / "kmers.csv"),
                    "-o", str(out), "-m", "3"], check=True, timeout=300)
    assert out.read_bytes() == (toy["dir"] / "ref_m3_csv.csv").read_bytes()


@pytest.mark.parametrize("name,args", [("m1_csv", ["-m", "1"]), ("m3_csv", ["-m", "3"]),
                                       ("m3_sam", ["-m", "3", "--format", "sam"])])
def test_config1_cli_byte_identical_to_reference(tmp_path, name, args):
    """BASELINE config 1 end to end: sacCer3-sized genome (regenerated from its seed), 1,000
    guides: the CLI's output file equals the reference binary's (tests/golden/config1)"""
    from importlib import import_module
    synth = import_module("guidescan-cli_amd.synth")
    gold = ol.ROOT / "tests" / "golden" / "config1"
    text, names, lengths = synth.make_genome(synth.SACCER3_LENGTHS, seed=1, probs=(.31, .19, .19, .31))
    text.tofile(tmp_path / "g.dna")
    (tmp_path / "g.gs").write_text("".join(f"{a}\n{b}\n" for a, b in zip(names, lengths)))
    ext = "sam" if "sam" in name else "csv"
    out = tmp_path / f"o.{ext}"
    subprocess.run([str(CLI), "enumerate", str(tmp_path / "g"), "-f", str(gold / "kmers.csv"), "-o", str(out)]
                   + args, check=True, timeout=600)
    assert out.read_bytes() == (gold / f"ref_{name}.{ext}").read_bytes()


def test_stored_suffix_arrays_skip_the_sort(tmp_path):
    """`guidescan index --store-sa` writes PREFIX.sa; `enumerate` then opens the index from text +
    stored suffix arrays (no sort) and writes the same file; a PREFIX.sa of another genome is ignored"""
    import shutil
    d = ol.ROOT / "tests" / "golden" / "toy"
    shutil.copy(d / "toy.fa", tmp_path / "toy.fa")
    subprocess.run([str(CLI), "index", "--index", str(tmp_path / "t"), "--store-sa", str(tmp_path / "toy.fa")], check=True,
                   timeout=300)
    assert (tmp_path / "t.sa").stat().st_size == 64 + 2 * 4 * ((tmp_path / "t.dna").stat().st_size + 1)
    out1, out2, out3 = tmp_path / "o1.csv", tmp_path / "o2.csv", tmp_path / "o3.csv"
    run = lambda prefix, out: subprocess.run([str(CLI), "enumerate", str(prefix), "-f", str(d / "kmers.csv"), "-o", str(out),
                                              "-m", "3", "-n", "1"], check=True, timeout=300, capture_output=True, text=True)
    run(tmp_path / "t", out1)
    assert out1.read_bytes() == (d / "ref_m3_csv.csv").read_bytes()
    # the Python binding of the same pair of entry points
    text = np.fromfile(tmp_path / "t.dna", dtype=np.uint8)
    g = api.GenomeIndex.build(text, device=0)
    g.save_sa(text, tmp_path / "u.sa")
    assert (tmp_path / "u.sa").read_bytes() == (tmp_path / "t.sa").read_bytes()
    g2 = api.GenomeIndex.open_sa(text, tmp_path / "u.sa", device=0)
    assert np.array_equal(g2.suffix_array(0), g.suffix_array(0)) and np.array_equal(g2.suffix_array(1), g.suffix_array(1))
    g.close()
    g2.close()
    other = text.copy()
    other[100:140] = other[100:140][::-1]
    with pytest.raises(api.GsError) as e:
        api.GenomeIndex.open_sa(other, tmp_path / "u.sa", device=0)
    assert e.value.status == 6
    # a stale .sa next to the text: ignored, index rebuilt, same output
    shutil.copy(tmp_path / "t.gs", tmp_path / "v.gs")
    other.tofile(tmp_path / "v.dna")
    shutil.copy(tmp_path / "t.sa", tmp_path / "v.sa")
    run(tmp_path / "v", out2)
    (tmp_path / "v.sa").unlink()
    run(tmp_path / "v", out3)
    assert out2.read_bytes() == out3.read_bytes()


def test_damaged_suffix_array_files_are_refused(tmp_path):
    """a PREFIX.sa of the right length whose payload is damaged never reaches the table builders:
    the payload checksums catch it (GS_ERR_FORMAT); a file in the older format without checksums,
    and a caller-supplied array, are checked for being permutations of their rows"""
    d = ol.ROOT / "tests" / "golden" / "toy"
    subprocess.run([str(CLI), "index", "--index", str(tmp_path / "t"), str(d / "toy.fa")], check=True, timeout=300)
    text = np.fromfile(tmp_path / "t.dna", dtype=np.uint8)
    g = api.GenomeIndex.build(text, device=0)
    g.save_sa(text, tmp_path / "a.sa")
    sa0, sa1 = g.suffix_array(0).copy(), g.suffix_array(1).copy()
    g.close()
    raw = bytearray((tmp_path / "a.sa").read_bytes())
    assert raw[:8] == b"GSAMDSA2"
    n = text.shape[0] + 1
    # (1) two rows swapped (still a permutation): only the checksum can tell
    bad = bytearray(raw)
    bad[64 + 4 * 10:64 + 4 * 11], bad[64 + 4 * 11:64 + 4 * 12] = raw[64 + 4 * 11:64 + 4 * 12], raw[64 + 4 * 10:64 + 4 * 11]
    (tmp_path / "b.sa").write_bytes(bad)
    with pytest.raises(api.GsError) as e:
        api.GenomeIndex.open_sa(text, tmp_path / "b.sa", device=0)
    assert e.value.status == 6
    # (2) the older header (no checksums) with an out-of-range value and with a repeated value
    for what in ("range", "twice"):
        old = bytearray(raw)
        old[:8] = b"GSAMDSA1"
        old[24:40] = bytes(16)
        at = 64 + 4 * (n + 7)  # a row of the reverse strand's array
        old[at:at + 4] = (0xFFFFFFF0).to_bytes(4, "little") if what == "range" else raw[at + 4:at + 8]
        (tmp_path / "c.sa").write_bytes(old)
        with pytest.raises(api.GsError) as e:
            api.GenomeIndex.open_sa(text, tmp_path / "c.sa", device=0)
        assert e.value.status == 6, what
    # the older format, undamaged, still opens
    old = bytearray(raw)
    old[:8] = b"GSAMDSA1"
    old[24:40] = bytes(16)
    (tmp_path / "d.sa").write_bytes(old)
    g3 = api.GenomeIndex.open_sa(text, tmp_path / "d.sa", device=0)
    assert np.array_equal(g3.suffix_array(1), sa1)
    g3.close()
    # (3) caller-supplied arrays (gs_index_build_with_sa): bad argument
    sa_bad = sa0.copy()
    sa_bad[5] = sa_bad[6]
    with pytest.raises(api.GsError) as e:
        api.GenomeIndex.build(text, device=0, sa_fwd=sa_bad, sa_rev=sa1)
    assert e.value.status == 1


def test_a_failing_batch_ends_the_run(tmp_path):
    """an error in any batch makes `enumerate` return 1 after the batches already handed out - it must
    not wait for batches no device thread will ever take (more batches than 2 x devices in flight)"""
    d = ol.ROOT / "tests" / "golden" / "toy"
    subprocess.run([str(CLI), "index", "--index", str(tmp_path / "t"), str(d / "toy.fa")], check=True, timeout=300)
    for extra in ([], ["--gpus", "1", "-n", "2"]):
        r = subprocess.run([str(CLI), "enumerate", str(tmp_path / "t"), "-f", str(d / "kmers.csv"), "-o",
                            str(tmp_path / "o.csv"), "-m", "8", "--batch-size", "1"] + extra,
                           timeout=120, capture_output=True, text=True)
        assert r.returncode == 1, r.stderr
        assert "error:" in r.stderr
        # and leaves no file behind that looks like a database with a hole in it
        assert not (tmp_path / "o.csv").exists()


def test_cli_fans_batches_out_over_several_workers(tmp_path):
    """`enumerate --gpus N`: one index and one host thread per device pull batches from a shared queue and
    the writer puts them back in input order.  With GS_CLI_SAME_DEVICE=1 the N workers all sit on device 0,
    so the queue / in-flight bound / ordered writer run with 3 workers on a one-GPU box: the file equals
    the reference's byte for byte, whatever the batch size"""
    d = ol.ROOT / "tests" / "golden" / "toy"
    subprocess.run([str(CLI), "index", "--index", str(tmp_path / "t"), str(d / "toy.fa")], check=True, timeout=300)
    env = dict(os.environ, GS_CLI_SAME_DEVICE="1")
    for bs in ("1", "5", "1000"):
        out = tmp_path / f"o{bs}.csv"
        subprocess.run([str(CLI), "enumerate", str(tmp_path / "t"), "-f", str(d / "kmers.csv"), "-o", str(out), "-m", "3",
                        "--gpus", "3", "--batch-size", bs, "-n", "2"], check=True, timeout=300, env=env,
                       capture_output=True, text=True)
        assert out.read_bytes() == (d / "ref_m3_csv.csv").read_bytes(), bs


def test_format_bam_decodes_to_the_sam_file(tmp_path):
    """`enumerate --format bam`: BGZF blocks of BAM records written by the formatting threads; decoded with
    the tests' own reader the file is the `--format sam` file of the same run (which equals the
    reference's), whatever the batch size and the number of formatting threads"""
    import bam_reader
    d = ol.ROOT / "tests" / "golden" / "toy"
    subprocess.run([str(CLI), "index", "--index", str(tmp_path / "t"), str(d / "toy.fa")], check=True, timeout=300)
    for extra, ref in ((["-m", "3", "-a", "NAG"], "ref_m3_sam_nag.sam"), (["-m", "2", "--mode", "succinct"], "ref_m2_sam_succinct.sam")):
        want = (d / ref).read_text()
        want = "".join("\t".join(f if (i != 2 or f) else "*" for i, f in enumerate(l.split("\t"))) if not l.startswith("@") else l
                       for l in want.splitlines(keepends=True))
        for bs, nt in (("1000", "1"), ("7", "3")):
            out = tmp_path / "o.bam"
            subprocess.run([str(CLI), "enumerate", str(tmp_path / "t"), "-f", str(d / "kmers.csv"), "-o", str(out), "--format", "bam",
                            "--batch-size", bs, "-n", nt] + extra, check=True, timeout=300, capture_output=True)
            assert bam_reader.to_sam(out) == want, (ref, bs, nt)
