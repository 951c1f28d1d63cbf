"""`--format bam` (host/bam_writer.hpp): the reference writes SAM text and leaves BAM to `samtools view -b`
(manual/manual.tex:581-582); the product's BAM decodes to exactly the SAM file.  Here, without a GPU: the
reference's own SAM files (tests/golden/toy) through `guidescan sam2bam` and back through an independent
reader (tests/bam_reader.py: BGZF framing, reg2bin, typed tags)."""
import subprocess

import pytest

import bam_reader
import oracle_lib as ol

CLI = ol.ROOT / "guidescan-cli_amd" / "bin" / "guidescan"
GOLD = ol.ROOT / "tests" / "golden" / "toy"


@pytest.mark.parametrize("sam", sorted(p.name for p in GOLD.glob("*.sam")))
def test_reference_sam_files_survive_the_bam_round_trip(sam, tmp_path):
    out = tmp_path / "o.bam"
    subprocess.run([str(CLI), "sam2bam", str(GOLD / sam), str(out)], check=True, timeout=60)
    sizes, eof = bam_reader.bgzf_blocks(out.read_bytes())
    assert eof and len(sizes) >= 2
    # a hit dropped at a chromosome boundary has an EMPTY reference name in the reference's SAM
    # (printer.hpp:330): BAM stores an index, the line comes back with '*'
    want = "".join("\t".join(f if (i != 2 or f) else "*" for i, f in enumerate(l.split("\t"))) if not l.startswith("@") else l
                   for l in (GOLD / sam).read_text().splitlines(keepends=True))
    assert bam_reader.to_sam(out) == want


def test_large_blocks_and_odd_fields(tmp_path):
    """more than one 64 KiB block, an odd-length sequence with N, integers of every width, a missing RNAME"""
    head = "@HD\tVN:1.0\tSO:unknown\n@SQ\tSN:c1\tLN:500000000\n@SQ\tSN:c2\tLN:1000\n"
    lines = []
    for i in range(3000):
        lines.append(f"q{i}\t{16 * (i % 2)}\tc{1 + i % 2}\t{1 + (i * 7919) % 900}\t100\t23M\t*\t0\t0\tACGTNACGTACGTACGTACGTAC\t*"
                     f"\tk0:i:{i}\tk1:i:{-i}\tk2:i:{i * 70000}\tof:H:{'ab' * (i % 40)}\tsp:f:{1 / (1 + i):.6f}")
    lines.append("u\t4\t*\t0\t0\t*\t*\t0\t0\tACG\t*\tk0:i:-70000")
    lines.append("far\t0\tc1\t300000000\t100\t5M2D5M\t=\t300000100\t110\tACGTACGTAC\tIIIIIIIIII")
    (tmp_path / "i.sam").write_text(head + "\n".join(lines) + "\n")
    subprocess.run([str(CLI), "sam2bam", str(tmp_path / "i.sam"), str(tmp_path / "o.bam")], check=True, timeout=60)
    sizes, eof = bam_reader.bgzf_blocks((tmp_path / "o.bam").read_bytes())
    assert eof and len(sizes) >= 4
    assert bam_reader.to_sam(tmp_path / "o.bam") == (tmp_path / "i.sam").read_text()


def test_values_bam_cannot_hold_are_errors(tmp_path):
    """l_read_name is one byte (name + NUL <= 255), MAPQ is one byte, integer tags are at most 32 bits wide: a SAM
    line beyond that is refused (exit code 1, 'malformed SAM line'), not written with a truncated field"""
    head = "@HD\tVN:1.0\tSO:unknown\n@SQ\tSN:c1\tLN:5000\n"
    ok = "q\t0\tc1\t10\t100\t23M\t*\t0\t0\tACGTNACGTACGTACGTACGTAC\t*\tk0:i:1\n"
    bad = {
        "long-name": "n" * 255 + "\t0\tc1\t10\t100\t23M\t*\t0\t0\tACGTNACGTACGTACGTACGTAC\t*\tk0:i:1\n",
        "mapq": "q\t0\tc1\t10\t256\t23M\t*\t0\t0\tACGTNACGTACGTACGTACGTAC\t*\tk0:i:1\n",
        "tag-high": "q\t0\tc1\t10\t100\t23M\t*\t0\t0\tACGTNACGTACGTACGTACGTAC\t*\tk0:i:4294967296\n",
        "tag-low": "q\t0\tc1\t10\t100\t23M\t*\t0\t0\tACGTNACGTACGTACGTACGTAC\t*\tk0:i:-2147483649\n",
    }
    (tmp_path / "ok.sam").write_text(head + "n" * 254 + ok[1:] + ok)
    subprocess.run([str(CLI), "sam2bam", str(tmp_path / "ok.sam"), str(tmp_path / "ok.bam")], check=True, timeout=60)
    assert bam_reader.to_sam(tmp_path / "ok.bam") == (tmp_path / "ok.sam").read_text()
    for name, line in bad.items():
        (tmp_path / "b.sam").write_text(head + ok + line)
        r = subprocess.run([str(CLI), "sam2bam", str(tmp_path / "b.sam"), str(tmp_path / "b.bam")], timeout=60,
                           capture_output=True, text=True)
        assert r.returncode != 0, name
        assert "malformed" in r.stderr.lower() or "error" in r.stderr.lower(), (name, r.stderr)
