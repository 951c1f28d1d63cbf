"""The reference-side binding of INTEGRATION.md, compiled and run: oracle/_ref/gs_ref_enumerate_gpu is
the reference's enumerate command (its kmers reader, genome structure, printers - compiled from
/root/reference in place) with integration/process_gpu.hpp in the place of process_kmers_to_stream,
i.e. the search behind libgsamd.so's C-ABI and the index opened from the reference's OWN index
files (gs_index_open_sdsl).  Its output files must equal those of oracle/_ref/gs_ref_enumerate, the
unmodified reference pipeline, on seeded genomes over every option set (bulges, --threshold, --start, alt PAMs)
and with guides and PAMs that hold symbols outside A,C,G,T.  GPU only; the
prebuilt binaries travel with the snapshot, nothing reads /root/reference here."""
import subprocess
from pathlib import Path

import pytest

import oracle_lib as ol
import test_oracle_vs_ref_pipeline as pipe

STUB = ol.ORACLE_DIR / "_ref" / "gs_ref_enumerate_gpu"
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(pipe.ref is None or not pipe.SHIM.exists() or not STUB.exists(),
                                 reason="oracle/_ref not built")]


def run_stub(prefix, kmers_csv, out, m=3, fmt="csv", complete=True, alt=(), maxo=-1, thr=-1, start=False, rna=0, dna=0):
    cmd = [str(STUB), str(prefix), str(kmers_csv), str(out), fmt, "complete" if complete else "succinct",
           str(m), str(rna), str(dna), str(thr), str(maxo), "1" if start else "0", *alt]
    subprocess.run(cmd, check=True, timeout=300)
    return Path(out).read_bytes()


@pytest.mark.parametrize("seed", [401, 402, 403, 404])
def test_reference_printers_fed_by_the_c_abi_equal_the_reference(seed, tmp_path):
    text, names, lengths, rows = pipe.random_case(seed)
    kcsv = tmp_path / "kmers.csv"
    pipe.synth.write_kmers_csv(kcsv, [r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows],
                               [names[0]] * len(rows), [1] * len(rows), [r[3] for r in rows])
    oidx = ol.OracleIndex(text)
    try:
        pipe.write_reference_index(oidx, text.shape[0] + 1, tmp_path / "r.idx", names, lengths)
    finally:
        oidx.close()
    n = 0
    for cfg in pipe.OPTION_SETS:   # the six bulge sets go through gs_enumerate_bulges inside the stub
        want = pipe.run_shim(tmp_path / "r.idx", kcsv, tmp_path / "want", **cfg)
        got = run_stub(tmp_path / "r.idx", kcsv, tmp_path / "got", **cfg)
        assert got == want, (seed, cfg)
        n += 1
    assert n == 18


ODD_SETS = [dict(m=2), dict(m=3, fmt="sam"), dict(m=2, alt=("NRG",)), dict(m=2, alt=("NAG", "NNGRRT")), dict(m=1, thr=1),
            dict(m=2, start=True, alt=("NAG",)), dict(m=1, rna=1, dna=1)]


@pytest.mark.parametrize("seed", [411, 412])
def test_guides_the_fast_path_flags_take_their_hits_from_the_general_path(seed, tmp_path):
    """a batch that mixes plain guides with guides holding an N or an IUPAC letter (matched literally against the
    genome, charged a mismatch otherwise: index.hpp:218-247), one whose own PAM holds a literal, alt PAMs with an
    IUPAC letter and of another length: the stub reads gs_result_view.guide_flags, sends the flagged guides through
    gs_enumerate_general, and the files still equal the unmodified reference's"""
    text, names, lengths, rows = pipe.random_case(seed)
    fam = rows[-4][1]
    rows = rows + [("withN", fam[:7] + "N" + fam[8:], "NGG", "+"), ("withR", fam[:12] + "R" + fam[13:], "NGG", "-"),
                   ("allN", "N" * 20, "NGG", "+"), ("ownpam", fam, "NRG", "+"), ("lower", fam[:5] + "a" + fam[6:], "NGG", "+")]
    kcsv = tmp_path / "kmers.csv"
    pipe.synth.write_kmers_csv(kcsv, [r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows],
                               [names[0]] * len(rows), [1] * len(rows), [r[3] for r in rows])
    oidx = ol.OracleIndex(text)
    try:
        pipe.write_reference_index(oidx, text.shape[0] + 1, tmp_path / "r.idx", names, lengths)
    finally:
        oidx.close()
    for cfg in ODD_SETS:
        want = pipe.run_shim(tmp_path / "r.idx", kcsv, tmp_path / "want", **cfg)
        got = run_stub(tmp_path / "r.idx", kcsv, tmp_path / "got", **cfg)
        assert got == want, (seed, cfg)
