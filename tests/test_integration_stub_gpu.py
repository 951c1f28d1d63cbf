"""The reference-side binding of INTEGRATION.md, compiled and run: oracle/_ref/gs_ref_enumerate_gpu is
the reference's enumerate command (its kmers reader, genome structure, printers - compiled from
/root/reference in place) with integration/process_gpu.hpp in the place of process_kmers_to_stream,
i.e. the search behind libgsamd.so's C-ABI and the index opened from the reference's OWN index
files (gs_index_open_sdsl).  Its output files must equal those of oracle/_ref/gs_ref_enumerate, the
unmodified reference pipeline, on seeded genomes over the bulge-free option sets.  GPU only; the
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


def run_stub(prefix, kmers_csv, out, m=3, fmt="csv", complete=True, alt=(), maxo=-1, thr=-1, start=False):
    cmd = [str(STUB), str(prefix), str(kmers_csv), str(out), fmt, "complete" if complete else "succinct",
           str(m), "0", "0", str(thr), str(maxo), "1" if start else "0", *alt]
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
    for cfg in pipe.OPTION_SETS:
        if cfg.get("rna") or cfg.get("dna"):
            continue
        want = pipe.run_shim(tmp_path / "r.idx", kcsv, tmp_path / "want", **cfg)
        got = run_stub(tmp_path / "r.idx", kcsv, tmp_path / "got", **cfg)
        assert got == want, (seed, cfg)
        n += 1
    assert n == 12
