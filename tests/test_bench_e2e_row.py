"""bench.py's end-to-end row (SURVEY 8d's rate iii: the built `guidescan enumerate`, src/guidescan.cxx:181-258, timed by its
own `Processed N kmers in S seconds`) with the command replaced by a stand-in: the row runs the command twice, reports the
faster run, lists both, and compares the first bytes of the output with the in-process formatter's.  Host logic only."""
import hashlib
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def test_the_row_reports_the_faster_of_two_runs_and_checks_the_first_bytes(monkeypatch, tmp_path):
    expect = b"id,sequence\n" + b"g0,ACGT\n" * 5
    seconds = iter([1.7, 0.35])          # a box's hiccup first, then what the path takes
    calls = []

    def fake_run(cmd, capture_output, text, timeout):
        out = Path(cmd[cmd.index("-o") + 1])
        assert not out.exists()          # the second run does not find the first run's file
        out.write_bytes(expect + b"tail of the file\n")
        s = next(seconds)
        calls.append(cmd)
        return SimpleNamespace(returncode=0, stderr="", stdout=(
            f"Built the index on device 0 in 8.9 s\nProcessed 3 kmers in {s} seconds.\n"
            f"Stages (overlapping): device {s / 2} s, text formatting 0.1 s, file writes 0.2 s\n"))

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setenv("GS_E2E_DIR", str(tmp_path))
    seqs = np.frombuffer(b"ACGTACGTACGTACGTACGT" * 3, dtype=np.uint8).reshape(3, 20)
    prep = {"n": 3, "seqs": seqs, "pos": np.array([5, 9, 100]), "strands": np.frombuffer(b"+-+", dtype=np.uint8),
            "in_hbm_ms": 17.0, "host_pointer_ms": 21.0, "expect_prefix": expect, "check_guides": 3,
            "expect_sha256": hashlib.sha256(expect).hexdigest()}
    text = np.frombuffer(b"ACGT" * 50, dtype=np.uint8)
    row = bench.e2e_cli_row(prep, text, ["chr1"], [200], "csv")
    assert len(calls) == 2 and calls[0][1] == "enumerate"
    assert row["iii_cli_seconds_each_run"] == [1.7, 0.35]
    assert row["iii_cli_seconds_after_index_load"] == 0.35 and abs(row["iii_guides_per_s"] - 3 / 0.35) < 1e-9
    assert row["cli_stage_seconds_overlapping"]["device"] == 0.175   # the stages of the run that is reported
    assert row["first_bytes_equal_in_process_formatter"] and row["sha256_first_bytes_cli"] == row["sha256_in_process"]
    assert row["output_bytes"] == len(expect) + len(b"tail of the file\n")
    assert not any(tmp_path.iterdir())   # the row's directory is removed
