"""Candidate-guide fixtures (tests/golden/kmers/*.json): input records and the rows the reference's
own scripts/generate_kmers.py gave for them (made by tools/make_kmers_goldens.py in the build
container), plus a brute-force property checker that shares no code with the product or with the
reference script.  TEST INFRASTRUCTURE."""
import json
from pathlib import Path

GOLDEN_DIR = Path(__file__).resolve().parent / "golden" / "kmers"
_RC = {"A": "T", "C": "G", "G": "C", "T": "A"}


def golden_cases():
    return [json.loads(p.read_text()) for p in sorted(GOLDEN_DIR.glob("*.json"))]


def expected_rows(case):
    return [(r[0], r[1], r[2]) for r in case["rows"]]


def _matches(window, pattern):
    return len(window) == len(pattern) and all((w in "ACGT") if p == "N" else (p == w)
                                               for w, p in zip(window, pattern))


def check_properties(record, pam, k, start, got):
    """got: list of (sequence, position 1-based, sense).  (1) every reported site is a site: its
    protospacer is k bases of ACGT next to a window matching the pattern ('N' = any base) on the
    stated strand; (2) every site of the record is reported exactly once; (3) order: + strand
    before - strand, within a strand grouped by the concrete PAM in the expansion order the first N
    -> A,C,T,G gives, within a group by position."""
    rec = record.upper()
    n, P = len(rec), len(pam)
    rc_pam = "".join(_RC.get(c, c) for c in reversed(pam))

    def site(pos0, sense):
        """(protospacer as the guide reads it, concrete PAM as it lies on the + strand) or None"""
        if sense == "+":
            if not start:
                lo, plo = pos0, pos0 + k          # protospacer, then PAM
            else:
                plo, lo = pos0, pos0 + P          # PAM, then protospacer
            pat = pam
        else:
            if not start:
                plo, lo = pos0, pos0 + P          # revcomp(PAM), then the protospacer's revcomp
            else:
                lo, plo = pos0, pos0 + k
            pat = rc_pam
        if lo < 0 or plo < 0 or lo + k > n or plo + P > n:
            return None
        proto, win = rec[lo:lo + k], rec[plo:plo + P]
        if not all(c in "ACGT" for c in proto) or not _matches(win, pat):
            return None
        if sense == "-":
            proto = "".join(_RC[c] for c in reversed(proto))
        return proto, win

    want = {}
    for sense in "+-":
        for pos0 in range(-(k + P), n + 1):
            s = site(pos0, sense)
            if s is not None:
                want[(pos0 + 1, sense)] = s
    seen = {}
    for seq, pos, sense in got:
        assert (pos, sense) in want, f"reported {(seq, pos, sense)} is not a site"
        assert want[(pos, sense)][0] == seq, (seq, pos, sense)
        assert (pos, sense) not in seen, f"{(pos, sense)} reported twice"
        seen[(pos, sense)] = True
    assert len(seen) == len(want), f"{len(want) - len(seen)} sites not reported"
    # order: bucket = rank of the concrete PAM (as on the + strand; for - strand sites the reverse
    # complement of that window) among the expansions, N -> A,C,T,G, first N varying slowest
    order = "ACTG"

    def bucket(pos, sense):
        win = want[(pos, sense)][1]
        if sense == "-":
            win = "".join(_RC[c] for c in reversed(win))
        b = 0
        for p, w in zip(pam, win):
            if p == "N":
                b = b * 4 + order.index(w)
        return (0 if sense == "+" else 1, b, pos)

    keys = [bucket(pos, sense) for _, pos, sense in got]
    assert keys == sorted(keys), "rows are not in (strand, PAM expansion, position) order"
