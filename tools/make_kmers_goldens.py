#!/usr/bin/env python3
"""Writes tests/golden/kmers/*.json: input records and the rows the REFERENCE's own
scripts/generate_kmers.py produces for them.  Build-container only (needs /root/reference).

The script cannot be imported as a module here (its first line imports Bio, which this image
lacks, and Bio is only used by its __main__ FASTA loop).  Its functions are therefore taken from
the file where it lies: the source is parsed with `ast`, the top-level assignments and function
definitions (NUCS, NUC_MAP, revcom, generate_pam_set, find_kmers, find_all_kmers) are compiled and
executed as they stand, and find_all_kmers() is called on each case's record.  Nothing of the
script is kept in this repository: the fixtures hold inputs and expected rows only.
"""
import ast
import json
import sys
from pathlib import Path

import numpy as np

REF = Path("/root/reference/scripts/generate_kmers.py")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "kmers"


def reference_functions():
    tree = ast.parse(REF.read_text(), filename=str(REF))
    keep = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.Assign))]
    ns = {}
    exec(compile(ast.Module(body=keep, type_ignores=[]), str(REF), "exec"), ns)
    return ns


def cases():
    rng = np.random.default_rng(5)
    alpha, p = list("ACGTNacgt"), [.22, .22, .22, .22, .04, .02, .02, .02, .02]
    seq = "GG" + "".join(rng.choice(alpha, 6000, p=p)) + "CC"     # PAMs at the very ends
    yield dict(name="ngg_k20", pam="NGG", k=20, start=False, record=seq)
    yield dict(name="nag_k20", pam="NAG", k=20, start=False, record=seq)
    yield dict(name="ngg_k20_start", pam="NGG", k=20, start=True, record=seq)
    yield dict(name="tttn_k23_start", pam="TTTN", k=23, start=True, record=seq)
    yield dict(name="nngaat_k21", pam="NNGAAT", k=21, start=False, record=seq)
    yield dict(name="nnn_k8", pam="NNN", k=8, start=False, record=seq[:400])
    yield dict(name="no_n_pam", pam="TTG", k=12, start=False, record=seq[:1500])
    # record ends: a PAM closer than k to either end, a record shorter than k + P, an empty record
    yield dict(name="short_records", pam="NGG", k=20, start=False, record="ACGTACGTAGGTTCCAACGT")
    yield dict(name="pam_at_both_ends", pam="NGG", k=5, start=False, record="AGGACGTTCGGACGTACCTACGTACCA")
    yield dict(name="empty_record", pam="NGG", k=20, start=False, record="")
    # N runs and lower-case stretches around sites
    s2 = "".join(rng.choice(list("ACGT"), 900))
    s2 = s2[:200] + "N" * 37 + s2[237:500].lower() + s2[500:]
    yield dict(name="n_run_lowercase", pam="NGG", k=20, start=False, record=s2)
    yield dict(name="n_run_lowercase_start", pam="NGG", k=20, start=True, record=s2)


def main():
    if not REF.exists():
        sys.exit("needs the reference tree (build container only)")
    ns = reference_functions()
    OUT.mkdir(parents=True, exist_ok=True)
    for c in cases():
        rows = [[r["sequence"], r["position"], r["sense"], r["pam"]]
                for r in ns["find_all_kmers"](c["pam"], c["k"], c["record"], end=not c["start"])]
        c["rows"] = rows
        c["pam_set"] = ns["generate_pam_set"](c["pam"])
        (OUT / (c["name"] + ".json")).write_text(json.dumps(c))
        print(c["name"], len(rows), "rows")


if __name__ == "__main__":
    main()
