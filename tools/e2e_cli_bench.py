#!/usr/bin/env python3
"""End-to-end timing of the C++ CLI (kmers CSV in -> CSV database out) on a synthetic genome:
what SURVEY.md section 8d calls rate (iii), incl. index build, kmers parsing and the text writer.
    python tools/e2e_cli_bench.py [workload=chr1] [n_guides=100000] [extra CLI arguments ...]"""
import subprocess
import sys
import time
from importlib import import_module
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
synth = import_module("guidescan-cli_amd.synth")


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "chr1"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    lengths = {"chr1": [synth.CHR1_LENGTH], "hg38": synth.GRCH38_LENGTHS, "saccer3": synth.SACCER3_LENGTHS}[workload]
    import os
    d = Path(os.environ.get("GS_E2E_DIR", "/tmp/gs_e2e"))
    d.mkdir(exist_ok=True)
    text, names, lengths = synth.make_genome(lengths, seed=1)
    text.tofile(d / "g.dna")
    (d / "g.gs").write_text("".join(f"{a}\n{b}\n" for a, b in zip(names, lengths)))
    seqs, pams, pos, strands = synth.sample_guides(text, n, seed=5)
    with open(d / "k.csv", "w") as f:
        f.write("id,sequence,pam,chromosome,position,sense\n")
        for i in range(n):
            f.write(f"g{i},{seqs[i].tobytes().decode()},NGG,chr1,{int(pos[i]) + 1},{chr(strands[i])}\n")
    cli = ROOT / "guidescan-cli_amd" / "bin" / "guidescan"
    t0 = time.time()
    r = subprocess.run([str(cli), "enumerate", str(d / "g"), "-f", str(d / "k.csv"), "-o", str(d / "o.csv"), "-m", "3"] + sys.argv[3:],
                       capture_output=True, text=True, timeout=3000)
    dt = time.time() - t0
    print(r.stdout.strip())
    print(r.stderr.strip())
    print(f"wall {dt:.2f} s for {n} guides ({workload}); output {(d / 'o.csv').stat().st_size / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
