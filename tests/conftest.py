import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def toy():
    """toy genome fixture: text, chromosome names/lengths, kmers (tests/golden/toy)."""
    from importlib import import_module
    seqio = import_module("guidescan-cli_amd.seqio")
    d = ROOT / "tests" / "golden" / "toy"
    text, names, lengths = seqio.parse_fasta(d / "toy.fa")
    kmers = seqio.read_kmers(d / "kmers.csv")
    return dict(dir=d, text=text, names=names, lengths=lengths, kmers=kmers)
