import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """On a GPU run (`-m gpu`), bring torch's HIP runtime up BEFORE the first test touches libgsamd.so: some
    tests move device buffers through torch, and torch initialised after the library had run kernels in the
    same process reported "No HIP GPUs are available" on the GPU box (whatever test came first decided)."""
    if not any(it.get_closest_marker("gpu") for it in items):
        return
    if "not gpu" in (config.getoption("-m") or ""):
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception as e:  # the tests that need torch will say so themselves
        print(f"[conftest] torch was not initialised ahead of the GPU tests: {e!r}", file=sys.stderr)


@pytest.fixture(scope="session")
def toy():
    """toy genome fixture: text, chromosome names/lengths, kmers (tests/golden/toy)."""
    from importlib import import_module
    seqio = import_module("guidescan-cli_amd.seqio")
    d = ROOT / "tests" / "golden" / "toy"
    text, names, lengths = seqio.parse_fasta(d / "toy.fa")
    kmers = seqio.read_kmers(d / "kmers.csv")
    return dict(dir=d, text=text, names=names, lengths=lengths, kmers=kmers)
