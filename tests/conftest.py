import gzip
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def read_fastq_gz(path):
    with gzip.open(path, "rb") as f:
        lines = f.read().split(b"\n")
    recs = []
    for i in range(0, len(lines) - 1, 4):
        recs.append((lines[i], lines[i + 1], lines[i + 3]))
    return recs


@pytest.fixture(scope="session")
def toyset():
    """Recovered toyset/rna input: list of (header, seq, qual) in seq_id order (= length-descending)."""
    return read_fastq_gz(os.path.join(GOLDEN, "toyset_rna.fastq.gz"))


@pytest.fixture(scope="session")
def toyset_clusters():
    from rattle_amd import hps
    return hps.decode(open(os.path.join(GOLDEN, "toyset_rna.clusters.out"), "rb").read(), fields=2)


@pytest.fixture(scope="session")
def oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc_mod
    orc_mod.build(ref=True)
    return orc_mod.Oracle()


@pytest.fixture(scope="session")
def ref_lib(oracle):
    import oracle as orc_mod
    if not orc_mod.have_ref():
        pytest.skip("oracle/_ref/libref.so not built (needs /root/reference)")
    return orc_mod.Ref()


@pytest.fixture(scope="session")
def gpu_ctx():
    from rattle_amd.api import Context
    ctx = Context(0)          # raises loudly without a device / library
    yield ctx
    ctx.close()
