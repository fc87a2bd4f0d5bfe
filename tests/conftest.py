import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def built():
    """Make sure librnamc.so and the oracle exist (cross-compiles without a GPU)."""
    import __graft_entry__ as g
    from rna_algos_amd import _lib
    if not os.path.exists(_lib.LIB_PATH) or not os.path.exists(
            os.path.join(ROOT, "oracle", "librnamc_oracle.so")):
        g.build()
    return True


@pytest.fixture(scope="session")
def params(built):
    from rna_algos_amd.utils import FoldScoreSets
    return FoldScoreSets.synthetic(1)


@pytest.fixture(scope="session")
def trnas(built):
    from rna_algos_amd.utils import read_fasta
    return read_fasta(os.path.join(ROOT, "tests", "golden", "sampled_trnas.fa"))
