import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle

    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def gpu():
    """The product library + a visible GPU, or a loud failure (never a silent skip on a GPU box)."""
    import torch

    from exon_duckdb_amd import load_library

    lib = load_library()
    assert torch.cuda.is_available(), "gpu-marked test needs a GPU"
    assert lib.exg_device_count() >= 1
    return lib
