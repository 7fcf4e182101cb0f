import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_vectors.npz"))


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def hrcore_lib():
    """The product library.  GPU tests fail loudly (not skip) when it is missing."""
    from heatray_amd import core
    return core.load_library()


def pytest_collection_finish(session):
    # Tests that hand torch device buffers to the library initialise torch's HIP context; do that before the first engine
    # exists, whatever subset of the GPU tests was selected (a late torch initialisation has failed with "No HIP GPUs are
    # available" on a box where the library had been using the card for a while).
    if any(item.get_closest_marker("gpu") for item in session.items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
