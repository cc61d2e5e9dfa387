"""pytest configuration: registers the ``gpu`` marker and puts the product package + oracle on sys.path."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "spateo-release_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def golden():
    """Arrays produced by the real reference twins (tests/golden/make_golden.py, run in the build container)."""
    path = os.path.join(ROOT, "tests", "golden", "ref_twins.npz")
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_align():
    """Outputs of the real alignment-side reference functions (tests/golden/make_golden_align.py)."""
    path = os.path.join(ROOT, "tests", "golden", "ref_align.npz")
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_em():
    """Inputs / outputs of the real Morpho_pairwise._construct_kernel + _update_nonrigid (tests/golden/make_golden_em.py):
    the in-tree statement of the SparseVFC M-step arithmetic."""
    path = os.path.join(ROOT, "tests", "golden", "ref_em.npz")
    with np.load(path) as z:
        return {k: z[k] for k in z.files}
