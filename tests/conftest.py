import os
import sys

import numpy as np
import pytest

# the library reads its CRPSPMM_* knobs once (csrc/knobs.cpp); the tests change some between calls of one process
os.environ.setdefault("CRPSPMM_KNOBS_LIVE", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
GOLDEN_NAMES = ("g_symm", "g_gen", "g_pat", "g_int")

# fp64 parity bar of BASELINE.json north_star: relative Frobenius error <= 1e-12
FP64_TOL = 1e-12


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def crp():
    import crp_spmm_amd
    crp_spmm_amd.load()
    return crp_spmm_amd


def load_golden(name, kind):
    return np.load(os.path.join(GOLDEN, "%s.%s.npz" % (name, kind)), allow_pickle=False)


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible")
    torch.cuda.set_device(0)
    return torch.device("cuda", 0)
