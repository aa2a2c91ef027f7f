import os
import sys
import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a machine without a GPU skips the gpu-marked tests.  When they are asked for explicitly
    (-m gpu, what the GPU box runs, or YOHO_FORCE_GPU_TESTS=1) nothing is skipped: a missing GPU or library must fail
    loudly there, never pass silently."""
    expr = (config.getoption("-m") or "").strip()
    asked = ("gpu" in expr and "not gpu" not in expr) or os.environ.get("YOHO_FORCE_GPU_TESTS") == "1"
    if asked:
        return
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a MI355X (torch.cuda.is_available() is False); run with -m gpu on the GPU box")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def tables():
    from yoho_amd.tables import default_tables
    return default_tables()


@pytest.fixture(scope="session")
def gold():
    def load(name):
        return np.load(os.path.join(GOLD, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def sd1():
    from yoho_amd import weights as W
    return W.synth_state_dict(W.PARTI_SPEC, 7)


@pytest.fixture(scope="session")
def sd2():
    from yoho_amd import weights as W
    return W.synth_state_dict(W.PARTII_SPEC, 8)


@pytest.fixture(scope="session")
def hip():
    """The loaded C-ABI library wrapper; GPU tests fail (not skip) if it cannot be loaded."""
    from yoho_amd import hip as _hip
    return _hip
