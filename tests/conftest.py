import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than ~30 s")


@pytest.fixture(scope="session")
def tiny_sd():
    from spann3r_amd.config import TINY
    from spann3r_amd.weights import synth_state_dict
    return synth_state_dict(0, TINY)


@pytest.fixture(scope="session")
def full_sd():
    from spann3r_amd.config import FULL
    from spann3r_amd.weights import synth_state_dict
    return synth_state_dict(0, FULL)


def load_golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name))


def rel_err(a, b):
    """max |a-b| / max|b|  (the '1e-3 rel' of BASELINE.json is read as max-norm relative error)."""
    import torch
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(autouse=True)
def _release_gpu_state_between_tests(request):
    """A GPU test builds models whose runners hold hipGraphs and GBs of workspaces in reference cycles (runner <-> model): finalise
    them at a defined point -- after the test, outside any capture -- instead of whenever the cyclic collector happens to run
    inside a later test."""
    yield
    if request.node.get_closest_marker("gpu") is not None and os.environ.get("SP3_TEST_NO_GC") != "1":
        import gc
        gc.collect()
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
        except ImportError:
            pass
