import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def rel_l2(a, b):
    """||a-b|| / ||b|| in fp64."""
    import numpy as np

    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))


def pytest_sessionfinish(session, exitstatus):
    """Free device objects while the HIP runtime is still up (captured graphs, cached workspaces): objects that live until
    interpreter teardown are destroyed after the runtime."""
    import gc
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            from adaface_dev_amd import graphs
            graphs._release_all()
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    except Exception:                    # noqa: BLE001  (teardown must never turn a green run red)
        pass
