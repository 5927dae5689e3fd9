import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")
    config.addinivalue_line("markers", "slow: minutes of GPU time; gated by an environment switch named in its skip reason")


def pytest_collection_modifyitems(config, items):
    """Kernel parity first, statistics last: the driver runs `pytest -x`, so a statistical (training-outcome) test must
    never sit in front of a kernel parity file.  Every item of the R2 acceptance file goes to the end of the run, whatever
    the file is called and however the files sort; the multi-process dry run just before it."""
    def rank(item):
        name = os.path.basename(str(item.fspath))
        if "r2_acceptance" in name:
            return 2
        if name == "test_dist_gpu.py":
            return 1
        return 0
    items.sort(key=rank)        # (stable: the collection order inside a rank stays)


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")
