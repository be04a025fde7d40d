import os
import sys

import pytest

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")  # HIP-runtime graph-replay workaround (keypointfusion_amd/graphs.py): before the first HIP call

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


_SD_CACHE = {}


def synthetic_sd(net):
    """Session cache of the synthetic state dict as torch CPU tensors."""
    import torch
    from keypointfusion_amd.weights import synthetic_state_dict
    if net not in _SD_CACHE:
        _SD_CACHE[net] = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(net, 0).items()}
    return _SD_CACHE[net]
