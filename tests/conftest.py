import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")
    # CPU-side references (oracle, reference Trainer) : eager PyTorch does not scale to the 256 hardware threads of the GPU boxes - the same
    # work runs several times SLOWER there than on 32 threads (bench.py cpu_baseline picks 32 of 256 by measurement)
    import torch

    torch.set_num_threads(min(32, os.cpu_count() or 1))


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
