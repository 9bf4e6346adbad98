import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle's torch ops: a GPU box hands a test run 16 CPUs of a 256-thread host, and torch's default pool of one
    # thread per LOGICAL CPU makes the oracle's small GEMMs crawl there (the trajectory tests: 262 s -> 26 s)
    import torch
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))


@pytest.fixture(scope="session")
def oracle():
    import pnr_oracle
    pnr_oracle.build_c_oracle()
    return pnr_oracle


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (the HIP path has no CPU fallback)")
    return torch.device("cuda:0")
