import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:          # helpers beside the tests (_fuzz_cases.py)
    sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle's small recurrent / convolution steps run into OpenMP fork-join overhead on a many-core host: on the 128-thread GPU box
    # one GRU-decoder oracle forward of 16 blocks took 196 s with torch's default thread count against 1.6 s with 8 threads here
    # (tools/lab/probes/lstm_slow.py).  The oracle's results do not depend on the thread count beyond fp32 summation order.
    import torch
    if torch.get_num_threads() > 16:
        torch.set_num_threads(16)


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu but no ROCm device is visible")
    return torch.device("cuda", 0)
