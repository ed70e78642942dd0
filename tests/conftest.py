import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _range_flag_hygiene(request):
    """The f16x3 range flag is one word per device, read by the next ``forward_loop``: a test that
    leaves it raised would fail whichever test runs after it.  Every GPU test starts with a clear
    flag and is failed HERE, by name, if it leaves the flag set."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch

    if not torch.cuda.is_available():
        yield
        return
    from multimodalfilter_amd import _abi, engine

    dev = torch.device("cuda:0")
    try:
        engine.check_range(dev)
    except _abi.MmfError:
        pass  # raised by something outside the tests (an import-time warm-up); now clear
    yield
    try:
        engine.check_range(dev)
    except _abi.MmfError:
        pytest.fail(f"{request.node.name} left the f16x3 range flag raised (an activation saturated the operand split)")
