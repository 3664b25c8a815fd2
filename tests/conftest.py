import os
import sys
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore", category=FutureWarning)
warnings.filterwarnings("ignore", category=UserWarning)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionstart(session):
    # the CPU oracle (ATen/MKL) is much slower with every SMT thread of a large host than with a subset
    import torch
    torch.set_num_threads(min(32, os.cpu_count() or 1))


@pytest.fixture(scope="session", autouse=True)
def plane_data_flow_at_every_size():
    """"pl_min_tokens" is 0 by default since r06 (one data flow at every call size; the thin-tile family of the plane GEMM).  The
    fixture keeps the process on that default and lets OMNITOK_TEST_PL_MIN_TOKENS=12288 run the whole suite with small calls on
    the fp32-activation flow instead (the A/B arm the option still selects)."""
    import torch
    if not torch.cuda.is_available():
        yield
        return
    from omnitokenizer_amd import _lib
    _lib.set_option("pl_min_tokens", int(os.environ.get("OMNITOK_TEST_PL_MIN_TOKENS", "0")))
    yield
    _lib.set_option("pl_min_tokens", 0)
