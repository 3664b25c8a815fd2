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
    """The engine routes calls below "pl_min_tokens" (12288 tokens) to the fp32-activation data flow (include/omnitok.h);
    the golden fixtures are all smaller than that, so the GPU session pins the threshold to 0 -- the tests then exercise
    the plane data flow the full-size workloads run, next to the explicit gemm_pl 0 parametrisations.  The rule itself is
    tested in tests/test_gpu_e2e.py::test_small_calls_take_the_fp32_activation_flow, which sets and restores it."""
    import torch
    if not torch.cuda.is_available():
        yield
        return
    from omnitokenizer_amd import _lib
    # OMNITOK_TEST_PL_MIN_TOKENS=12288 runs the whole suite on the flow small calls take by default instead
    # (profiles/r04_gpu_tests_small_call_flow.txt)
    _lib.set_option("pl_min_tokens", int(os.environ.get("OMNITOK_TEST_PL_MIN_TOKENS", "0")))
    yield
    _lib.set_option("pl_min_tokens", 12288)
