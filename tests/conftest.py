import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # After a recovered time-out the engine keeps NEW trajectories off the cluster / slab kernels for a cool-down
    # period; the fault-injection tests must not silently move the tests that follow them onto the tile kernel
    # (tests/test_gpu_slab.py::test_time_out_cool_down sets its own value).
    os.environ.setdefault("CCVM_AMD_EXCHANGE_COOLDOWN", "0")


@pytest.fixture(scope="session")
def hip_lib():
    """The built C-ABI library (compiles it if stale; no GPU needed to load)."""
    import __graft_entry__ as entry

    entry.build_hip()
    from ccvm_amd import _lib

    return _lib.load()
