import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present() -> bool:
    return os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK)


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _abort_trace():
    """On the GPU box: if the HIP / ROCr runtime ever abort()s the test process (its way of reporting a GPU fault), leave the native
    call stack in gpurun_out/abort_native_stack.log -- pytest holds descriptor 2 while tests run, so stderr alone would lose it."""
    log = None
    if _gpu_present():
        try:
            from dspeed_amd import _lib

            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            log = open(os.path.join(ROOT, "gpurun_out", "abort_native_stack.log"), "a")
            _lib.lib().dsp_install_abort_trace(log.fileno())
        except Exception:
            pass
    yield
    if log is not None:
        try:
            from dspeed_amd import _lib

            _lib.lib().dsp_uninstall_abort_trace()
            log.close()
        except Exception:
            pass
