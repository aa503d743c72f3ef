"""The drop-in boundary (include/dspeed_hip.h -> dspeed_amd/libdspeed_hip.so) on a machine without a GPU: the library builds
(hipcc cross-compiles for gfx950), loads, and exports every function the header declares; the ctypes layer knows each of them;
nothing here makes a compute call."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dspeed_hip.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # comments mention function names too
    return sorted(set(re.findall(r"^\s*(?:const\s+char\s*\*|int)\s+(dsp_[a-z0-9_]+)\s*\(", text, flags=re.M)))


@pytest.fixture(scope="module")
def library():
    from dspeed_amd import _lib, build

    build.build(force=False, verbose=False)
    return ctypes.CDLL(_lib.LIB_PATH)


def test_header_declares_the_expected_surface():
    names = _declared()
    assert len(names) >= 60
    for must in ("dsp_chain_create", "dsp_chain_execute", "dsp_chain_check", "dsp_chain_destroy", "dsp_malloc", "dsp_h2d_async",
                 "dsp_trap_filter_f32", "dsp_trap_filter_f64", "dsp_fixed_time_pickoff_f32", "dsp_mean_below_threshold_f32",
                 "dsp_last_error", "dsp_set_device", "dsp_host_register", "dsp_stream_wait_event"):
        assert must in names


def test_library_exports_every_declared_symbol(library):
    missing = [n for n in _declared() if not hasattr(library, n)]
    assert not missing, f"declared in include/dspeed_hip.h but not exported: {missing}"


def test_ctypes_layer_covers_the_header():
    from dspeed_amd import _lib

    declared = set(_declared())
    assert declared == set(_lib.EXPORTS), (sorted(declared - set(_lib.EXPORTS)), sorted(set(_lib.EXPORTS) - declared))


def test_status_messages_are_the_reference_texts(library):
    """DSP_E_* codes carry the reference's DSPFatal messages (e.g. processors/trap_filters.py:54-60); no GPU needed"""
    library.dsp_fatal_message.restype = ctypes.c_char_p
    library.dsp_version.restype = ctypes.c_char_p
    msgs = [library.dsp_fatal_message(c).decode() for c in range(1, 18)]
    assert all(msgs) and len(set(msgs)) == len(msgs)
    assert any("wider than the waveform" in m for m in msgs)
    assert b"gfx950" in library.dsp_version()


def test_product_path_fails_loudly_without_the_library(monkeypatch, tmp_path):
    """no CPU fallback: a missing library is an error, not a detour through the oracle"""
    import importlib

    from dspeed_amd import _lib

    monkeypatch.setenv("DSPEED_HIP_LIB", str(tmp_path / "nope.so"))
    fresh = importlib.reload(_lib)
    try:
        with pytest.raises((OSError, RuntimeError, FileNotFoundError)):
            fresh.lib()
    finally:
        monkeypatch.delenv("DSPEED_HIP_LIB")
        importlib.reload(_lib)


def test_abort_trace_installed_twice_still_terminates(tmp_path):
    """a second dsp_install_abort_trace must not make the handler its own predecessor (an abort() then looped for ever).
    Runs in a child process: install(fd), install(-1), abort() -> one trace, death by SIGABRT within seconds."""
    import signal
    import subprocess
    import sys

    log = tmp_path / "trace.log"
    code = (
        "import ctypes, os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from dspeed_amd import _lib\n"
        "L = ctypes.CDLL(_lib.LIB_PATH)\n"
        f"fd = os.open({str(log)!r}, os.O_WRONLY | os.O_CREAT)\n"
        "assert L.dsp_install_abort_trace(fd) == 0\n"
        "assert L.dsp_install_abort_trace(-1) == 0\n"
        "os.abort()\n"
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == -signal.SIGABRT
    assert r.stderr.count("SIGABRT -- native call stack") == 1


def test_abort_trace_uninstall_restores_the_previous_disposition():
    import signal
    import subprocess
    import sys

    code = (
        "import ctypes, os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from dspeed_amd import _lib\n"
        "L = ctypes.CDLL(_lib.LIB_PATH)\n"
        "assert L.dsp_install_abort_trace(2) == 0 and L.dsp_uninstall_abort_trace() == 0 and L.dsp_uninstall_abort_trace() == 0\n"
        "os.abort()\n"
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == -signal.SIGABRT and "native call stack" not in r.stderr
