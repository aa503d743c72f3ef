"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only: GPU sanitizers are not available on the pool): the C
restatement is what every parity claim rests on, so an out-of-bounds read or signed overflow in it would be a silent error in the
checker.  Builds oracle/_san/libdsp_oracle_san.so and runs the golden-vector suite against it in a child process."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    out = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_golden_vectors_under_asan_and_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc has no AddressSanitizer runtime here")
    out_dir = os.path.join(ROOT, "oracle", "_san")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libdsp_oracle_san.so")
    subprocess.check_call(["gcc", "-O1", "-g", "-fno-fast-math", "-ffp-contract=off", "-fopenmp", "-fPIC", "-std=c11", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-shared", "-o", lib,
                           os.path.join(ROOT, "oracle", "dsp_oracle.c"), "-lm"])
    env = dict(os.environ, DSP_ORACLE_LIB=lib, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "passed" in r.stdout
