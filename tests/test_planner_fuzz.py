"""The chain planner under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build; GPU sanitizers are not available on the pool).

``dsp_chain_create`` used to validate, pack LDS, match the specialised kernels' shapes and allocate device memory in one 800-line function
that no CPU test could reach.  Everything up to the first HIP call now lives in ``dspeed_amd/csrc/dsp_plan.cpp`` (no HIP header, exported as
``dsp_chain_plan``); ``tests/planner_fuzz.cpp`` feeds it random programs -- op lists over every opcode with mostly-valid operands, and the
specialised kernels' shapes with random geometry and single-field mutations -- and checks the invariants of every accepted plan: slots
alive together never share LDS, regions and register file inside the wavefront's LDS and the CU's 160 kB, a specialised kernel's argument
block carrying the bindings' offsets / strides / lengths.  (What this found when first run: four signed overflows on out-of-range slot
lengths, register indices, binding offsets and fit windows -- all now refused with DSP_ERR_ARG / DSP_ERR_TOO_LONG.)"""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dspeed_amd", "csrc")


@pytest.fixture(scope="module")
def fuzzer(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    probe = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not (os.path.isabs(probe) and os.path.exists(probe)):
        pytest.skip("g++ has no AddressSanitizer runtime here")
    exe = str(tmp_path_factory.mktemp("planner_fuzz") / "planner_fuzz")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "planner_fuzz.cpp"), os.path.join(CSRC, "dsp_plan.cpp"),
                           "-o", exe])
    return exe


def _run(exe, programs, seed):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    for k in list(env):
        if k.startswith("DSPEED_HIP_"):  # (the planner reads the library's switches: the default plan is what is fuzzed)
            del env[k]
    r = subprocess.run([exe, str(programs), str(seed)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:] + r.stderr[-6000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_ten_thousand_random_programs_plan_clean_under_asan_and_ubsan(fuzzer):
    rep = _run(fuzzer, 12000, 0xD5BEED)
    assert rep["programs"] == 12000 and rep["accepted"] >= 3000
    # every specialised kernel's shape was reached (and its argument block checked), as were the interpreter and its two-wavefront teams
    for kernel in ("scalar", "pz_rows", "reduce", "current", "fir", "rows", "energy_rr", "vm_team", "vm", "fir_runs"):
        assert rep["kernels"][kernel] > 0, (kernel, rep)
    assert rep["integer_programs"] > 50  # (compute type DSP_I64: the integer programs of round 4)


@pytest.mark.parametrize("seed", [1, 2])
def test_other_seeds(fuzzer, seed):
    rep = _run(fuzzer, 6000, seed)
    assert rep["accepted"] >= 1500


def test_the_library_plans_without_a_device():
    """the same planner through the C ABI of the product library (``dsp_chain_plan``): kernel choice, LDS layout and the reference's
    constant-only DSPFatal conditions, on a machine without a GPU"""
    from dspeed_amd import _lib
    from dspeed_amd.chain import Program, Scalar, energy_chain_program, plan
    from dspeed_amd.errors import DSPFatal

    info = plan(energy_chain_program(4096, 1716.28, 625, 188))
    assert info["kernel"] == "dsp_energy_rr_kernel" and info["note"] == "" and info["slots"][0]["chunk"] == 64
    info = plan(energy_chain_program(4000, 1716.28, 625, 188))
    assert info["kernel"].startswith("dsp_vm_kernel") and "4000 samples" in info["note"]
    with pytest.raises(DSPFatal, match="rise section must be positive"):  # trap_filters.py:53-54
        plan(energy_chain_program(4096, 1716.28, -1, 188))
    with pytest.raises(DSPFatal, match="wider than the waveform"):  # trap_filters.py:59-60
        plan(energy_chain_program(1024, 1716.28, 500, 188))
    # two waveforms that are never alive together share LDS; two that are do not
    p = Program()
    a, b, c = p.add_slot(2048), p.add_slot(2048), p.add_slot(2048)
    io_in = p.add_io("wf", _lib.IO_WF_IN, np.float32, 2048)
    io_out = p.add_io("out", _lib.IO_WF_OUT, np.float32, 2048)
    p.add_op(_lib.OP_LOAD, dst=a, io=io_in)
    p.add_op(_lib.OP_TRAP_FILTER, dst=b, src=a, ip=(10, 5))
    p.add_op(_lib.OP_TRAP_FILTER, dst=c, src=b, ip=(10, 5))
    p.add_op(_lib.OP_STORE, src=c, io=io_out)
    s = plan(p)["slots"]
    span = lambda k: (s[k]["base"], s[k]["base"] + s[k]["elems"])  # noqa: E731
    disjoint = lambda x, y: span(x)[1] <= span(y)[0] or span(y)[1] <= span(x)[0]  # noqa: E731
    assert disjoint(0, 1) and disjoint(1, 2) and not disjoint(0, 2)
    with pytest.raises(ValueError, match="2\\^20 samples"):
        q = Program()
        q.add_slot(1 << 21)
        q.add_io("wf", _lib.IO_WF_IN, np.float32, 1 << 21)
        q.add_op(_lib.OP_LOAD, dst=0, io=0)
        plan(q)
