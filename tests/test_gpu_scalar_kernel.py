"""The row-per-lane kernel for programs of scalar ops only (dsp_scalar.hip) against the waveform VM's own ops on the same program: every
output bit-identical, in the float32 and the float64 loop, for every function of the language, the coordinate conversion with each
rounding mode, bool outputs, per-row operands of any input type and binding offsets; and the whole Ge recipe with and without the cut."""
import numpy as np
import pytest

import recipes

pytestmark = pytest.mark.gpu


def _program(ft):
    from dspeed_amd import _lib
    from dspeed_amd.chain import Program, Scalar

    p = Program()
    p.n_sregs = 40
    io_a = p.add_io("a", _lib.IO_SCALAR_IN, ft)
    io_b = p.add_io("b", _lib.IO_SCALAR_IN, np.float32, 1, 1, 3)   # element 1 of rows of 3
    io_i = p.add_io("i", _lib.IO_SCALAR_IN, np.int32)
    io_u = p.add_io("u", _lib.IO_SCALAR_IN, np.uint16)
    A, B, I, U = Scalar.input(io_a), Scalar.input(io_b), Scalar.input(io_i), Scalar.input(io_u)
    c = Scalar.const
    r = 0
    outs = []

    def emit(opcode, ip=(), sp=(), bool_out=False):
        nonlocal r
        p.add_op(opcode, dst=r, ip=ip, sp=sp)
        name = f"o{r}"
        io = p.add_io(name, _lib.IO_SCALAR_OUT, np.bool_ if bool_out else ft, 1, r % 2, 2)  # (interleaved columns: an offset on outputs too)
        p.add_op(_lib.OP_STORE_SCALAR, io=io, ip=(r,))
        outs.append((name, bool_out))
        r += 1
        return Scalar.reg(r - 1)

    x = emit(_lib.OP_SCALAR_AFFINE, sp=(A, c(1.000001), B))
    y = emit(_lib.OP_SCALAR_AFFINE, sp=(I, U, c(-3.25)))
    emit(_lib.OP_SCALAR_DIV, sp=(x, y))
    emit(_lib.OP_SCALAR_DIV, sp=(A, c(0.0)))
    for fn in range(16):
        emit(_lib.OP_SCALAR_FUNC, ip=(fn,), sp=(x, y if fn != _lib.FN_WHERE else A, B), bool_out=fn in (_lib.FN_LT, _lib.FN_LE, _lib.FN_GT, _lib.FN_GE,
                                                                                                       _lib.FN_EQ, _lib.FN_NE, _lib.FN_ISNAN, _lib.FN_ISFINITE))
    for mode in range(5):
        emit(_lib.OP_SCALAR_CONVERT, ip=(mode,), sp=(x, B, I, c(1.0 / 16.0)))
        emit(_lib.OP_SCALAR_CONVERT, ip=(mode,), sp=(y, c(3000.25), c(-17.5), c(16.0)))
    return p, outs


@pytest.mark.parametrize("ft", [np.float32, np.float64])
def test_scalar_programs_row_per_lane_equal_the_vm(ft):
    from dspeed_amd.chain import Chain
    from dspeed_amd.device import DeviceArray

    rng = np.random.default_rng(11)
    n = 1000  # (not a multiple of 64: the last wavefront has idle lanes)
    a = rng.normal(0, 1000, n).astype(ft)
    a[[3, 77]] = np.nan
    a[5] = np.inf
    b = rng.normal(0, 10, (n, 3)).astype(np.float32)
    i = rng.integers(-100000, 100000, n).astype(np.int32)
    u = rng.integers(0, 65535, n).astype(np.uint16)
    prog, outs = _program(ft)
    got = {}
    for fused in (1, 0):
        ch = Chain(prog, "scalars", ft)
        assert ch.set_fused(fused) == bool(fused)
        assert ("dsp_scalar_kernel" in ch.kernel_name) == bool(fused)
        bufs = {"a": DeviceArray.from_numpy(a), "b": DeviceArray.from_numpy(b), "i": DeviceArray.from_numpy(i), "u": DeviceArray.from_numpy(u)}
        for name, is_bool in outs:
            bufs[name] = DeviceArray.zeros((n, 2), np.bool_ if is_bool else ft)
        ch.execute(bufs, n)
        ch.check()
        got[fused] = {name: bufs[name].to_numpy() for name, _ in outs}
    for name, _ in outs:
        assert np.array_equal(got[1][name], got[0][name], equal_nan=True), name
    # a few of them against NumPy in the loop's type
    x = a * ft(1.000001) + b[:, 1].astype(ft)
    assert np.array_equal(got[1]["o0"][:, 0], x, equal_nan=True)
    with np.errstate(all="ignore"):
        assert np.array_equal(got[1]["o3"][:, 1], a / ft(0.0), equal_nan=True)


def test_whole_recipe_with_and_without_the_scalar_tail(monkeypatch):
    from dspeed_amd.processing_chain import WaveformInput, build_processing_chain
    from test_gpu_icpc_recipe import _synth

    rng = np.random.default_rng(12)
    n = 200
    wf, bl = _synth(rng, n)
    tb = {"waveform": WaveformInput(wf, 16.0, (rng.integers(2900, 3100, n) * 16).astype(np.float32)), "baseline": bl}
    chain, _, out = build_processing_chain(recipes.ICPC, tb)
    assert chain._tail is not None
    chain.execute()
    assert "dsp_scalar_kernel" in chain._lanes[0].tail.kernel_name
    monkeypatch.setenv("DSPEED_HIP_NO_SCALAR_TAIL", "1")
    whole, _, ref = build_processing_chain(recipes.ICPC, tb)
    assert whole._tail is None
    whole.execute()
    for k in ref:
        assert np.array_equal(out[k], ref[k], equal_nan=True), k
