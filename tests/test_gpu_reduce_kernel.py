"""Per-event values read straight off rows (dsp_reduce.hip): min_max, numpy.amax and a sample at a constant integral time of float32 /
int16 / uint16 rows, streamed through registers.  Comparisons only, so every output is bit-identical to the oracle (reference
processors/min_max.py:11-82: first occurrence of the extremes, four NaNs for a row with a NaN; fixed_time_pickoff.py:68-80) and to the
waveform VM running the same program."""
import numpy as np
import pytest

import oracle
import recipes

pytestmark = pytest.mark.gpu


def _program(dtype, length, offset, stride, picks, with_amax=True, walks=()):
    from dspeed_amd import _lib
    from dspeed_amd.chain import Program, Scalar

    p = Program()
    p.slots = [length]
    p.n_sregs = 5 + len(picks)
    wf = p.add_io("wf", _lib.IO_WF_IN, dtype, length, offset, stride)
    p.add_op(_lib.OP_LOAD, dst=0, io=wf)
    p.add_op(_lib.OP_MIN_MAX, dst=0, src=0)
    if with_amax:
        p.add_op(_lib.OP_AMAX, dst=4, src=0)
    for k, (t, kind) in enumerate(picks):
        p.add_op(_lib.OP_PICKOFF, dst=5 + k, src=0, ip=(ord("n"), kind), sp=(Scalar.const(float(t)),))
    outs = ["t_min", "t_max", "a_min", "a_max"] + (["amax"] if with_amax else []) + [f"pick{k}" for k in range(len(picks))]
    regs = [0, 1, 2, 3] + ([4] if with_amax else []) + [5 + k for k in range(len(picks))]
    if walks:
        thr = p.add_io("thr", _lib.IO_SCALAR_IN, np.float32)
        for k, (from_, forward, thr_const) in enumerate(walks):
            r = p.n_sregs
            p.n_sregs += 1
            start = Scalar.reg(1) if from_ == "t_max" else (Scalar.reg(0) if from_ == "t_min" else Scalar.const(float(from_)))
            p.add_op(_lib.OP_TIME_POINT_THRESH, dst=r, src=0, sp=(Scalar.input(thr) if thr_const is None else Scalar.const(float(thr_const)), start, Scalar.const(float(forward))))
            outs.append(f"walk{k}")
            regs.append(r)
    for name, r in zip(outs, regs):
        io = p.add_io(name, _lib.IO_SCALAR_OUT, np.float32, 1, r % 2, 2)  # (interleaved columns: offsets and strides on the outputs)
        p.add_op(_lib.OP_STORE_SCALAR, io=io, ip=(r,))
    return p, outs


def _rows(rng, n, total, dtype):
    if np.dtype(dtype) == np.float32:
        w = rng.normal(0, 1000, (n, total)).astype(np.float32)
    else:
        info = np.iinfo(dtype)
        w = rng.integers(info.min, info.max + 1, (n, total)).astype(dtype)
    return w


@pytest.mark.parametrize("dtype,length,offset,stride", [(np.uint16, 8192, 0, 8192), (np.int16, 8192, 0, 8192), (np.float32, 8192, 0, 8192),
                                                        (np.float32, 301, 0, 301), (np.uint16, 1001, 3, 1011), (np.float32, 4096, 8, 4200),
                                                        (np.int16, 64, 0, 64), (np.float32, 5, 0, 5)])
def test_reductions_off_rows(dtype, length, offset, stride):
    from dspeed_amd.chain import Chain
    from dspeed_amd.device import DeviceArray

    rng = np.random.default_rng(length + offset)
    n = 517  # (not a multiple of the four rows a workgroup takes)
    w = _rows(rng, n, stride, dtype)
    rows = w[:, offset:offset + length]
    rows[1, :] = rows[1, 0]                     # a constant row: both extremes at sample 0
    rows[2, [length // 3, length - 1]] = rows[2].max()  # the maximum twice: the first one counts
    rows[3, [0, length // 2]] = rows[3].min()
    if np.dtype(dtype) == np.float32:
        rows[4, length // 2] = np.nan           # min_max: four NaNs; amax: NaN; fixed_time_pickoff: NaN; the plain sample: itself
        rows[5, :] = np.nan
        rows[6, length - 1] = np.inf
        rows[7, 0] = -np.inf
        rows[8, :] = 0.0
        rows[8, length // 4] = -0.0             # -0.0 < 0.0 is false: sample 0 stays the minimum, with its sign
    picks = [(0, 0), (length - 1, 0), (length, 0), (-1, 0), ]
    picks = picks[:3] + [(min(2, length - 1), 1)]
    mid = 30000.0 if np.dtype(dtype) == np.uint16 else 1.0
    walks = [("t_max", 0, None), ("t_min", 1, mid)] if length >= 64 else []
    prog, outs = _program(dtype, length, offset, stride, picks, walks=walks)
    thr = rng.uniform(-500, 500, n).astype(np.float32) if np.dtype(dtype) != np.uint16 else rng.uniform(20000, 40000, n).astype(np.float32)
    thr[9] = np.nan
    got = {}
    for fused in (1, 0):
        ch = Chain(prog, "reductions", np.float32)
        assert ch.set_fused(fused) == bool(fused)
        assert ("dsp_reduce_kernel" in ch.kernel_name) == bool(fused), ch.kernel_name
        bufs = {"wf": DeviceArray.from_numpy(w), "thr": DeviceArray.from_numpy(thr)}
        for name in outs:
            bufs[name] = DeviceArray.zeros((n, 2), np.float32)
        ch.execute(bufs, n)
        ch.check()
        got[fused] = {name: bufs[name].to_numpy() for name in outs}
    regs_col = [r % 2 for r in range(len(outs))]  # (registers 0 .. in the order of the outputs)
    for k, name in enumerate(outs):
        col = regs_col[k]
        assert np.array_equal(got[1][name][:, col], got[0][name][:, col], equal_nan=True), name
        assert np.all(got[1][name][:, 1 - col] == 0), "only the binding's own column is written"
    f = np.ascontiguousarray(rows).astype(np.float32)
    t_min, t_max, a_min, a_max, rc = oracle.min_max(f)
    assert rc == 0
    g = {name: got[1][name][:, regs_col[k]] for k, name in enumerate(outs)}
    for name, want in (("t_min", t_min), ("t_max", t_max), ("a_min", a_min), ("a_max", a_max)):
        assert np.array_equal(g[name], want, equal_nan=True), name
        assert np.array_equal(np.signbit(g[name]), np.signbit(want)), name
    with np.errstate(invalid="ignore"):
        assert np.array_equal(g["amax"], np.max(f, axis=1), equal_nan=True)
    for k, (from_, forward, thr_const) in enumerate(walks):
        start = t_max if from_ == "t_max" else t_min
        ok = ~np.isnan(start)
        want = np.full(n, np.nan, dtype=np.float32)
        th = thr if thr_const is None else np.full(n, thr_const, dtype=np.float32)
        res, rc = oracle.time_point_thresh(f[ok], th[ok], start[ok], float(forward))
        assert rc == 0
        want[ok] = res
        assert np.array_equal(g[f"walk{k}"], want, equal_nan=True), (k, from_, forward)
        assert np.isnan(g["walk0"][9]) and (~np.isnan(g[f"walk{k}"])).sum() > n // 8
    for k, (t, kind) in enumerate(picks):
        if kind == 0:
            want, rc = oracle.fixed_time_pickoff(f, float(t), "n")
            assert rc == 0
        else:
            want = f[:, t]
        assert np.array_equal(g[f"pick{k}"], want, equal_nan=True), (k, t, kind)


def test_shapes_the_kernel_leaves_to_the_program():
    from dspeed_amd import _lib
    from dspeed_amd.chain import Chain, Scalar

    prog, _ = _program(np.float32, 256, 0, 256, [(10.5, 0)])  # between two samples: the interpolating pick-off
    assert "vm" in Chain(prog, "x", np.float32).kernel_name
    prog, _ = _program(np.float32, 256, 0, 256, [])
    prog.ops.insert(1, (_lib.OP_BL_SUBTRACT, 0, 0, 0, (0,), (Scalar.const(1.0),)))  # something between the load and the reductions
    assert "dsp_reduce_kernel" not in Chain(prog, "x", np.float32).kernel_name


def test_ge_recipe_reads_them_off_the_rows(monkeypatch):
    from dspeed_amd.processing_chain import WaveformInput, build_processing_chain
    from test_gpu_icpc_recipe import _synth

    rng = np.random.default_rng(3)
    n = 300
    wf, bl = _synth(rng, n)
    tb = {"waveform": WaveformInput(wf, 16.0, (rng.integers(2900, 3100, n) * 16).astype(np.float32)), "baseline": bl}
    chain, _, out = build_processing_chain(recipes.ICPC, tb)
    chain.execute()
    ks = dict(chain.kernels())
    # (min_max of the raw waveform goes along with the kernel that writes the pole-zero rows: it streams the raw rows anyway)
    assert ks["wf_pz -> HBM + min_max of waveform"] == "dsp_pz_rows_kernel" and ks["per-event values of wf_cusp off its rows"] == "dsp_reduce_kernel"
    monkeypatch.setenv("DSPEED_HIP_NO_ROW_REDUCTIONS", "1")
    whole, _, ref = build_processing_chain(recipes.ICPC, tb)
    assert not any("off its rows" in w or "min_max of" in w for w, _k in whole.kernels())
    whole.execute()
    for k in ref:
        assert np.array_equal(np.asarray(out[k]), np.asarray(ref[k]), equal_nan=True), k
    t_min, t_max, a_min, a_max, rc = oracle.min_max(wf.astype(np.float32))
    assert np.array_equal(out["wf_min"], a_min) and np.array_equal(out["wf_max"], a_max)


@pytest.mark.parametrize("promise", [1, 0])
def test_walks_from_per_event_starts_with_fractions_of_a_column_as_thresholds(promise):
    """the rise-time walks of a recipe as a launch of their own: time_point_thresh (time_point_thresh.py:12-92) whose threshold is a fraction of a
    per-event column (SCALAR_AFFINE in front of it) and whose start is a column or where an earlier walk ended -- a chain of four and one beside
    it, as the Ge recipes have them.  Comparisons and one float multiplication: the interpreter's bits, and the oracle's walk by walk; a
    fractional or outside start is the processor's DSPFatal, a NaN operand a NaN."""
    from dspeed_amd import _lib
    from dspeed_amd.chain import Chain, Program, Scalar
    from dspeed_amd.device import DeviceArray
    from dspeed_amd.errors import DSPFatal

    rng = np.random.default_rng(77 + promise)
    n, L = 1500, 8192
    i = np.arange(L)[None, :]
    t0 = np.floor(rng.uniform(0.3, 0.6, (n, 1)) * L)
    rise = rng.uniform(1, 60, (n, 1))
    amp = rng.uniform(500, 15000, (n, 1))
    w = (amp * np.clip((i - t0) / rise, 0, 1) * np.exp(-np.clip(i - t0 - rise, 0, None) / 30000.0) + 4.0 * rng.standard_normal((n, L))).astype(np.float32)
    peak = w.max(axis=1).astype(np.float32)
    ts = (t0[:, 0] - 40).astype(np.float32)
    w[3] = np.nan             # a NaN waveform
    peak[4] = np.nan          # a NaN threshold
    ts[5] = np.nan            # a NaN start
    w[6] = 0.0                # nothing to cross: the first walk finds nothing, the walks that start from it are NaN
    p = Program()
    p.slots = [L]
    p.n_sregs = 12
    wf = p.add_io("wf", _lib.IO_WF_IN, np.float32, L, 0, L)
    pk = p.add_io("peak", _lib.IO_SCALAR_IN, np.float32)
    st = p.add_io("ts", _lib.IO_SCALAR_IN, np.float32)
    p.add_op(_lib.OP_LOAD, dst=0, io=wf, ip=(0, 0, promise))
    p.add_op(_lib.OP_SCALAR_AFFINE, dst=8, sp=(Scalar.input(pk), Scalar.const(0.99), Scalar.const(-0.0)))
    p.add_op(_lib.OP_TIME_POINT_THRESH, dst=0, src=0, sp=(Scalar.reg(8), Scalar.input(st), Scalar.const(1.0)))   # forward from the column
    for k, frac in enumerate((0.9, 0.5, 0.1)):                                                                    # backward, each from the one before
        p.add_op(_lib.OP_SCALAR_AFFINE, dst=9 + k, sp=(Scalar.input(pk), Scalar.const(frac), Scalar.const(0.0)))
        p.add_op(_lib.OP_TIME_POINT_THRESH, dst=1 + k, src=0, sp=(Scalar.reg(9 + k), Scalar.reg(k), Scalar.const(0.0)))
    p.add_op(_lib.OP_TIME_POINT_THRESH, dst=4, src=0, sp=(Scalar.input(pk), Scalar.input(st), Scalar.const(1.0)))  # the peak itself, forward
    outs = [f"walk{k}" for k in range(5)]
    for k, name in enumerate(outs):
        p.add_op(_lib.OP_STORE_SCALAR, io=p.add_io(name, _lib.IO_SCALAR_OUT, np.float32), ip=(k,))
    got = {}
    for fused in (1, 0):
        ch = Chain(p, "walks", np.float32)
        assert ch.set_fused(fused) == bool(fused) and ("dsp_reduce_kernel" in ch.kernel_name) == bool(fused), ch.kernel_name
        bufs = {"wf": DeviceArray.from_numpy(w), "peak": DeviceArray.from_numpy(peak), "ts": DeviceArray.from_numpy(ts)}
        bufs.update({name: DeviceArray.zeros((n,), np.float32) for name in outs})
        ch.execute(bufs, n)
        ch.check()
        got[fused] = {name: bufs[name].to_numpy() for name in outs}
    for name in outs:
        np.testing.assert_array_equal(got[1][name], got[0][name], err_msg=name)
    # the oracle, walk by walk (its inputs: the float32 thresholds the device formed)
    g = got[1]
    starts = [ts, g["walk0"], g["walk1"], g["walk2"], ts]
    thrs = [np.float32(0.99) * peak, np.float32(0.9) * peak, np.float32(0.5) * peak, np.float32(0.1) * peak, peak]
    for k in range(5):
        ok = ~(np.isnan(starts[k]) | np.isnan(thrs[k]) | np.isnan(w).any(axis=1))
        want = np.full(n, np.nan, np.float32)
        res, rc = oracle.time_point_thresh(w[ok], thrs[k][ok], starts[k][ok], 1.0 if k in (0, 4) else 0.0)
        assert rc == 0
        want[ok] = res
        np.testing.assert_array_equal(g[f"walk{k}"], want, err_msg=f"walk{k}")
    assert np.isnan(g["walk0"][[3, 4, 5, 6]]).all() and np.isnan(g["walk3"][6]) and (~np.isnan(g["walk3"])).sum() > n // 2
    # the processor's DSPFatal for a start between two samples / outside the waveform, with its row
    for bad, row in ((1234.5, 17), (float(L), 18), (-1.0, 19)):
        ts2 = ts.copy()
        ts2[row] = bad
        ch = Chain(p, "walks", np.float32)
        bufs = {"wf": DeviceArray.from_numpy(w), "peak": DeviceArray.from_numpy(peak), "ts": DeviceArray.from_numpy(ts2)}
        bufs.update({name: DeviceArray.zeros((n,), np.float32) for name in outs})
        ch.execute(bufs, n)
        with pytest.raises(DSPFatal) as e:
            ch.check()
        assert e.value.wf_range is not None and row in e.value.wf_range, (bad, e.value.wf_range)
