"""The run-length FIR (dsp_fir_runs.hip): convolve_wf (reference processors/convolutions.py:14-72) with a piecewise-constant kernel -- the t0
filter of the Ge recipes (kernels.py t0_filter: 8 + 125 taps), moving averages, step kernels -- as a handful of differences of float64
prefix sums per output, with the per-event values a recipe reads off the filtered waveform (min_max.py:11-82, time_point_thresh.py:12-92,
fixed_time_pickoff.py:68-80, numpy.amax) taken in the same pass.

The filter is float arithmetic in another order than np.convolve's: the bar is 1e-6 of the filtered waveform's peak against float64 sums
(north_star).  The per-event values are comparisons and selections on the float32 samples the kernel itself produced: bit-identical to
the oracle's min_max / time_point_thresh on those samples, whether the filtered waveform is stored or stays in the wavefront's scratch
row."""
import numpy as np
import pytest

import golden_util
import oracle

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _program(n, offset, stride, taps, mode, keep=True, reductions=True, hint=1, walk_from=("t_max", 100), picks=(0, 57)):
    from dspeed_amd import _lib
    from dspeed_amd.chain import Program, Scalar

    m = len(taps)
    P = {"v": n - m + 1, "s": n, "f": n + m - 1}[mode]
    p = Program()
    p.slots = [n, P]
    wf = p.add_io("wf", _lib.IO_WF_IN, np.float32, n, offset, stride)
    padded = -(-m // 16) * 16
    tp = p.add_io("taps", _lib.IO_TAPS, np.float32, padded, 0, 0)
    p.add_op(_lib.OP_LOAD, dst=0, io=wf)
    p.add_op(_lib.OP_CONVOLVE, dst=1, src=0, io=tp, ip=(ord(mode), 0, hint, m))
    outs = []
    if keep:
        o = p.add_io("filtered", _lib.IO_WF_OUT, np.float32, P, 0, P + 8)
        p.add_op(_lib.OP_STORE, src=1, io=o)
    if reductions:
        thr = p.add_io("thr", _lib.IO_SCALAR_IN, np.float32)
        p.n_sregs = 5
        p.add_op(_lib.OP_MIN_MAX, dst=0, src=1)
        p.add_op(_lib.OP_AMAX, dst=4, src=1)
        outs += [("t_min", 0), ("t_max", 1), ("a_min", 2), ("a_max", 3), ("amax", 4)]
        for k, start in enumerate(walk_from):
            r = p.add_sregs(1)
            s = Scalar.reg(1) if start == "t_max" else (Scalar.reg(0) if start == "t_min" else Scalar.const(float(min(start, P - 1))))
            p.add_op(_lib.OP_TIME_POINT_THRESH, dst=r, src=1, sp=(Scalar.input(thr), s, Scalar.const(float(k))))  # (the first walks back, the second forward)
            outs.append((f"walk{k}", r))
        for k, t in enumerate(picks):
            r = p.add_sregs(1)
            p.add_op(_lib.OP_PICKOFF, dst=r, src=1, ip=(ord("n"), 0), sp=(Scalar.const(float(t)),))
            outs.append((f"pick{k}", r))
        for name, r in outs:
            io = p.add_io(name, _lib.IO_SCALAR_OUT, np.float32)
            p.add_op(_lib.OP_STORE_SCALAR, io=io, ip=(r,))
    return p, P, [name for name, _ in outs]


def _pulses(rng, rows, total):
    i = np.arange(total, dtype=np.float64)[None, :]
    A = rng.uniform(200, 12000, (rows, 1))
    t0 = np.floor(rng.uniform(0.3, 0.7, (rows, 1)) * total)
    rise = rng.uniform(1, 30, (rows, 1))
    x = A * (1 - np.exp(-np.clip(i - t0, 0, None) / rise)) * np.exp(-np.clip(i - t0, 0, None) / 1716.0) + 4.0 * rng.standard_normal((rows, total))
    return x.astype(np.float32)


def _taps(which, rng):
    if which == "t0":
        return golden_util.recipe_kernel("t0")  # the reference's own t0_filter(8, 125), from the golden book
    if which == "boxcar16":
        return np.full(16, 1 / 16, np.float32)
    if which == "step":
        return np.array([1.0] * 5 + [-1.0] * 7, np.float32)
    if which == "one":
        return np.array([-2.5], np.float32)
    if which == "runs24":  # as many runs as the kernel takes, zeros in front, in the middle and at the end
        v = rng.uniform(-1, 1, 24).astype(np.float32)
        v[[0, 11, 23]] = 0.0
        v[12] = v[10]
        return np.repeat(v, rng.integers(1, 22, 24)).astype(np.float32)[:512]
    if which == "long512":
        return np.repeat(np.array([0.25, -0.5, 0.125, 1.0], np.float32), 128)
    raise KeyError(which)


def _execute(prog, outs, P, w, taps, thr, fused, keep):
    from dspeed_amd.chain import Chain
    from dspeed_amd.device import DeviceArray

    n_rows = len(w)
    ch = Chain(prog, "fir runs", np.float32)
    assert ch.set_fused(fused) == bool(fused)
    padded = prog.io[1][3]
    bufs = {"wf": DeviceArray.from_numpy(w), "taps": DeviceArray.from_numpy(np.concatenate([taps, np.zeros(padded - len(taps), np.float32)]))}
    if outs:
        bufs["thr"] = DeviceArray.from_numpy(thr)
    if keep:
        bufs["filtered"] = DeviceArray.zeros((n_rows, P + 8), np.float32)
    for name in outs:
        bufs[name] = DeviceArray.zeros((n_rows,), np.float32)
    ch.execute(bufs, n_rows)
    ch.check()
    got = {name: bufs[name].to_numpy() for name in outs}
    if keep:
        got["filtered"] = bufs["filtered"].to_numpy()[:, :P]
    return ch.kernel_name, got


def _reductions_of(Y, thr, walk_from, picks):
    """what the oracle's processors give on the float32 rows Y"""
    Y = np.ascontiguousarray(Y, dtype=np.float32)
    t_min, t_max, a_min, a_max, _rc = oracle.min_max(Y)
    want = {"t_min": t_min, "t_max": t_max, "a_min": a_min, "a_max": a_max}
    with np.errstate(invalid="ignore"):
        want["amax"] = np.where(np.isnan(Y).any(axis=1), np.float32(np.nan), Y.max(axis=1)).astype(np.float32)
    for k, start in enumerate(walk_from):
        s = t_max if start == "t_max" else (t_min if start == "t_min" else np.full(len(Y), float(min(start, Y.shape[1] - 1)), np.float32))
        want[f"walk{k}"], _rc = oracle.time_point_thresh(Y, np.asarray(thr, np.float32), s.astype(np.float32), k)
    for k, t in enumerate(picks):
        want[f"pick{k}"], _rc = oracle.fixed_time_pickoff(Y, float(t), "n")
    return want


@pytest.mark.parametrize("which,mode,n,offset,stride", [("t0", "s", 8192, 0, 8192), ("t0", "v", 1000, 8, 1016), ("t0", "f", 1024, 0, 1024),
                                                         ("boxcar16", "s", 2048, 4, 2060), ("step", "v", 512, 0, 512), ("step", "f", 64, 0, 64),
                                                         ("one", "s", 256, 0, 256), ("runs24", "s", 4096, 0, 4096), ("runs24", "v", 1024, 0, 1024),
                                                         ("long512", "s", 2048, 0, 2048), ("long512", "v", 512, 0, 512), ("long512", "f", 520, 0, 520)])
def test_filter_and_reductions(which, mode, n, offset, stride):
    rng = np.random.default_rng(n + offset + len(which))
    taps = _taps(which, rng)
    if len(taps) > n:
        pytest.skip("kernel longer than the waveform")
    rows = 1031  # (not a multiple of the four rows of a workgroup)
    w = _pulses(rng, rows, stride)
    x = w[:, offset:offset + n]
    x[1] = 0.0                      # everything equal: both extremes at sample 0
    x[2] = 1234.5                   # a constant: the filter's edges are the only structure
    x[3, n // 2] = np.nan           # -> NaN waveform, NaN values
    x[4, :] = np.nan
    x[5, n // 3] = np.inf           # -> done tap by tap
    x[6, [5, n - 5]] = [-np.inf, np.inf]
    thr = rng.uniform(5, 200, rows).astype(np.float32)
    thr[7] = np.nan
    walk_from, picks = ("t_max", 100), (0, 57)
    prog, P, outs = _program(n, offset, stride, taps, mode, walk_from=walk_from, picks=picks)
    kernel, got = _execute(prog, outs, P, w, taps, thr, 1, True)
    assert "dsp_fir_runs_kernel" in kernel, kernel
    full = {"v": "valid", "s": "same", "f": "full"}[mode]
    k64 = taps.astype(np.float64)
    finite = [r for r in range(rows) if np.isfinite(x[r]).all()]
    worst = 0.0
    for r in finite:
        want = np.convolve(x[r].astype(np.float64), k64, full)
        peak = max(np.abs(want).max(), 1e-30)
        worst = max(worst, np.abs(got["filtered"][r] - want).max() / peak)
    assert worst <= TOL, worst
    for r in (3, 4):
        assert np.isnan(got["filtered"][r]).all()
    for r in (5, 6):  # an infinity: the samples np.convolve multiplies, tap by tap (inf * 0 only where a tap is zero)
        with np.errstate(invalid="ignore", over="ignore"):
            want = np.convolve(x[r].astype(np.float64), k64, full)
        np.testing.assert_array_equal(np.isnan(got["filtered"][r]), np.isnan(want))
        np.testing.assert_array_equal(np.isposinf(got["filtered"][r]), np.isposinf(want))
        np.testing.assert_array_equal(np.isneginf(got["filtered"][r]), np.isneginf(want))
        ok = np.isfinite(want)
        if not ok.any():
            continue
        assert np.abs(got["filtered"][r][ok] - want[ok]).max() <= 1e-5 * max(np.abs(want[ok]).max(), 1.0)
    # the per-event values: the oracle's processors on the samples the kernel stored -- bit for bit
    want = _reductions_of(got["filtered"], thr, walk_from, picks)
    for name in outs:
        np.testing.assert_array_equal(got[name], want[name], err_msg=name)
    # the same values when the filtered waveform is not stored at all (it lives in the wavefront's scratch row)
    prog2, _, outs2 = _program(n, offset, stride, taps, mode, keep=False, walk_from=walk_from, picks=picks)
    kernel2, got2 = _execute(prog2, outs2, P, w, taps, thr, 1, False)
    assert "dsp_fir_runs_kernel" in kernel2
    for name in outs:
        np.testing.assert_array_equal(got2[name], got[name], err_msg=name)
    # and the waveform VM on the same program (float32 sums tap by tap): the filter bar between the two
    kernel3, got3 = _execute(prog, outs, P, w, taps, thr, 0, True)
    assert "dsp_vm_kernel" in kernel3
    for r in finite[:200]:
        if r == 2:
            continue  # (the constant row: where the kernel sums to zero, what is left is the rounding of either form)
        peak = max(np.abs(got3["filtered"][r]).max(), 1e-30)
        f32_sums = 2.0 ** -23 * np.abs(taps).sum() * np.abs(x[r]).max()  # (what a float32 sum over the taps may lose: 512 taps of both signs)
        assert np.abs(got3["filtered"][r] - got["filtered"][r]).max() <= max(4 * TOL * peak, f32_sums)


def test_stored_only_and_many_rounds():
    """LOAD, CONVOLVE, STORE alone (what a recipe stages for a filter other processors read), on more rows than resident wavefronts"""
    rng = np.random.default_rng(5)
    taps = _taps("t0", rng)
    n, rows = 1024, 9000
    w = _pulses(rng, rows, n)
    prog, P, outs = _program(n, 0, n, taps, "s", reductions=False)
    kernel, got = _execute(prog, outs, P, w, taps, None, 1, True)
    assert "dsp_fir_runs_kernel" in kernel
    want = np.stack([np.convolve(w[r].astype(np.float64), taps.astype(np.float64), "same") for r in range(0, rows, 37)])
    assert (np.abs(got["filtered"][::37] - want).max(axis=1) / np.abs(want).max(axis=1)).max() <= TOL
    # the values off scratch rows that are reused round after round
    prog2, _, outs2 = _program(n, 0, n, taps, "s", keep=False)
    thr = rng.uniform(5, 200, rows).astype(np.float32)
    _, got2 = _execute(prog2, outs2, P, w, taps, thr, 1, False)
    want = _reductions_of(got["filtered"], thr, ("t_max", 100), (0, 57))
    for name in outs2:
        np.testing.assert_array_equal(got2[name], want[name], err_msg=name)


def test_a_kernel_that_is_not_piecewise_constant_after_all():
    """the hint is a hint: the taps are a binding, the kernel looks at them ahead of every launch.  More runs than it takes -> every row tap by
    tap (float32, as the fix-up of the matrix-core FIR does them); a NaN among the taps -> NaN (convolutions.py:45-46)"""
    rng = np.random.default_rng(11)
    n, rows = 512, 300
    w = _pulses(rng, rows, n)
    taps = rng.uniform(-1, 1, 40).astype(np.float32)
    prog, P, outs = _program(n, 0, n, taps, "s")
    thr = rng.uniform(5, 200, rows).astype(np.float32)
    kernel, got = _execute(prog, outs, P, w, taps, thr, 1, True)
    assert "dsp_fir_runs_kernel" in kernel
    want = np.stack([np.convolve(w[r].astype(np.float64), taps.astype(np.float64), "same") for r in range(rows)])
    assert (np.abs(got["filtered"] - want).max(axis=1) / np.abs(want).max(axis=1)).max() <= 4 * TOL
    want_v = _reductions_of(got["filtered"], thr, ("t_max", 100), (0, 57))
    for name in outs:
        np.testing.assert_array_equal(got[name], want_v[name], err_msg=name)
    bad = taps.copy()
    bad[7] = np.nan
    _, got = _execute(prog, outs, P, w, bad, thr, 1, True)
    assert np.isnan(got["filtered"]).all() and all(np.isnan(got[name]).all() for name in ("t_min", "a_max", "amax", "walk0", "pick0"))


def test_shapes_the_kernel_does_not_take_stay_where_they_were():
    """no hint, int16 rows, rows off 16-byte boundaries: the matrix-core FIR or the waveform VM, as before"""
    from dspeed_amd.chain import plan

    taps = _taps("t0", None)
    for kwargs, kernel in [(dict(hint=0, reductions=False), "dsp_fir_f16_kernel"), (dict(hint=0), "dsp_vm_kernel"), (dict(hint=1), "dsp_fir_runs_kernel")]:
        prog, _, _ = _program(8192, 0, 8192, taps, "s", **kwargs)
        assert kernel in plan(prog)["kernel"], (kwargs, plan(prog)["kernel"])
    prog, _, _ = _program(8192, 2, 8200, taps, "s")
    assert "dsp_vm_kernel" in plan(prog)["kernel"] and "float32, 16-byte aligned" in plan(prog)["note"]


def test_recipe_with_the_t0_filter():
    """through build_processing_chain: pole_zero -> t0_filter -> convolve_wf -> min_max / time_point_thresh (the t0 estimate of
    icpc-dsp-config.json:69-100), against the oracle end to end"""
    from dspeed_amd.processing_chain import WaveformInput, build_processing_chain

    M = "dspeed.processors"
    rng = np.random.default_rng(3)
    rows, n = 700, 8192
    w = np.rint(_pulses(rng, rows, n) + 300).astype(np.int16)
    recipe = {"outputs": ["tp_0_est", "conv_max", "tp_start", "conv_tmin", "conv_min"], "processors": {
        "wf_blsub": f"{M}.bl_subtract(waveform, baseline, wf_blsub)",
        "wf_pz": f"{M}.pole_zero(wf_blsub, 1716*16*ns, wf_pz)",
        "t0_kernel": {"function": "t0_filter", "module": M, "args": ["128*ns/wf_pz.period", "2*us/wf_pz.period", "t0_kernel(round((128*ns+2*us)/wf_pz.period), 'f')"]},
        "wf_t0_filter": {"function": "convolve_wf", "module": M, "args": ["wf_pz", "t0_kernel", "'s'", "wf_t0_filter(len(wf_pz), 'f', grid=wf_pz.grid)"]},
        "conv_tmin, tp_start, conv_min, conv_max": f"{M}.min_max(wf_t0_filter, conv_tmin, tp_start, conv_min, conv_max)",
        "tp_0_est": f"{M}.time_point_thresh(wf_t0_filter, thr, tp_start, 0, tp_0_est(unit=ns))"}}
    thr = rng.uniform(2, 40, rows).astype(np.float32)
    bl = np.full(rows, 300.0, np.float32)
    tb = {"waveform": WaveformInput(w, 16.0, 0.0), "baseline": bl, "thr": thr}
    chain, _, out = build_processing_chain(recipe, tb)
    chain.execute()
    assert any("dsp_fir_runs_kernel" in k for _w, k in chain.kernels()), chain.kernels()
    taps = golden_util.recipe_kernel("t0")
    for r in range(0, rows, 7):
        pz = oracle.pole_zero((w[r].astype(np.float32) - bl[r]).astype(np.float32), 1716.0)[0][0]
        y64 = np.convolve(pz.astype(np.float64), taps.astype(np.float64), "same")
        t_min, t_max, a_min, a_max = (v[0] for v in oracle.min_max(y64.astype(np.float32))[:4])
        assert abs(out["conv_max"][r] - a_max) <= 4 * TOL * np.abs(y64).max() and abs(out["conv_min"][r] - a_min) <= 4 * TOL * np.abs(y64).max()
        if out["tp_start"][r] == t_max * 16.0:  # (an extreme decided within the bar may sit elsewhere: then the walk starts elsewhere too)
            want = oracle.time_point_thresh(y64.astype(np.float32), thr[r], t_max, 0)[0][0]
            got = out["tp_0_est"][r]
            assert (np.isnan(want) and np.isnan(got)) or abs(got - want * 16.0) <= 16.0, (r, got, want)
