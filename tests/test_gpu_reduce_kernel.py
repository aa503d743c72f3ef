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
