"""linear_slope_fit on the rows of a batch, one waveform per lane (dsp_linear_slope_fit_rows, dspeed_amd/csrc/dsp_fit.hip): against the
oracle's bl_subtract -> pole_zero -> linear_slope_fit run processor by processor (linear_slope_fit.py:11-91, PARITY UNPINNED like the
oracle's fit itself), against the in-chain op of the waveform VM, and through the recipe builder, which moves eligible fits of a recipe
there."""
import ctypes as C

import numpy as np
import pytest

import oracle
import recipes
from dspeed_amd import _lib, build_processing_chain
from dspeed_amd.device import DeviceArray, dtype_code, sync
from dspeed_amd.processing_chain import WaveformInput

pytestmark = pytest.mark.gpu
M = "dspeed.processors"


@pytest.fixture(scope="module")
def P():
    from dspeed_amd import processors

    return processors


def _fit_rows(w, fits, sub=None, mode=0, tau=None, ft=np.float32, lo=0, length=None):
    n, full = w.shape
    length = full - lo if length is None else length
    d_w = DeviceArray.from_numpy(w)
    d_sub = DeviceArray.from_numpy(sub) if isinstance(sub, np.ndarray) else None
    out = DeviceArray((4 * len(fits), n), ft)
    win = (_lib.FitWindow * len(fits))(*[_lib.FitWindow(*f) for f in fits])
    rc = _lib.lib().dsp_linear_slope_fit_rows(d_w.ptr + lo * w.dtype.itemsize, dtype_code(w.dtype), n, length, full, dtype_code(ft),
                                              d_sub.ptr if d_sub is not None else None, dtype_code(sub.dtype) if d_sub is not None else 0,
                                              0.0 if sub is None or d_sub is not None else float(sub), mode, int(tau is not None), float(tau or 0.0),
                                              win, len(fits), out.ptr, None)
    _lib.check(rc, what="fit rows")
    sync()
    return out.to_numpy().reshape(len(fits), 4, n)


def _oracle_fits(w, fits, sub, mode, tau, ft):
    y = w.astype(ft)
    if mode == 1:
        y, _ = oracle.bl_subtract(y, sub)
    elif mode == 2:
        y = (y - (np.asarray(sub, dtype=ft)[:, None] if isinstance(sub, np.ndarray) else ft(sub))).astype(ft)
    z = oracle.pole_zero(y, tau)[0] if tau is not None else None
    res = []
    for stage, first, count in fits:
        src = z if stage else y
        res.append(np.stack(oracle.linear_slope_fit(np.ascontiguousarray(src[:, first:first + count]))[:4]))
    return np.stack(res)


def _rows(n, length, dtype, seed):
    rng = np.random.default_rng(seed)
    w = 3000 + 6 * rng.standard_normal((n, length)) + rng.uniform(-0.05, 0.05, (n, 1)) * np.arange(length)[None, :]
    w[:, length // 2:] += rng.uniform(50, 4000, (n, 1)) * np.exp(-np.arange(length - length // 2) / 800.0)[None, :]
    return w.astype(dtype)


def _close(got, want, scale, what):
    assert np.array_equal(np.isnan(got), np.isnan(want)), what
    ok = ~np.isnan(want)
    assert np.max(np.abs(got[ok] - want[ok]) / scale, initial=0.0) <= 2e-6, what


@pytest.mark.parametrize("dtype,ft", [(np.uint16, np.float32), (np.float32, np.float32), (np.int32, np.float64), (np.float64, np.float64)])
def test_fit_rows_against_the_oracle_pipeline(dtype, ft):
    n, length = 130, 1000  # (130 rows: two full wavefronts of waveforms and a partial one)
    w = _rows(n, length, dtype, 5)
    bl = np.random.default_rng(6).uniform(2900, 3100, n).astype(np.float32)
    fits = [(0, 0, 300), (1, 400, 600), (0, 100, 50), (1, 0, 1000)]
    got = _fit_rows(w, fits, bl, 1, 271.25, ft)
    want = _oracle_fits(w, fits, bl.astype(ft), 1, 271.25, ft)
    assert got.dtype == ft
    for k, f in enumerate(fits):
        # the same operation sequence per waveform as the oracle's three processors: mean and deviation to the last bit
        assert np.array_equal(got[k, 0], want[k, 0]) and np.array_equal(got[k, 1], want[k, 1]), f
        _close(got[k, 2], want[k, 2], 0.05, f)
        _close(got[k, 3], want[k, 3], 4000.0, f)


def test_fit_rows_nan_rules():
    n, length = 70, 600
    w = _rows(n, length, np.float32, 8)
    bl = np.full(n, 3000, np.float32)
    w[3, 50], w[7, 599], bl[5] = np.nan, np.nan, np.nan
    fits = [(0, 0, 200), (1, 300, 300), (0, 100, 400)]
    # bl_subtract: a NaN anywhere makes the waveform NaN -> every fit of rows 3, 5, 7
    got = _fit_rows(w, fits, bl, 1, 200.0)
    bad = np.zeros(n, bool)
    bad[[3, 5, 7]] = True
    for k in range(3):
        assert np.array_equal(np.isnan(got[k]).all(axis=0), bad) and not np.isnan(got[k][:, ~bad]).any()
    want = _oracle_fits(w, fits, bl, 1, 200.0, np.float32)
    assert np.array_equal(got[:, :2], want[:, :2], equal_nan=True)
    # numpy.subtract keeps NaN samples single: a fit is NaN if one lies in its window; pole_zero turns any into a NaN waveform
    got = _fit_rows(w, fits, bl, 2, 200.0)
    assert np.isnan(got[0][:, 3]).all() and not np.isnan(got[0][:, 7]).any() and np.isnan(got[0][:, 5]).all()
    assert np.isnan(got[1][:, [3, 5, 7]]).all() and not np.isnan(got[1][:, [0, 1, 2, 4, 6]]).any()
    assert not np.isnan(got[2][:, [3, 7]]).any() and np.isnan(got[2][:, 5]).all()  # (samples 50 and 599 are outside [100, 500))
    # no subtraction, a window of a slice of the rows (pointer offset), constant instead of a column
    a = _fit_rows(w, [(0, 10, 100)], None, 0, None, lo=60, length=400)
    b2 = _fit_rows(np.ascontiguousarray(w[:, 60:460]), [(0, 10, 100)], None, 0, None)
    assert np.array_equal(a, b2, equal_nan=True) and np.isnan(a[0][:, 3]).sum() == 0  # (sample 50 is outside [70, 170))
    c = _fit_rows(w, [(0, 0, 200)], 3000.0, 1, None)
    assert np.array_equal(c[0][:, ~bad], _fit_rows(w, [(0, 0, 200)], bl, 1, None)[0][:, ~bad])
    for args in (dict(fits=[(1, 0, 10)]), dict(fits=[(0, 590, 20)]), dict(fits=[(0, 0, 10)] * 5)):
        with pytest.raises(Exception):
            _fit_rows(w, args["fits"], None, 0, None)


def test_fit_rows_is_the_in_chain_op_bit_for_bit(P):
    """whole-waveform fit of plain rows: the lane-per-waveform kernel and the VM's op run the same operation sequence"""
    x = _rows(200, 750, np.float32, 12)
    got = _fit_rows(x, [(0, 0, 750)])
    vm = P.linear_slope_fit(x)
    for q in range(4):
        assert np.array_equal(got[0, q], vm[q]), q


@pytest.mark.parametrize("host", [False, True])
def test_recipe_fits_run_on_the_rows_and_agree_with_the_in_chain_fits(host, monkeypatch):
    n = 300
    rng = np.random.default_rng(21)
    wf = (_rows(n, 8192, np.float32, 22) * 3).astype(np.uint16)
    tb = {"waveform": WaveformInput(wf if host else DeviceArray.from_numpy(wf), 16.0, 0.0),
          "baseline": (wf[:, :500].mean(axis=1) + rng.uniform(-3, 3, n)).astype(np.float32)}
    outs = ["bl_mean", "bl_std", "bl_slope", "bl_intercept", "pz_mean", "pz_std", "pz_slope", "trapEmax", "tp_0_est", "cuspEmax"]
    outs = [o for o in outs if o in recipes.ICPC["outputs"]]
    res = {}
    for mode in ("rows", "chain"):
        if mode == "chain":
            monkeypatch.setenv("DSPEED_HIP_FIT_IN_CHAIN", "1")
        chain, _, out = build_processing_chain(recipes.ICPC, tb, outputs=outs)
        assert (len(chain._aux) == 1) == (mode == "rows")
        chain.pipeline_bytes = 1 << 20  # (host columns: several pieces, the fit kernel runs per piece)
        chain.execute()
        res[mode] = {k: np.array(v) for k, v in out.items()}
    for k in outs:
        a, b2 = res["rows"][k], res["chain"][k]
        if k.startswith("bl_"):
            assert np.array_equal(a, b2, equal_nan=True), k  # same arithmetic on the same samples
        else:  # downstream of pole_zero: sequential float64 recurrence (rows) against the scan formulation (chain), 1e-12 apart before rounding
            scale = np.nanmax(np.abs(b2)) or 1.0
            assert np.allclose(a, b2, rtol=0, atol=2e-6 * scale, equal_nan=True), k
