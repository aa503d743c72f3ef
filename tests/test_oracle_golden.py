"""Pins the CPU oracle (oracle/dsp_oracle.c):

1. against the reference's own known-answer tests, restated here with the reference file:line, and
2. against the fixtures in tests/golden produced by executing the reference kernel bodies
   (oracle/gen_golden.py).

Bars: bit-exact wherever the reference arithmetic is a fixed sequence of IEEE operations
(every nopython kernel); 1e-6 of max|out| for np.convolve-based outputs whose float32 summation
order is NumPy-internal.
"""
import numpy as np
import pytest

import oracle
from golden_util import assert_rel_to_peak, cases, zerodiv


def _eq(got, want, what):
    got, want = np.asarray(got), np.asarray(want)
    assert got.dtype == want.dtype, f"{what}: dtype {got.dtype} != {want.dtype}"
    assert np.array_equal(got, want, equal_nan=True), f"{what}: not bit-identical, max diff {np.nanmax(np.abs(got - want))}"


# ------------------------------------------------------------------ reference known answers (restated)
def test_known_pole_zero_step():
    """reference tests/processors/test_pole_zero.py:14-48"""
    tau, amp = 30000, 17500
    ts = np.arange(0, 8192, dtype=np.float64)
    expected = np.concatenate([np.zeros(20), np.full(len(ts), amp)])
    for dt, rtol in ((np.float32, 1e-6), (np.float64, 1e-7)):
        pulse = np.zeros(len(ts) + 20, dtype=dt)
        pulse[20:] = amp * np.exp(-ts / tau)
        out, rc = oracle.pole_zero(pulse, tau)
        assert rc == 0 and out.dtype == dt
        assert np.allclose(out[0], expected, rtol=rtol)
    w = np.ones(100)
    w[4] = np.nan
    out, rc = oracle.pole_zero(w, tau)
    assert rc == 0 and np.isnan(out).all()


def test_known_double_pole_zero_step():
    """reference tests/processors/test_pole_zero.py:51-96"""
    wf_len, tp0, amp, tau1, tau2, frac = 8192, 20, 17500, 1000, 30000, 0.98
    ts = np.arange(0, wf_len - tp0, dtype=np.float64)
    ys = amp * (1 - frac) * np.exp(-ts / tau1) + amp * frac * np.exp(-ts / tau2)
    expected = np.full(wf_len, amp, dtype=float)
    expected[:tp0] = 0
    for dt, rtol in ((np.float64, 1e-7), (np.float32, 1e-6)):
        pulse = np.zeros(wf_len, dtype=dt)
        pulse[tp0:] = ys
        out, rc = oracle.double_pole_zero(pulse, tau1, tau2, frac)
        assert rc == 0
        assert np.allclose(out[0], expected, rtol=rtol)
    _, rc = oracle.double_pole_zero(np.ones(2), tau1, tau2, frac)
    assert oracle.E_NAMES[rc] == "DPZ_SHORT"
    w = np.ones(wf_len)
    w[4] = np.nan
    out, rc = oracle.double_pole_zero(w, tau1, tau2, frac)
    assert rc == 0 and np.isnan(out).all()


def test_known_fixed_time_pickoff():
    """reference tests/processors/test_fixed_time_pickoff.py:15-108"""
    n = 20
    w = np.ones(n)
    w[4] = np.nan
    assert np.isnan(oracle.fixed_time_pickoff(w, 1, "i")[0][0])
    w = np.ones(n)
    for t in (np.nan, -1, n):
        out, rc = oracle.fixed_time_pickoff(w, t, "i")
        assert rc == 0 and np.isnan(out[0])
    assert oracle.E_NAMES[oracle.fixed_time_pickoff(w, 1.5, "i")[1]] == "FTP_INT"
    assert oracle.E_NAMES[oracle.fixed_time_pickoff(w, 1.5, " ")[1]] == "FTP_MODE"
    ramp = np.arange(n, dtype=float)
    assert oracle.fixed_time_pickoff(ramp, 3, "i")[0][0] == 3
    for ch, sol in zip("nfclhs", [4, 3, 4, 3.5, 3.5, 3.5]):
        assert oracle.fixed_time_pickoff(ramp, 3.5, ch)[0][0] == sol
    sine = np.sin(np.arange(n))
    sols = [0.1411200080598672, 0.1411200080598672, -0.7568024953079282, -0.08336061778208165, -0.09054574599004982,
            -0.10707938709427486]
    for ch, sol in zip("nfclhs", sols):
        assert np.isclose(oracle.fixed_time_pickoff(sine, 3.25, ch)[0][0], sol)
    for ftp, sol in zip([0.2, n - 1.8], [0.1806725096462211, -0.6150034250096629]):
        assert np.isclose(oracle.fixed_time_pickoff(sine, ftp, "h")[0][0], sol)


def test_known_time_point_thresh():
    """reference tests/processors/test_time_point_thresh.py:13-115"""
    saw = np.concatenate([np.arange(-1, 5, 1), np.arange(-1, 5, 1)]).astype(float)
    w = saw.copy()
    w[4] = np.nan
    assert np.isnan(oracle.time_point_thresh(w, 1, 11, 0)[0][0])
    assert np.isnan(oracle.time_point_thresh(saw, np.nan, 11, 0)[0][0])
    assert np.isnan(oracle.time_point_thresh(saw, 1, np.nan, 0)[0][0])
    assert np.isnan(oracle.time_point_thresh(saw, 1, 11, np.nan)[0][0])
    assert oracle.E_NAMES[oracle.time_point_thresh(saw, 1, 10.5, 0)[1]] == "TPT_START_INT"
    assert oracle.E_NAMES[oracle.time_point_thresh(saw, 1, 11, 0.5)[1]] == "TPT_WALK_INT"
    assert oracle.E_NAMES[oracle.time_point_thresh(saw, 1, 12, 0)[1]] == "TPT_RANGE"
    assert oracle.time_point_thresh(saw, 1, 11, 0)[0][0] == 8.0
    assert oracle.time_point_thresh(saw, 3, 0, 1)[0][0] == 4.0
    assert oracle.time_point_thresh(np.array([5.0, 4, 3, 2, 1, 0, -1]), 2.5, 0, 1)[0][0] == 2.0
    assert oracle.time_point_thresh(np.array([0.0, 1, 2, 3, 4, 5]), 2.5, 0, 1)[0][0] == 2.0
    assert oracle.time_point_thresh(np.array([-5.0, -4, -3, -2, -1, 0]), -2.5, 0, 1)[0][0] == 2.0
    assert oracle.time_point_thresh(np.array([0.0, -1, -2, -3, -4, -5]), -2.5, 0, 1)[0][0] == 2.0


def test_known_interpolated_time_point_thresh():
    """reference tests/processors/test_time_point_thresh.py:118-218"""
    saw = np.concatenate([np.arange(-1, 5, 1), np.arange(-1, 5, 1)]).astype(float)
    w = saw.copy()
    w[4] = np.nan
    f = lambda *a: oracle.interpolated_time_point_thresh(*a)[0][0]  # noqa: E731
    assert np.isnan(f(w, 1.0, 11.0, 0, 105)) and np.isnan(f(saw, np.nan, 11.0, 0, 105)) and np.isnan(f(saw, 1.0, np.nan, 0, 105))
    assert np.isnan(f(saw, 1.0, 12, 0, 105))
    for thr, ts, walk, mode, want in ((1, 11, 0, 105, 7.0), (3, 0, 1, 105, 4.0), (1, 11, 0, 102, 8.0), (3, 0, 1, 102, 5.0),
                                      (1, 11, 0, 99, 7.0), (3, 0, 1, 99, 4.0), (1, 11, 0, 110, 7.5), (3, 0, 1, 110, 4.5),
                                      (1.5, 11, 0, 108, 8.5), (3.5, 0, 1, 108, 4.5)):
        assert f(saw, thr, ts, walk, mode) == want, (thr, ts, walk, chr(mode))


def test_known_dwt():
    """reference tests/processors/test_dwt.py:8-43"""
    out, rc = oracle.dwt_haar(np.ones(16), 2, "a", 4)
    assert rc == 0 and np.allclose(out[0], np.ones(4) * 2 ** (2 / 2))
    assert oracle.E_NAMES[oracle.dwt_haar(np.ones(16), -1, "a", 4)[1]] == "DWT_LEVEL"
    w = np.ones(16)
    w[4] = np.nan
    out, rc = oracle.dwt_haar(w, 2, "a", 4)
    assert rc == 0 and np.isnan(out).all()


def test_known_edge_semantics():
    """SURVEY.md 8(a) edge semantics, observed by executing the reference bodies."""
    r16 = np.arange(1, 17, dtype=np.float32)
    assert np.isnan(oracle.trap_filter(r16, 0, 3)[0]).all()
    assert list(oracle.trap_filter(r16, 2, 0)[0][0][:5]) == [1, 3, 4, 4, 4]
    assert list(oracle.trap_filter(r16, 1, 1)[0][0][:4]) == [1, 2, 2, 2]
    assert list(oracle.asym_trap_filter(r16, 2, 1, 4)[0][0][:7]) == [0.5, 1.5, 2.5, 3.25, 3.75, 4, 4]
    tmin, tmax, amin, amax, rc = oracle.min_max(np.array([3, 1, 1, 5, 5, 2], dtype=np.float32))
    assert (tmin[0], tmax[0], amin[0], amax[0]) == (1, 3, 1, 5)


# ------------------------------------------------------------------ golden fixtures
def _eq_or_libm(c, got, want):
    """float32 loop: bit-exact.  float64 loop: exp(-1/tau) comes from libm in the oracle (as under numba/LLVM) but from
    NumPy's own SIMD exp in the golden generator; the two may differ in the last bit, which is visible only in float64."""
    if c.tag == "f32":
        _eq(got, want, c.name)
    else:
        fin = np.isfinite(want)
        assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)])
        assert_rel_to_peak(np.where(fin, got, 0), np.where(fin, want, 0), 1e-13, c.name)


def _check_fatal(c, rc):
    assert (rc != 0) == c.fatal, f"{c}: oracle rc={oracle.E_NAMES.get(rc)} but golden fatal={c.fatal}"


@pytest.mark.parametrize("c", cases("bl_subtract"), ids=lambda c: c.name)
def test_golden_bl_subtract(c):
    out, rc = oracle.bl_subtract(c["w_in"], c["baseline"])
    _check_fatal(c, rc)
    _eq(out[0], c["w_out"], c.name)


@pytest.mark.parametrize("c", cases("pole_zero"), ids=lambda c: c.name)
def test_golden_pole_zero(c):
    out, rc = oracle.pole_zero(c["w_in"], c.params["tau"])
    _check_fatal(c, rc)
    _eq_or_libm(c, out[0], c["w_out"])


@pytest.mark.parametrize("c", cases("double_pole_zero"), ids=lambda c: c.name)
def test_golden_double_pole_zero(c):
    p = c.params
    out, rc = oracle.double_pole_zero(c["w_in"], p["tau1"], p["tau2"], p["frac"])
    _check_fatal(c, rc)
    _eq_or_libm(c, out[0], c["w_out"])


@pytest.mark.parametrize("c", cases("trap_filters"), ids=lambda c: c.name)
def test_golden_traps(c):
    p = c.params
    args = [p["rise"], p["flat"]] + ([p["fall"]] if c.kernel == "asym_trap_filter" else [])
    out, rc = getattr(oracle, c.kernel)(c["w_in"], *args)
    if zerodiv(c) and not c.fatal:
        assert oracle.E_NAMES[rc] == "ZERODIV"
        return
    _check_fatal(c, rc)
    _eq(out[0], c["w_out"], c.name)


@pytest.mark.parametrize("c", cases("fixed_time_pickoff"), ids=lambda c: c.name)
def test_golden_fixed_time_pickoff(c):
    out, rc = oracle.fixed_time_pickoff(c["w_in"], c.params["t_in"], c.params["mode"])
    _check_fatal(c, rc)
    want = c["a_out"]
    if c.params["mode"] in "hs" and c.tag == "f64":
        # the golden body evaluates x**3 through libm pow, numba through x*(x*x): last-bit differences in float64
        assert np.isclose(out[0], want, rtol=1e-13, atol=0, equal_nan=True), c.name
    else:
        _eq(out[0], want, c.name)


@pytest.mark.parametrize("c", cases("time_point_thresh"), ids=lambda c: c.name)
def test_golden_time_point_thresh(c):
    p = c.params
    out, rc = oracle.time_point_thresh(c["w_in"], p["a_threshold"], p["t_start"], p["walk_forward"])
    _check_fatal(c, rc)
    _eq(out[0], c["t_out"], c.name)


@pytest.mark.parametrize("c", cases("interpolated_time_point_thresh"), ids=lambda c: c.name)
def test_golden_interpolated_time_point_thresh(c):
    p = c.params
    out, rc = oracle.interpolated_time_point_thresh(c["w_in"], p["a_threshold"], p["t_start"], p["walk_forward"], p["mode"])
    _check_fatal(c, rc)
    _eq(out[0], c["t_out"], c.name)


@pytest.mark.parametrize("c", cases("min_max_norm"), ids=lambda c: c.name)
def test_golden_min_max_norm(c):
    out, rc = oracle.min_max_norm(c["w_in"], c.params["a_min"], c.params["a_max"])
    _check_fatal(c, rc)
    _eq(out[0], c["w_out"], c.name)


def test_known_min_max_norm():
    """reference tests/processors/test_min_max_norm.py:6-41"""
    w = np.ones(10)
    wn = w.copy()
    wn[4] = np.nan
    assert np.isnan(oracle.min_max_norm(wn, 1, 1)[0]).all()
    assert np.allclose(oracle.min_max_norm(w, 0, 0)[0], np.ones(10))
    assert np.allclose(oracle.min_max_norm(w, -1, 2)[0], np.ones(10) / 2) and np.allclose(oracle.min_max_norm(w, -2, 1)[0], np.ones(10) / 2)


@pytest.mark.parametrize("c", cases("min_max"), ids=lambda c: c.name)
def test_golden_min_max(c):
    *o, rc = oracle.min_max(c["w_in"])
    _check_fatal(c, rc)
    _eq(np.array([x[0] for x in o]), c["out"], c.name)


@pytest.mark.parametrize("c", cases("windows"), ids=lambda c: c.name)
def test_golden_windows(c):
    p = c.params
    if c.kernel == "windower":
        out, rc = oracle.windower(c["w_in"], p["t0_in"], c["w_out"].shape[-1])
        want = c["w_out"]
    elif c.kernel == "avg_current":
        out, rc = oracle.avg_current(c["w_in"], p["length"], c["w_out"].shape[-1])
        want = c["w_out"]
    else:
        out, rc = oracle.trap_pickoff(c["w_in"], p["rise"], p["flat"], p["t_pickoff"])
        want = c["a_out"]
    _check_fatal(c, rc)
    _eq(out[0], want, c.name)


@pytest.mark.parametrize("c", cases("linear_slope_fit"), ids=lambda c: c.name)
def test_golden_linear_slope_fit_emulated_typing(c):
    """PARITY UNPINNED: these fixtures restate numba's typing in Python (oracle/gen_golden.py), they are not reference output"""
    *o, rc = oracle.linear_slope_fit(c["w_in"])
    assert rc == 0
    _eq(np.array([x[0] for x in o]), c["out"], c.name)
    # the plain NumPy-2 execution of the reference body differs only in how the mean / variance updates round
    if not np.isnan(c["out"]).any() and c["w_in"].size > 10:
        assert np.allclose(c["out"], c["numpy2_out"], rtol=5e-5, atol=1e-3)


@pytest.mark.parametrize("c", cases("current"), ids=lambda c: c.name)
def test_golden_current_branch(c):
    p = c.params
    if c.kernel == "upsampler":
        out, rc = oracle.upsampler(c["w_in"], p["upsample"], c["w_out"].shape[-1])
    else:
        out, rc = oracle.moving_window_multi(c["w_in"], p["length"], p["num_mw"], p["mw_type"])
    _check_fatal(c, rc)
    _eq(out[0], c["w_out"], c.name)


@pytest.mark.parametrize("c", cases("arithmetic"), ids=lambda c: c.name)
def test_golden_mean_below_threshold(c):
    out, rc = oracle.mean_below_threshold(c["w_in"], c.params["threshold"])
    _check_fatal(c, rc)
    _eq(out[0], c["result"], c.name)


def test_reference_known_answers_mean_below_threshold():
    """reference tests/processors/test_arithmetic.py:8-40"""
    w = np.array([1.0, 2.0, 3.0, 4.0, 5.0])
    assert oracle.mean_below_threshold(w, 4.0)[0][0] == 2.0
    assert oracle.mean_below_threshold(w, 100.0)[0][0] == 3.0
    assert np.isnan(oracle.mean_below_threshold(np.array([10.0, 20.0, 30.0, 40.0, 50.0]), 10.0)[0][0])
    assert np.isnan(oracle.mean_below_threshold(np.array([1.0, 2.0, np.nan, 4.0, 5.0]), 4.0)[0][0])


@pytest.mark.parametrize("c", cases("convolutions"), ids=lambda c: c.name)
def test_golden_convolve(c):
    w, k, want = c["w_in"], c["kernel"], c["w_out"]
    out, rc = oracle.convolve_wf(w, k, c.params["mode"], want.shape[-1], in_len=c.params.get("slice_stop"))
    _check_fatal(c, rc)
    # np.convolve / scipy fftconvolve float32 summation order is library-internal
    tol = 1e-6 if c.tag == "f32" else 1e-12
    assert_rel_to_peak(out, want, tol, c.name)


@pytest.mark.parametrize("c", cases("dwt"), ids=lambda c: c.name)
def test_golden_dwt(c):
    want = c["w_out"]
    out, rc = oracle.dwt_haar(c["w_in"], c.params["level"], c.params["coeff"], len(want))
    assert rc == 0
    _eq(out[0], want, c.name)


def test_golden_chains():
    (c1, c2, c5) = cases("chains")
    out, rc = oracle.chain_pz_trap(c1["waveform"], c1.params["tau"], c1.params["rise"], c1.params["flat"])
    assert rc == 0
    _eq(out, c1["wf_trap"], "c1")
    p = c2.params
    for threads in (1, 4):
        e, rc = oracle.chain_energy(c2["waveform"], c2["baseline"], c2["t_pick"], p["tau"], p["rise"], p["flat"], p["mode"],
                                    block_width=5, n_threads=threads)
        assert rc == 0
        _eq(e, c2["trapEftp"], "c2")
    assert np.isnan(c2["trapEftp"][5]) and np.isnan(c2["trapEftp"][7])
    # C5: int16 -> float32 loop (ProcessorManager type matching, processing_chain.py:1565-1572)
    p = c5.params
    w = c5["waveform"].astype(np.float32)
    dpz, rc = oracle.double_pole_zero(w, p["tau1"], p["tau2"], p["frac"])
    assert rc == 0
    _eq(dpz[:1], c5["wf_pz"], "c5 dpz")
    at, rc = oracle.asym_trap_filter(dpz, p["rise"], p["flat"], p["fall"])
    _eq(at[:1], c5["wf_atrap"], "c5 atrap")
    tmin, tmax, amin, amax, rc = oracle.min_max(at)
    _eq(np.stack([tmin, tmax, amin, amax], axis=1), c5["min_max"], "c5 minmax")
    tp0, rc = oracle.time_point_thresh(at, c5["thr"], tmax, 0)
    assert rc == 0
    _eq(tp0, c5["tp_0"], "c5 tp0")
