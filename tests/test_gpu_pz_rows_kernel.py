"""[bl_subtract ->] pole_zero of whole rows written back as rows (dsp_pz.hip): the recurrence as a prefix sum in float64, walked in memory
order by one wavefront per row.  Against the oracle (reference processors/pole_zero.py:24-77) to the filter bar, against the waveform VM's own
scan formulation to a last place of float32, and the reference's NaN rule and DSPFatal."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
M = "dspeed.processors"
TOL = 1e-6


def _rows(rng, n, L, dtype):
    i = np.arange(L)[None, :]
    t0 = np.floor(rng.uniform(0.3, 0.6, (n, 1)) * L)
    x = rng.uniform(800, 3000, (n, 1)) + rng.uniform(500, 15000, (n, 1)) * np.exp(-np.clip(i - t0, 0, None) / 1716.28) * (i >= t0)
    x += 5 * rng.standard_normal((n, L))
    return (np.rint(x) if np.dtype(dtype).kind in "iu" else x).astype(dtype)


def _run(rec, tb, fused):
    from dspeed_amd.processing_chain import build_processing_chain

    chain, _, out = build_processing_chain(rec, tb)
    chain._ensure()
    assert chain._chain.set_fused(1 if fused else 0) == bool(fused)
    chain.execute()
    return chain, out


@pytest.mark.parametrize("dtype,L,use_bl", [(np.uint16, 8192, True), (np.int16, 8192, True), (np.float32, 8192, True), (np.float32, 4096, False),
                                             (np.uint16, 1000, True), (np.float32, 8, False), (np.int16, 520, False)])
def test_pole_zero_rows(dtype, L, use_bl):
    rng = np.random.default_rng(L + int(use_bl))
    n = 203
    wf = _rows(rng, n, L, dtype)
    bl = rng.uniform(800, 3000, n).astype(np.float32)
    procs = {"wf_pz": f"{M}.pole_zero(wf_bl, 1716.28, wf_pz)", "wf_bl": f"{M}.bl_subtract(waveform, baseline, wf_bl)"} if use_bl else \
        {"wf_pz": f"{M}.pole_zero(waveform, 1716.28, wf_pz)"}
    rec = {"outputs": ["wf_pz"], "processors": procs}
    tb = {"waveform": wf, "baseline": bl}
    chain, out = _run(rec, tb, True)
    assert chain._chain.kernel_name == "dsp_pz_rows_kernel"
    x = wf.astype(np.float32)
    if use_bl:
        x, rc = oracle.bl_subtract(x, bl)
        assert rc == 0
    want, rc = oracle.pole_zero(x, 1716.28)
    assert rc == 0
    peak = np.abs(want).max(axis=1, keepdims=True)
    assert out["wf_pz"].shape == want.shape and np.max(np.abs(out["wf_pz"] - want) / peak) <= TOL
    _, vm = _run(rec, tb, False)
    ulp = np.spacing(np.abs(vm["wf_pz"]).astype(np.float32))
    assert np.max(np.abs(out["wf_pz"] - vm["wf_pz"]) / np.maximum(ulp, np.float32(1e-30))) <= 2  # (another order of the same float64 sums)
    assert np.mean(out["wf_pz"] == vm["wf_pz"]) > 0.98


@pytest.mark.parametrize("dtype,L,use_bl", [(np.uint16, 8192, True), (np.int16, 4096, True), (np.float32, 8192, True), (np.float32, 520, False)])
def test_min_max_of_the_raw_rows_goes_along(dtype, L, use_bl):
    """every Ge recipe asks for tp_min / tp_max / wf_min / wf_max of the raw waveform (min_max.py:11-82): the kernel that writes the pole-zero
    rows streams the raw rows anyway and keeps their first-occurrence extremes -- comparisons only: the oracle's values bit for bit (ties,
    constant rows, a NaN -> four NaNs, infinities, -0.0), and the interpreter's on the same program"""
    rng = np.random.default_rng(L + 7)
    n = 333
    wf = _rows(rng, n, L, dtype)
    wf[1, :] = wf[1, 0]                                   # a constant row: both extremes at sample 0
    wf[2, [L // 3, L - 1]] = wf[2].max()                  # the maximum twice: the first one counts
    wf[3, [0, L // 2]] = wf[3].min()
    wf[4, 9:] = wf[4, 9]                                  # an extreme inside one lane's eight samples, then nothing new
    if np.dtype(dtype) == np.float32:
        wf[5, L // 2] = np.nan                            # -> four NaNs (and a NaN waveform)
        if L % 1024 == 0:
            wf[6, L - 1] = np.inf                         # (the last sample: the recurrence ends on an infinity, not on inf - inf)
        wf[8, :] = 0.0
        wf[8, L // 4] = -0.0                              # -0.0 < 0.0 is false: sample 0 stays the minimum
    bl = rng.uniform(800, 3000, n).astype(np.float32)
    procs = {"wf_bl": f"{M}.bl_subtract(waveform, baseline, wf_bl)", "wf_pz": f"{M}.pole_zero(wf_bl, 1716.28, wf_pz)"} if use_bl else \
        {"wf_pz": f"{M}.pole_zero(waveform, 1716.28, wf_pz)"}
    procs["t_lo, t_hi, v_lo, v_hi"] = f"{M}.min_max(waveform, t_lo, t_hi, v_lo, v_hi)"
    rec = {"outputs": ["wf_pz", "t_lo", "t_hi", "v_lo", "v_hi"], "processors": procs}
    tb = {"waveform": wf, "baseline": bl}
    chain, out = _run(rec, tb, True)
    assert [k for _w, k in chain.kernels()] == ["dsp_pz_rows_kernel"]  # (the whole program: one launch)
    want = oracle.min_max(np.ascontiguousarray(wf, dtype=np.float32))
    for k, w in zip(("t_lo", "t_hi", "v_lo", "v_hi"), want[:4]):
        np.testing.assert_array_equal(out[k], w, err_msg=k)
    _, vm = _run(rec, tb, False)
    for k in ("t_lo", "t_hi", "v_lo", "v_hi"):
        np.testing.assert_array_equal(out[k], vm[k], err_msg=k)
    ok = ~np.isnan(vm["wf_pz"]).any(axis=1) & np.isfinite(vm["wf_pz"]).all(axis=1)
    ulp = np.spacing(np.abs(vm["wf_pz"][ok]).astype(np.float32))
    assert np.max(np.abs(out["wf_pz"][ok] - vm["wf_pz"][ok]) / np.maximum(ulp, np.float32(1e-30))) <= 2


def test_nan_rule_and_dspfatal():
    from dspeed_amd.errors import DSPFatal

    rng = np.random.default_rng(3)
    wf = _rows(rng, 40, 2048, np.float32)
    wf[3, 700] = np.nan            # a NaN anywhere: the whole waveform NaN (pole_zero.py:55-58)
    bl = np.full(40, 1000.0, dtype=np.float32)
    bl[5] = np.nan                 # ... also through the baseline
    rec = {"outputs": ["wf_pz"], "processors": {"wf_bl": f"{M}.bl_subtract(waveform, baseline, wf_bl)", "wf_pz": f"{M}.pole_zero(wf_bl, 1716.28, wf_pz)"}}
    chain, out = _run(rec, {"waveform": wf, "baseline": bl}, True)
    assert chain._chain.kernel_name == "dsp_pz_rows_kernel"
    assert np.isnan(out["wf_pz"][[3, 5]]).all() and not np.isnan(np.delete(out["wf_pz"], [3, 5], axis=0)).any()
    wf2 = wf.copy()
    wf2[3, 700] = 0.0
    wf2[7, 100] = np.inf           # inf - inf inside the recurrence: a NaN of its own making -> DSPFatal with the row (:76-77)
    bl[5] = 1000.0
    with pytest.raises(DSPFatal) as e:
        _run(rec, {"waveform": wf2, "baseline": bl}, True)
    assert e.value.wf_range is not None and 7 in e.value.wf_range


def test_row_scales_travel_with_the_rows(monkeypatch):
    """a float16 FIR behind the pole-zero rows takes the rows' scales and flags from the kernel that wrote them (dsp_chain_share_row_scales):
    bit for bit what it finds when it reads the rows itself -- ordinary rows, an all-NaN row, all-zero rows, tiny and huge ones"""
    from dspeed_amd.processing_chain import build_processing_chain

    rng = np.random.default_rng(11)
    n, L = 300, 4096
    wf = _rows(rng, n, L, np.float32)
    wf[4, 100] = np.nan
    wf[9] = 0.0
    wf[10] *= np.float32(1e-30)
    wf[11] *= np.float32(1e30)
    wf[12] *= np.float32(1e34)     # (the scale's exponent beyond +-100: the FIR's slow path)
    bl = np.zeros(n, dtype=np.float32)
    rec = {"outputs": ["wf_f", "f_max"], "processors": {
        "wf_pz": f"{M}.pole_zero(waveform, 1716.28, wf_pz)",
        "kern": {"function": "t0_filter", "module": M, "args": ["8", "125", "kern(133, 'f')"]},
        "wf_f": {"function": "convolve_wf", "module": M, "args": ["wf_pz", "kern", "'s'", "wf_f(4096, 'f')"]},
        "f_max": "numpy.amax(wf_f, 1, f_max)"}}

    monkeypatch.setenv("DSPEED_HIP_NO_FIR_RUNS", "1")  # (this kernel is piecewise constant: by default the run-length FIR, which needs no scales)

    def run(shared):
        monkeypatch.setenv("DSPEED_HIP_NO_SHARED_ROW_SCALES", "0" if shared else "1")
        chain, _, out = build_processing_chain(rec, {"waveform": wf, "baseline": bl})
        chain.execute()
        kinds = [k for _w, k in chain.kernels()]
        assert "dsp_pz_rows_kernel" in kinds and "dsp_fir_f16_kernel" in kinds, kinds
        return {k: np.array(v) for k, v in out.items()}

    a, b = run(True), run(False)
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert np.isnan(a["wf_f"][4]).all() and not np.isnan(a["wf_f"][[0, 9, 10, 11]]).any()
    import dspeed_amd.processors as P

    kern = np.zeros(133, dtype=np.float32)
    P.t0_filter(8, 125, kern)
    want = oracle.convolve_wf(oracle.pole_zero(np.delete(wf, [4, 12], axis=0), 1716.28)[0], kern, "s", 4096)[0]
    got = np.delete(a["wf_f"], [4, 12], axis=0)
    peak = np.abs(want).max(axis=1, keepdims=True)
    ok = peak[:, 0] > 1e-38
    assert np.max(np.abs(got - want)[ok] / peak[ok]) <= 2e-6


@pytest.mark.parametrize("dtype", [np.int16, np.float32])
def test_time_constant_per_event(dtype):
    """tau as a per-event column: exp(-1/tau) per row in float64 on the device (pole_zero.py:60), a NaN time constant -> a NaN waveform"""
    rng = np.random.default_rng(17)
    n, L = 90, 2048
    wf = _rows(rng, n, L, dtype)
    tau = rng.uniform(500, 3000, n).astype(np.float32)
    tau[11] = np.nan
    rec = {"outputs": ["wf_pz"], "processors": {"wf_pz": f"{M}.pole_zero(waveform, tau, wf_pz)"}}
    chain, out = _run(rec, {"waveform": wf, "tau": tau}, True)
    assert chain._chain.kernel_name == "dsp_pz_rows_kernel"
    assert np.isnan(out["wf_pz"][11]).all()
    x = wf.astype(np.float32)
    for r in range(n):
        if r == 11:
            continue
        want = oracle.pole_zero(x[r:r + 1], float(tau[r]))[0][0]
        assert np.max(np.abs(out["wf_pz"][r] - want)) / np.abs(want).max() <= TOL, r


def test_row_scales_follow_the_batch(monkeypatch):
    """the same chain on a second, larger batch (the scale arrays grow, the rows live elsewhere) and on a batch handed over in pieces
    (execute(begin, end): every piece is an execute of its own, the producer's note is good for exactly one): as without the link"""
    from dspeed_amd.processing_chain import build_processing_chain

    rng = np.random.default_rng(23)
    L = 2048
    rec = {"outputs": ["wf_f"], "processors": {
        "wf_pz": f"{M}.pole_zero(waveform, 1716.28, wf_pz)",
        "kern": {"function": "t0_filter", "module": M, "args": ["8", "125", "kern(133, 'f')"]},
        "wf_f": {"function": "convolve_wf", "module": M, "args": ["wf_pz", "kern", "'s'", f"wf_f({L}, 'f')"]}}}
    batches = [_rows(rng, 100, L, np.int16), _rows(rng, 700, L, np.int16), _rows(rng, 64, L, np.int16)]

    monkeypatch.setenv("DSPEED_HIP_NO_FIR_RUNS", "1")  # (this kernel is piecewise constant: by default the run-length FIR, which needs no scales)

    def run(shared):
        monkeypatch.setenv("DSPEED_HIP_NO_SHARED_ROW_SCALES", "0" if shared else "1")
        chain, _, _ = build_processing_chain(rec, {"waveform": batches[0]})
        res = []
        for wf in batches:
            out = {"wf_f": np.full((len(wf), L), -1.0, dtype=np.float32)}
            chain.link({"waveform": wf}, out)
            chain.execute()
            res.append(out["wf_f"].copy())
            out2 = {"wf_f": np.full((len(wf), L), -1.0, dtype=np.float32)}
            chain.link({"waveform": wf}, out2)
            half = len(wf) // 2
            chain.execute(0, half)
            chain.execute(half, len(wf))
            assert np.array_equal(out2["wf_f"], out["wf_f"])
        return res

    for a, b in zip(run(True), run(False)):
        assert np.array_equal(a, b) and not np.isnan(a).any() and np.abs(a).max() > 0


def test_many_rows_scale_exactly():
    """32 768 rows of 8192 int16 samples (a production piece): pole_zero(2 x) == 2 pole_zero(x) bit for bit (a power of two commutes with every
    sum, product and rounding of the recurrence), and every 1024th row against the oracle"""
    rng = np.random.default_rng(29)
    n, L = 32768, 8192
    wf = _rows(rng, 64, L, np.int16)
    wf = np.clip(np.tile(wf, (n // 64, 1)) + rng.integers(-40, 40, (n, 1), dtype=np.int16), -16000, 16000).astype(np.int16)
    bl = rng.uniform(800, 3000, n).astype(np.float32)
    rec = {"outputs": ["wf_pz"], "processors": {"wf_bl": f"{M}.bl_subtract(waveform, baseline, wf_bl)", "wf_pz": f"{M}.pole_zero(wf_bl, 1716.28, wf_pz)"}}
    chain, a = _run(rec, {"waveform": wf, "baseline": bl}, True)
    assert chain._chain.kernel_name == "dsp_pz_rows_kernel"
    a = np.array(a["wf_pz"])
    _, b = _run(rec, {"waveform": (2 * wf).astype(np.int16), "baseline": (2 * bl).astype(np.float32)}, True)
    assert np.array_equal(np.array(b["wf_pz"]), 2 * a)
    pick = np.arange(0, n, 1024)
    want = oracle.pole_zero(oracle.bl_subtract(wf[pick].astype(np.float32), bl[pick])[0], 1716.28)[0]
    assert np.max(np.abs(a[pick] - want) / np.abs(want).max(axis=1, keepdims=True)) <= TOL
