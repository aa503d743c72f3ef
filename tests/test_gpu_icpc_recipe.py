"""The whole Ge recipe (structure of the reference's tests/configs/icpc-dsp-config.json) as ONE device program, against the oracle run
processor by processor with the reference's unit handling restated in NumPy (processing_chain.py:1556-1732 grid of a processor,
:1806-1908 + unit_conversion.py:16-21 coordinate conversion, :832-891 expressions as ufunc processors, :1193-1266 round onto a grid,
:1990-2014 time coordinates written in their unit)."""
import numpy as np
import pytest

import golden_util
import oracle
import recipes

pytestmark = pytest.mark.gpu
F = np.float32


def _synth(rng, n_wf, wf_len=8192):
    i = np.arange(wf_len, dtype=np.float64)[None, :]
    B = rng.uniform(9000, 11000, (n_wf, 1))
    A = rng.uniform(2000, 15000, (n_wf, 1))
    t0 = np.floor(rng.uniform(0.45, 0.55, (n_wf, 1)) * wf_len)
    rise = 1.0 / (1.0 + np.exp(np.clip(-(i - t0) / 6.0, -60, 60)))  # ~100 ns charge collection: the current pulse has a width
    x = B + A * rise * np.exp(-np.maximum(i - t0, 0) / 1716.25) + 5.0 * rng.standard_normal((n_wf, wf_len))
    return np.rint(x).astype(np.uint16), B[:, 0].astype(F)


def _convert(x, off_in, off_out, ratio, rounding=None):
    """unit_conversion.py:16-21: float64 arithmetic, the variable's type back"""
    r = (x.astype(np.float64) + off_in) * ratio - off_out
    if rounding is not None:
        r = rounding(r)
    return r.astype(x.dtype)


def _expected(wf, bl, t0_ns, F=np.float32, dt=16.0, par=recipes.ICPC_PARAMS):
    """the oracle's processors composed as the recipe says, with the reference's unit handling restated; ``par``: the recipe's parameter values
    (recipes.ICPC_PARAMS / ICPC_REF_PARAMS).  The kernels are the reference generators' own (fixtures: golden_util.recipe_kernel)."""
    e = {}
    w = wf.astype(F)
    bl, t0_ns = bl.astype(F), t0_ns.astype(F)
    off = _convert(t0_ns, 0.0, 0.0, 1.0 / dt)  # the grid offset in samples, in t0's type (reference :126-136)
    to_ns = lambda t: _convert(t, off.astype(np.float64), 0.0, dt)  # noqa: E731   (index + offset) * period

    tmin, tmax, e["wf_min"], e["wf_max"], _ = oracle.min_max(w)
    e["tp_min"], e["tp_max"] = to_ns(tmin), to_ns(tmax)
    blsub = oracle.bl_subtract(w, bl)[0]
    e["bl_mean"], e["bl_std"], e["bl_slope"], e["bl_intercept"], _ = oracle.linear_slope_fit(np.ascontiguousarray(blsub[:, :par["bl_window"]]))
    pz = oracle.pole_zero(blsub, F(par["tau_samples"]))[0]
    e["pz_mean"], e["pz_std"], e["pz_slope"], _, _ = oracle.linear_slope_fit(np.ascontiguousarray(pz[:, par["pz_from"]:]))
    k0 = golden_util.recipe_kernel("t0")  # (declared 'f': float32 taps in either loop)
    wt0 = oracle.convolve_wf(pz, k0, "s", 8192)[0]
    _, tp_start, _, _, _ = oracle.min_max(wt0)
    atrap = oracle.asym_trap_filter(pz, 8, 4, 125)[0]
    tp_atrap = oracle.time_point_thresh(atrap, e["bl_std"], tp_start, 0)[0]
    tp0 = oracle.time_point_thresh(wt0, e["bl_std"], tp_start, 0)[0]
    e["tp_0_atrap"], e["tp_0_est"] = to_ns(tp_atrap), to_ns(tp0)
    trap = oracle.trap_norm(pz, 625, 188)[0]
    e["trapTmax"] = np.max(trap, axis=1)
    etrap = oracle.trap_norm(pz, *par["etrap"])[0]
    e["trapEmax"] = np.max(etrap, axis=1)
    # round(tp_0_est + rise + flat*0.8, wf_etrap.grid): two float32 additions, then rint on the same grid
    t_pick = _convert((tp0 + F(par["pick_ns"][0] / dt)) + F(par["pick_ns"][1] / dt), off.astype(np.float64), off.astype(np.float64), 1.0, np.rint)
    e["trapEftp"] = oracle.fixed_time_pickoff(etrap, t_pick, "l")[0]
    cusp = oracle.convolve_wf(blsub, golden_util.recipe_kernel("cusp"), "v", 301, in_len=8192 - 2100)[0]
    e["cuspEmax"] = np.max(cusp, axis=1)
    e["cuspEftp"] = oracle.fixed_time_pickoff(cusp, F(50), "i")[0]
    e["_peak:cuspEftp"] = np.max(np.abs(cusp), axis=1)  # (what a sample of the filtered waveform is measured against: the waveform's peak)
    if par["zac"]:
        zac = oracle.convolve_wf(blsub, golden_util.recipe_kernel("zac"), "v", 301, in_len=8192 - 2100)[0]
        e["zacEmax"] = np.max(zac, axis=1)
        e["zacEftp"] = oracle.fixed_time_pickoff(zac, F(50), "i")[0]
        e["_peak:zacEftp"] = np.max(np.abs(zac), axis=1)  # (the zero-area kernel on rows that still decay: the undershoot is several times zacEmax)
    tmx = e["trapTmax"]
    walks = {"tp_100": oracle.time_point_thresh(pz, tmx, tp0, 1)[0], "tp_99": oracle.time_point_thresh(pz, F(0.99) * tmx, tp0, 1)[0]}
    for name, frac, start in par["ladder"]:  # each rung walks backward from the one above it
        walks[name] = oracle.time_point_thresh(pz, tmx * F(frac), walks[start], 0)[0]
    for k, v in walks.items():
        e[k] = to_ns(v)
    trap2 = oracle.trap_norm(pz, 250, 6)[0]
    q = oracle.fixed_time_pickoff(trap2, tp0 + F(8096.0 / dt), "l")[0]
    e["QDrift"] = q * F(16)
    e["dt_eff"] = e["QDrift"] / tmx
    le = oracle.windower(pz, tp0, 301)[0]
    cur = oracle.avg_current(le, 1)[0]
    up = oracle.upsampler(cur, 16, 4784)[0]
    av = oracle.moving_window_multi(up, 48, 3, 0)[0]
    _, ta, _, e["A_max"], _ = oracle.min_max(av)
    e["tp_aoe_max"] = ta  # no grid on the windowed waveform: stays an index of the upsampled current, whatever its "unit" says
    e["tp_aoe_samp"] = to_ns(tp0 + ta / F(16))
    return e, tp0


@pytest.mark.parametrize("t0_kind,rows_dtype", [("per_row", np.uint16), ("constant", np.uint16), ("per_row", np.int32)])
def test_whole_ge_recipe_is_one_device_program(t0_kind, rows_dtype):
    """uint16 rows run the float32 loop, int32 rows the float64 loop (first castable signature, processing_chain.py:1565-1572)"""
    from dspeed_amd.processing_chain import WaveformInput, build_processing_chain

    rng = np.random.default_rng(2026)
    n = 48
    wf, bl = _synth(rng, n)
    wf = wf.astype(rows_dtype)
    ft = np.float64 if rows_dtype == np.int32 else np.float32
    t0_ns = (rng.integers(2900, 3100, n) * 16).astype(F) if t0_kind == "per_row" else np.full(n, 48000.0, dtype=F)
    tb = {"waveform": WaveformInput(wf, 16.0, t0_ns if t0_kind == "per_row" else 48000.0), "baseline": bl}
    chain, mask, out = build_processing_chain(recipes.ICPC, tb)
    assert sorted(mask) == ["baseline", "waveform"] and chain.loop_dtype == ft
    chain.execute()
    want, tp0 = _expected(wf, bl, t0_ns, ft)
    assert set(out) == set(recipes.ICPC["outputs"]) and all(v.dtype == ft for v in out.values())
    want = {k: v for k, v in want.items() if not k.startswith("_")}

    # Measured on this batch (tools/icpc_parity_measure.py, profiles/r03_icpc_parity_by_output.json): every index / time output, the six fit
    # outputs (the fits run on the rows exactly as the oracle's loops do) and the current branch (dsp_current.hip) equal the all-oracle run
    # bit for bit in 48 of 48 rows, in the float32 and in the float64 loop; the energies differ by what the trapezoid replay and the FIR's
    # summation order leave: <= 3e-7 of the value in float32, <= 4e-14 in float64.  The assertions are those numbers, not a blanket bound.
    exact = ["tp_min", "tp_max", "wf_min", "wf_max", "tp_0_est", "tp_0_atrap", "tp_10", "tp_50", "tp_90", "tp_99", "tp_100", "tp_aoe_max",
             "tp_aoe_samp", "bl_mean", "bl_std", "bl_slope", "bl_intercept", "pz_mean", "pz_std"]
    if ft == np.float32:
        exact.append("A_max")
    tol = 1e-6 if ft == np.float32 else 1e-12
    rel = {"trapTmax": tol, "trapEmax": tol, "trapEftp": tol, "cuspEmax": tol, "cuspEftp": tol, "QDrift": tol, "dt_eff": tol, "A_max": tol}
    for k in exact:
        assert np.array_equal(out[k], want[k], equal_nan=True), (k, int(np.sum(out[k] != want[k])))
    for k, bound in rel.items():
        scale = np.maximum(np.abs(want[k]), 1e-3 * np.max(np.abs(want[k])))
        assert not np.isnan(out[k]).any(), k
        err = np.max(np.abs(out[k] - want[k]) / scale)
        assert err <= bound, f"{k}: {err:.3g}"
    # the times are in ns with the waveform's t0 in them: the rise sits ~ t0 + 0.5 * 8192 * 16 ns
    assert np.all(np.abs(out["tp_0_est"] - (t0_ns + 0.5 * 8192 * 16)) < 0.08 * 8192 * 16)


def test_the_references_parameter_values_all_34_outputs():
    """The recipe with the parameter values and the full output list of the reference's own Ge test configuration
    (tests/configs/icpc-dsp-config.json:1-347; recipes.ICPC_REF, shown op for op equal to that file's translation by
    tests/test_recipe_language_cpu.py) on the device against the all-oracle run: 34 outputs."""
    from dspeed_amd.processing_chain import WaveformInput, build_processing_chain

    rng = np.random.default_rng(2031)
    n = 64
    wf, bl = _synth(rng, n)
    t0_ns = (rng.integers(2900, 3100, n) * 16).astype(F)
    tb = {"waveform": WaveformInput(wf, 16.0, t0_ns), "baseline": bl}
    chain, mask, out = build_processing_chain(recipes.ICPC_REF, tb)
    assert len(out) == 34 and set(out) == set(recipes.ICPC_REF["outputs"])
    chain.execute()
    want, _tp0 = _expected(wf, bl, t0_ns, np.float32, 16.0, recipes.ICPC_REF_PARAMS)
    # index / time / fit / current-branch outputs: the oracle's bit for bit wherever the filtered samples a walk compares do not differ
    # (measured: all rows of this batch); energies to the filter bar
    exact = ["tp_min", "tp_max", "wf_min", "wf_max", "tp_0_est", "tp_0_atrap", "tp_01", "tp_10", "tp_20", "tp_50", "tp_80", "tp_90", "tp_95", "tp_99",
             "tp_100", "tp_aoe_max", "tp_aoe_samp", "bl_mean", "bl_std", "bl_slope", "bl_intercept", "pz_mean", "pz_std", "pz_slope", "A_max"]
    rel = ["trapTmax", "trapEmax", "trapEftp", "cuspEmax", "cuspEftp", "zacEmax", "zacEftp", "QDrift", "dt_eff"]
    assert sorted(exact + rel) == sorted(recipes.ICPC_REF["outputs"])
    for k in exact:
        assert np.array_equal(out[k], want[k], equal_nan=True), (k, int(np.sum(out[k] != want[k])))
    # A maximum is held to 1e-6 of its own value.  A sample picked off a filtered waveform at a fixed time (cuspEftp, zacEftp: sample 50 of 301,
    # on the filter's flank) is a filter OUTPUT SAMPLE: the bar is the waveform's, 1e-6 of the filtered waveform's peak (north_star; against
    # float64 the device's zero-area filter is within 3.8e-7 of the peak at every sample, the oracle within 6e-8: profiles/r04_fir_zac_samples.json)
    measured = {}
    for k in rel:
        scale = want[f"_peak:{k}"] if f"_peak:{k}" in want else np.maximum(np.abs(want[k]), 1e-3 * np.max(np.abs(want[k])))
        assert not np.isnan(out[k]).any(), k
        err = np.max(np.abs(out[k] - want[k]) / scale)
        measured[k] = {"rel_to_bar_scale": float(err), "rel_to_own_value": float(np.max(np.abs(out[k] - want[k]) / np.abs(want[k])))}
        assert err <= 1e-6, f"{k}: {err:.3g}"
    import json
    import os

    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/icpc_ref_values_parity.json", "w") as f:
        json.dump({"recipe": "recipes.ICPC_REF (the reference file's values, 34 outputs)", "rows": n, "bit_exact_outputs": exact, "float_outputs": measured}, f, indent=1)
    assert (out["tp_01"] <= out["tp_10"]).all() and (out["tp_80"] <= out["tp_95"]).all() and not np.isnan(out["tp_01"]).any()


def test_index_outputs_are_bit_exact_on_the_devices_own_waveforms():
    """SURVEY H5 made checkable: every index / threshold / pick-off / extremum output of the recipe is recomputed by the ORACLE from the
    DEVICE's own intermediate waveforms and per-event values (requested as extra outputs, a few at a time: they all live in LDS) and must
    agree bit for bit in 100 % of the rows.  Whatever differs end to end can then only come from the <= 1e-6 differences of the filtered
    waveforms themselves, which are asserted beside it; and the production program (no extra outputs, fused ops) must print the same
    numbers as the instrumented ones."""
    from dspeed_amd.processing_chain import WaveformInput, build_processing_chain

    rng = np.random.default_rng(2027)
    n, dt = 96, 16.0
    wf, bl = _synth(rng, n)
    t0_ns = (rng.integers(2900, 3100, n) * 16).astype(F)
    tb = {"waveform": WaveformInput(wf, dt, t0_ns), "baseline": bl}
    off = _convert(t0_ns, 0.0, 0.0, 1.0 / dt)
    to_ns = lambda t: _convert(t, off.astype(np.float64), 0.0, dt)  # noqa: E731
    exact = lambda got, want, what: np.testing.assert_array_equal(got, want, err_msg=what)  # noqa: E731  (NaN == NaN; 100 % of the rows)

    def to_index(t_ns, what):
        """the sample index a device time stands for (the conversion is exact on the way back: checked)"""
        idx = np.rint(t_ns.astype(np.float64) / dt - off.astype(np.float64)).astype(F)
        idx[np.isnan(t_ns)] = np.nan
        exact(to_ns(idx), t_ns, f"{what}: not a sample of the waveform's grid")
        return idx

    def run(outs):
        chain, _, o = build_processing_chain(dict(recipes.ICPC, outputs=list(outs)), tb)
        chain.execute()
        return o

    chain_p, _, prod = build_processing_chain(recipes.ICPC, tb)
    chain_p.execute()
    seen = {}

    def same_as_production(o, names):
        for k in names:
            exact(o[k], prod[k], f"instrumented vs production program: {k}")
            seen[k] = True

    # ---- the t0 search: extremum of the t0-filtered waveform, threshold walks on it and on the asymmetric trapezoid
    g = run(["bl_std", "tp_0_est", "tp_0_atrap", "tp_start", "wf_t0_filter", "wf_atrap"])
    _, tp_start, _, _, _ = oracle.min_max(g["wf_t0_filter"])
    exact(g["tp_start"], to_ns(tp_start), "tp_start")
    exact(g["tp_0_atrap"], to_ns(oracle.time_point_thresh(g["wf_atrap"], g["bl_std"], tp_start, 0)[0]), "tp_0_atrap")
    exact(g["tp_0_est"], to_ns(oracle.time_point_thresh(g["wf_t0_filter"], g["bl_std"], tp_start, 0)[0]), "tp_0_est")
    same_as_production(g, ["bl_std", "tp_0_est", "tp_0_atrap"])
    wt0_dev, atrap_dev = g["wf_t0_filter"], g["wf_atrap"]
    # ---- rise-time points on the pole-zero corrected waveform, thresholds from the trapezoid's maximum
    g = run(["trapTmax", "tp_0_est", "tp_100", "tp_99", "tp_90", "tp_50", "tp_10", "wf_pz", "wf_trap"])
    pz, tmx = g["wf_pz"], g["trapTmax"]
    tp0 = to_index(g["tp_0_est"], "tp_0_est")
    exact(tmx, np.max(g["wf_trap"], axis=1), "trapTmax")
    t100 = oracle.time_point_thresh(pz, tmx, tp0, 1)[0]
    t99 = oracle.time_point_thresh(pz, F(0.99) * tmx, tp0, 1)[0]
    t90 = oracle.time_point_thresh(pz, tmx * F(0.9), t99, 0)[0]
    t50 = oracle.time_point_thresh(pz, tmx * F(0.5), t90, 0)[0]
    t10 = oracle.time_point_thresh(pz, tmx * F(0.1), t50, 0)[0]
    for k, v in (("tp_100", t100), ("tp_99", t99), ("tp_90", t90), ("tp_50", t50), ("tp_10", t10)):
        exact(g[k], to_ns(v), k)
    same_as_production(g, ["trapTmax", "tp_100", "tp_99", "tp_90", "tp_50", "tp_10"])
    pz_dev, trap_dev = pz, g["wf_trap"]
    # ---- energies: maxima and pick-offs; the time point is formed from the device's tp_0_est exactly as the recipe says
    g = run(["tp_0_est", "trapTmax", "trapEmax", "trapEftp", "trapQftp", "QDrift", "dt_eff", "wf_etrap", "wf_trap2"])
    tp0 = to_index(g["tp_0_est"], "tp_0_est")
    exact(g["trapEmax"], np.max(g["wf_etrap"], axis=1), "trapEmax")
    t_pick = _convert((tp0 + F(8000.0 / dt)) + F(2000.0 * 0.8 / dt), off.astype(np.float64), off.astype(np.float64), 1.0, np.rint)
    exact(g["trapEftp"], oracle.fixed_time_pickoff(g["wf_etrap"], t_pick, "l")[0], "trapEftp")
    q = oracle.fixed_time_pickoff(g["wf_trap2"], tp0 + F(8096.0 / dt), "l")[0]
    exact(g["trapQftp"], q, "trapQftp")
    exact(g["QDrift"], q * F(16), "QDrift")
    exact(g["dt_eff"], g["QDrift"] / g["trapTmax"], "dt_eff")
    same_as_production(g, ["trapEmax", "trapEftp", "QDrift", "dt_eff"])
    etrap_dev, trap2_dev = g["wf_etrap"], g["wf_trap2"]
    # ---- the cusp energy and the current branch
    g = run(["tp_0_est", "cuspEmax", "cuspEftp", "tp_aoe_max", "A_max", "tp_aoe_samp", "wf_cusp", "curr_av", "wf_pz"])
    tp0 = to_index(g["tp_0_est"], "tp_0_est")
    exact(g["cuspEmax"], np.max(g["wf_cusp"], axis=1), "cuspEmax")
    exact(g["cuspEftp"], oracle.fixed_time_pickoff(g["wf_cusp"], F(50), "i")[0], "cuspEftp")
    _, ta, _, amax_, _ = oracle.min_max(g["curr_av"])
    exact(g["tp_aoe_max"], ta, "tp_aoe_max")
    exact(g["A_max"], amax_, "A_max")
    exact(g["tp_aoe_samp"], to_ns(tp0 + ta / F(16)), "tp_aoe_samp")
    same_as_production(g, ["cuspEmax", "cuspEftp"])
    # window, difference quotient and repetition keep the samples as they are; the three moving averages are a filter: inside a program
    # (the instrumented one, which keeps curr_av) they replay the reference's rounding, 1e-6 of the peak ...
    up = oracle.upsampler(oracle.avg_current(oracle.windower(g["wf_pz"], tp0, 301)[0], 1)[0], 16, 4784)[0]
    av = oracle.moving_window_multi(up, 48, 3, 0)[0]
    assert np.max(np.nanmax(np.abs(g["curr_av"] - av), axis=1) / np.nanmax(np.abs(av), axis=1)) <= 1e-6
    # ... and the production recipe runs the branch on the lane-per-waveform kernel (dsp_current.hip), which walks the reference's loops
    # themselves: its outputs are the ORACLE's on the device's pole-zero rows and start time, bit for bit
    _, ta_x, _, amax_x, _ = oracle.min_max(av)
    exact(prod["tp_aoe_max"], ta_x, "tp_aoe_max (production)")
    exact(prod["A_max"], amax_x, "A_max (production)")
    exact(prod["tp_aoe_samp"], to_ns(tp0 + ta_x / F(16)), "tp_aoe_samp (production)")
    assert "dsp_current_kernel" in [st["chain"].kernel_name for st in chain_p._stages]
    # (the t0 filter -- piecewise constant -- with min_max and tp_0_est's walk on the run-length FIR kernel: wf_t0_filter is not stored)
    assert [k for _what, k in chain_p.kernels()] == ["dsp_fit_rows_kernel", "dsp_pz_rows_kernel", "dsp_fir_runs_kernel", "dsp_fir_f16_kernel",
                                                     "dsp_rows_kernel", "dsp_current_kernel", "dsp_reduce_kernel",
                                                     "dsp_scalar_kernel", "dsp_vm_kernel<float>", "dsp_reduce_kernel", "dsp_scalar_kernel"]  # (scalar head, program, walks, tail)
    seen.update({"tp_aoe_max": True, "A_max": True, "tp_aoe_samp": True})
    cusp_dev = g["wf_cusp"]
    assert all(seen.get(k) for k in recipes.ICPC["outputs"] if k not in ("tp_min", "tp_max", "wf_min", "wf_max", "bl_mean", "bl_slope",
                                                                         "bl_intercept", "pz_mean", "pz_std"))
    # ---- the filtered waveforms against the oracle's own, 1e-6 of each waveform's peak (north_star's bar for float32 filter outputs)
    w = wf.astype(F)
    o_bl = oracle.bl_subtract(w, bl)[0]
    o_pz = oracle.pole_zero(o_bl, F(27460.0 / dt))[0]
    k0, kc = golden_util.recipe_kernel("t0"), golden_util.recipe_kernel("cusp")  # (the reference generators' own output: fixtures)
    pairs = {"wf_pz": (pz_dev, o_pz), "wf_t0_filter": (wt0_dev, oracle.convolve_wf(o_pz, k0, "s", 8192)[0]),
             "wf_atrap": (atrap_dev, oracle.asym_trap_filter(o_pz, 8, 4, 125)[0]), "wf_trap": (trap_dev, oracle.trap_norm(o_pz, 625, 188)[0]),
             "wf_etrap": (etrap_dev, oracle.trap_norm(o_pz, 500, 125)[0]), "wf_trap2": (trap2_dev, oracle.trap_norm(o_pz, 250, 6)[0]),
             "wf_cusp": (cusp_dev, oracle.convolve_wf(o_bl, kc, "v", 301, in_len=8192 - 2100)[0])}
    worst = {}
    for k, (got, want) in pairs.items():
        worst[k] = float(np.max(np.max(np.abs(got - want), axis=1) / np.max(np.abs(want), axis=1)))
        assert worst[k] <= 1e-6, (k, worst[k])
    # ---- what that leaves end to end: rows whose index outputs differ from the oracle run on its OWN waveforms, and by how much
    want_all, _ = _expected(wf, bl, t0_ns)
    report = {}
    for k in ("tp_0_est", "tp_0_atrap", "tp_10", "tp_50", "tp_90", "tp_99", "tp_100", "tp_aoe_max"):
        d = np.abs(prod[k].astype(np.float64) - want_all[k].astype(np.float64)) / (1.0 if k == "tp_aoe_max" else dt)
        d = np.where(np.isnan(d), 0.0 if np.array_equal(np.isnan(prod[k]), np.isnan(want_all[k])) else np.inf, d)
        report[k] = {"rows_differing": int(np.sum(d > 0)), "of": n, "max_samples": float(d.max())}
    print("filter outputs, worst deviation / peak:", {k: f"{v:.1e}" for k, v in worst.items()})
    print("end-to-end index outputs vs the all-oracle run:", report)
    import json
    import os

    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/icpc_parity_report.json", "w") as f:
        json.dump({"rows": n, "filter_outputs_worst_rel_to_peak": worst, "end_to_end_index_outputs": report}, f, indent=1)


def test_time_coordinates_between_grids():
    """test_proc_chain_coordinate_grid of the reference (tests/test_processing_chain.py:324-386), restated on synthetic rows: a time
    picked off a window of the waveform equals the one picked off the whole waveform, whatever grid the processor works on; and
    test_proc_chain_unit_conversion (:289-318): a constant in any time unit is the same sample."""
    from dspeed_amd.processing_chain import WaveformInput, build_processing_chain

    rng = np.random.default_rng(7)
    n = 40
    wf, bl = _synth(rng, n)
    t0_ns = (rng.integers(100, 200, n) * 16).astype(F)
    M = "dspeed.processors"
    rec = {"outputs": ["a_unitless", "a_ns", "a_us", "a_window", "a_whole", "tp", "tp_window"], "processors": {
        "a_unitless": {"function": "fixed_time_pickoff", "module": M, "args": ["waveform", 100, "'n'", "a_unitless"]},
        "a_ns": {"function": "fixed_time_pickoff", "module": M, "args": ["waveform", "1600*ns", "'n'", "a_ns"]},
        "a_us": {"function": "fixed_time_pickoff", "module": M, "args": ["waveform", "1.6*us", "'n'", "a_us"]},
        "a_window": {"function": "fixed_time_pickoff", "module": M, "unit": ["ADC"],
                     "args": ["waveform[2625:6025]", "70.4*us + waveform.offset", "'i'", "a_window"]},
        "a_whole": {"function": "fixed_time_pickoff", "module": M, "unit": ["ADC"],
                    "args": ["waveform", "70.4*us + waveform.offset", "'i'", "a_whole"]},
        "tp": {"function": "time_point_thresh", "module": M, "unit": "ns",
               "args": ["waveform", "a_window", "72*us+waveform.offset", 0, "tp"]},
        "tp_window": {"function": "time_point_thresh", "module": M, "unit": "ns",
                      "args": ["waveform[2625:6025]", "a_window", "72*us+waveform.offset", 0, "tp_window"]}}}
    chain, _, out = build_processing_chain(rec, {"waveform": WaveformInput(wf, 16.0, t0_ns)})
    chain.execute()
    assert np.array_equal(out["a_unitless"], wf[:, 100].astype(F))
    assert np.array_equal(out["a_unitless"], out["a_ns"]) and np.array_equal(out["a_unitless"], out["a_us"])
    assert np.array_equal(out["a_window"], wf[:, 4400].astype(F)) and np.array_equal(out["a_window"], out["a_whole"])
    assert np.array_equal(out["tp_window"], out["tp"]) and not np.isnan(out["tp"]).any()
    w = wf.astype(F)
    idx = oracle.time_point_thresh(w, out["a_window"], F(4500), 0)[0]
    assert np.array_equal(out["tp"], ((idx.astype(np.float64) + t0_ns / 16.0) * 16.0).astype(F))


def test_rounding_functions_of_the_argument_language():
    """test_proc_chain_round of the reference (tests/test_processing_chain.py:389-449): time coordinates and constants"""
    from dspeed_amd.processing_chain import WaveformInput, build_processing_chain

    rng = np.random.default_rng(9)
    wf, _ = _synth(rng, 16)
    M = "dspeed.processors"
    rec = {"outputs": ["tp_max", "t_round", "t_floor", "t_ceil", "t_trunc", "c_round", "c_floor", "c_ceil", "c_trunc"], "processors": {
        "tp_min, tp_max, wf_min, wf_max": {"function": "min_max", "module": M, "args": ["waveform", "tp_min", "tp_max", "wf_min", "wf_max"],
                                           "unit": ["us", "us", "ADC", "ADC"]},
        "t_round": "round(tp_max, 1*us)", "t_floor": "floor(tp_max, 1*us)", "t_ceil": "ceil(tp_max, 1*us)", "t_trunc": "trunc(tp_max, 1*us)",
        "c_round": "round(1*us, waveform.period)", "c_floor": "floor(1*us, waveform.period)", "c_ceil": "ceil(1*us, waveform.period)",
        "c_trunc": "trunc(1*us, waveform.period)"}}
    chain, _, out = build_processing_chain(rec, {"waveform": WaveformInput(wf, 16.0)})
    chain.execute()
    tp = out["tp_max"]
    assert np.array_equal(tp, (np.argmax(wf, axis=1) * 0.016).astype(F))  # written in us
    assert np.array_equal(np.rint(tp), out["t_round"]) and np.array_equal(np.floor(tp), out["t_floor"])
    assert np.array_equal(np.ceil(tp), out["t_ceil"]) and np.array_equal(np.trunc(tp), out["t_trunc"])
    assert out["c_round"][0] == 992 and out["c_floor"][0] == 992 and out["c_ceil"][0] == 1008 and out["c_trunc"][0] == 992


def test_cpu_port_of_the_whole_recipe_beside_the_device():
    """Not a parity test: times the oracle composition above (the CPU port of the same recipe, one thread, processor by processor on
    whole arrays) beside the device program on the same rows and leaves both figures in gpurun_out/ for profiles/."""
    import json
    import os
    import time

    from dspeed_amd.device import DeviceArray, sync
    from dspeed_amd.processing_chain import WaveformInput, build_processing_chain

    rng = np.random.default_rng(11)
    n_cpu, n_gpu = 128, 32768
    wf, bl = _synth(rng, n_cpu)
    t0_ns = np.full(n_cpu, 48000.0, dtype=F)
    t = time.perf_counter()
    want, _ = _expected(wf, bl, t0_ns)
    cpu_rate = n_cpu / (time.perf_counter() - t)
    reps = n_gpu // n_cpu
    d_wf, d_bl = DeviceArray.from_numpy(np.tile(wf, (reps, 1))), DeviceArray.from_numpy(np.tile(bl, reps))
    tb = {"waveform": WaveformInput(d_wf, 16.0, 48000.0), "baseline": d_bl}
    chain, _, _ = build_processing_chain(recipes.ICPC, tb)
    outs = {k: DeviceArray((n_gpu,), np.float32) for k in recipes.ICPC["outputs"]}
    chain.link(tb, outs)
    chain.execute()
    sync()
    t = time.perf_counter()
    for _ in range(3):
        chain.execute()
    sync()
    gpu_rate = 3 * n_gpu / (time.perf_counter() - t)
    got = outs["trapEmax"].to_numpy()[:n_cpu]
    assert np.max(np.abs(got - want["trapEmax"]) / want["trapEmax"]) <= 1e-6
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/icpc_cpu_port.json", "w") as f:
        json.dump({"recipe": "ICPC structure (tests/recipes.py), 8192-sample uint16 rows", "cpu_port_waveforms_per_s_1_thread": cpu_rate,
                   "cpu_rows": n_cpu, "device_waveforms_per_s": gpu_rate, "device_rows": n_gpu, "ratio": gpu_rate / cpu_rate}, f, indent=1)
    assert gpu_rate > 10 * cpu_rate


def test_stage_buffers_are_bounded_and_pieces_give_the_same_results():
    """the rows the stages ahead of the program leave in HBM (the pole-zero rows and the cusp-filtered ones: 33 kB per waveform here) are allocated per piece: with a small bound the batch
    is walked in equal pieces -- host-resident and device-resident -- and every output equals the one-piece run bit for bit"""
    from dspeed_amd.device import DeviceArray
    from dspeed_amd.processing_chain import WaveformInput, build_processing_chain

    rng = np.random.default_rng(7)
    n = 150
    wf, bl = _synth(rng, n)
    wf = wf.astype(np.uint16)
    tb = {"waveform": WaveformInput(wf, 16.0, 48000.0), "baseline": bl}
    chain, _, out = build_processing_chain(recipes.ICPC, tb)
    assert len(chain._stages) == 7
    chain.execute()
    ref = {k: np.array(v) for k, v in out.items()}
    per_row = sum(4 * (1 if ln is None else ln) for st in chain._stages for _o, _k, ln in st["outs"])
    chain.stage_bytes = 40 * per_row  # -> 4 pieces of 38 rows
    for v in out.values():
        v[...] = 0
    chain.execute()
    assert max(len(b) for st in chain._stages for b in st["bufs"].values()) >= 150  # (allocated by the first run; not grown)
    for k in ref:
        assert np.array_equal(out[k], ref[k], equal_nan=True), k
    # device-resident rows: views of the columns per piece
    chain2, _, out2 = build_processing_chain(recipes.ICPC, tb)
    chain2.stage_bytes = 40 * per_row
    d_in = {"waveform": WaveformInput(DeviceArray.from_numpy(wf), 16.0, 48000.0), "baseline": DeviceArray.from_numpy(bl)}
    d_out = {k: DeviceArray(v.shape, v.dtype) for k, v in out2.items()}
    chain2.link(d_in, d_out)
    chain2.execute()
    assert max(len(b) for st in chain2._stages for b in st["bufs"].values()) == 38
    assert len(chain2._lanes) == 2 and max(len(b) for held in chain2._lanes[1].stage_bufs for b in held.values()) == 38  # two pieces in flight
    for k in ref:
        assert np.array_equal(d_out[k].to_numpy(), ref[k], equal_nan=True), k


def test_results_do_not_depend_on_the_neighbours():
    """the wavefronts of a workgroup keep their waveforms in neighbouring LDS regions, and which rows are neighbours depends on the
    launch: the same rows in one launch of 2 000 and in launches of 37 must give the same bits (a store past a slot's guard into the
    next region would show here)"""
    from dspeed_amd.processing_chain import WaveformInput, build_processing_chain

    rng = np.random.default_rng(99)
    n = 2000
    wf, bl = _synth(rng, n)
    wf = wf.astype(np.uint16)
    tb = {"waveform": WaveformInput(wf, 16.0, 48000.0), "baseline": bl}
    chain, _, out = build_processing_chain(recipes.ICPC, tb)
    chain.execute()
    whole = {k: np.array(v) for k, v in out.items()}
    for v in out.values():
        v[...] = 0
    for a in range(0, n, 37):
        chain.execute(a, min(n, a + 37))
    for k in whole:
        assert np.array_equal(out[k], whole[k], equal_nan=True), k


def test_current_branch_alone_moves_its_start_ahead_too():
    """asked for the current branch only, the recipe has no t0 chain on rows that would have put tp_0_est into HBM: the builder moves what
    computes the window's start ahead of the program by itself, the branch runs on the lane-per-waveform kernel, and the numbers are the
    full recipe's"""
    from dspeed_amd.processing_chain import WaveformInput, build_processing_chain

    rng = np.random.default_rng(31)
    n = 70
    wf, bl = _synth(rng, n)
    tb = {"waveform": WaveformInput(wf, 16.0, 48000.0), "baseline": bl}
    full, _, ref = build_processing_chain(recipes.ICPC, tb)
    full.execute()
    part, _, out = build_processing_chain(recipes.ICPC, tb, outputs=["A_max", "tp_aoe_max", "tp_aoe_samp"])
    part.execute()
    assert [st["what"] for st in part._stages][-2:] == ["convolve_wf wf_t0_filter on prefix sums + per-event values of wf_t0_filter",
                                                        "current branch of wf_pz on rows"]
    assert "dsp_current_kernel" in [k for _w, k in part.kernels()]
    for k in out:
        assert np.array_equal(out[k], ref[k], equal_nan=True), k


def test_stages_on_side_streams_give_the_same_results():
    """stages that do not read each other's results on streams of their own (ProcessingChain.concurrent_stages): the same numbers, and the plan
    keeps every reader behind what it reads"""
    from dspeed_amd.processing_chain import WaveformInput, build_processing_chain

    rng = np.random.default_rng(17)
    n = 700
    wf, bl = _synth(rng, n)
    tb = {"waveform": WaveformInput(wf.astype(np.uint16), 16.0, 48000.0), "baseline": bl}
    chain, _, out = build_processing_chain(recipes.ICPC, tb)
    chain.execute()
    ref = {k: np.array(v) for k, v in out.items()}
    chain2, _, out2 = build_processing_chain(recipes.ICPC, tb)
    chain2.concurrent_stages = True
    plan = chain2._stage_plan()
    assert plan["n_side"] >= 2 and all(i < j for j, d in enumerate(plan["deps"]) for i in d)
    what = [st["what"] for st in chain2._stages]
    pzs, t0f = what.index("wf_pz -> HBM + min_max of waveform"), what.index("convolve_wf wf_t0_filter on prefix sums + per-event values of wf_t0_filter")
    assert pzs in plan["deps"][t0f] and t0f in plan["deps"][what.index("asym_trap_filter wf_atrap on rows")]
    assert plan["stream_of"][what.index("fft_convolve_wf wf_cusp")] != plan["stream_of"][t0f]
    for _ in range(3):  # (a lane's buffers are reused by its next pass: the side streams wait for the pass before)
        for v in out2.values():
            v[...] = 0
        chain2.execute()
        for k in ref:
            assert np.array_equal(out2[k], ref[k], equal_nan=True), k


def test_row_scales_from_the_kernel_that_reads_the_same_rows(monkeypatch):
    """the cusp filter runs on the float16 matrix instructions and needs every row's scale (a power of two from max |waveform[0:6092] - baseline|)
    and flags; the kernel that writes the pole-zero rows reads the same rows and subtracts the same baseline, so it leaves them
    (dsp_chain_share_row_scales, input side) and the filter skips its own pass over the rows: bit for bit what it computes itself -- ordinary
    rows, rows equal to their baseline, a NaN baseline, a huge one"""
    from dspeed_amd.processing_chain import WaveformInput, build_processing_chain

    rng = np.random.default_rng(41)
    n = 500
    wf, bl = _synth(rng, n)
    wf = wf.astype(np.int16)
    bl = bl.copy()
    wf[3] = 1200
    bl[3] = 1200.0          # x - baseline = 0 everywhere: the scale of a zero row
    bl[4] = np.nan          # -> flags: a NaN row
    bl[5] = 3e38            # a magnitude whose scale would leave float32: the filter's slow path
    bl[6] = -2.5e-3
    tb = {"waveform": WaveformInput(wf, 16.0, 48000.0), "baseline": bl}
    outs = ["cuspEmax", "cuspEftp", "tp_0_est", "trapEmax", "wf_max", "tp_max"]

    def run(shared):
        monkeypatch.setenv("DSPEED_HIP_NO_SHARED_ROW_SCALES", "0" if shared else "1")
        chain, _, out = build_processing_chain(recipes.ICPC, tb, outputs=outs)
        chain.execute()
        ks = [k for _w, k in chain.kernels()]
        assert "dsp_pz_rows_kernel" in ks and "dsp_fir_f16_kernel" in ks, ks
        return {k: np.array(v) for k, v in out.items()}

    a, b = run(True), run(False)
    for k in outs:
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert np.isnan(a["cuspEmax"][4]) and np.isfinite(a["cuspEmax"][[0, 1, 2, 6]]).all() and a["cuspEmax"][3] == 0.0


def test_programs_that_shed_ops_give_the_same_bits(monkeypatch):
    """the scalar head (arithmetic that needs nothing of the program, run ahead of it with a row per lane), the thresholds folded into the walks
    (`time_point_thresh(wf, 0.9 * trapTmax, ...)`: the planner drops the multiplication's op and the walk multiplies) and the rise-time walks as
    a launch of their own behind the program change where an operation runs, not the operation: every output bit-identical with either switched
    off"""
    from dspeed_amd.processing_chain import WaveformInput, build_processing_chain

    rng = np.random.default_rng(29)
    n = 600
    wf, bl = _synth(rng, n)
    tb = {"waveform": WaveformInput(wf.astype(np.uint16), 16.0, 48000.0), "baseline": bl}
    chain, _, out = build_processing_chain(recipes.ICPC_REF, tb)
    chain.execute()
    assert "per-event arithmetic ahead of the program" in [st["what"] for st in chain._stages]
    ref = {k: np.array(v) for k, v in out.items()}
    # (and the fits on a stream of their own beside the first stages, or on the lane's stream: DSPEED_HIP_FITS_BESIDE)
    for switch, value in (("DSPEED_HIP_NO_SCALAR_HEAD", "1"), ("DSPEED_HIP_NO_THRESHOLD_FOLD", "1"), ("DSPEED_HIP_NO_SCALAR_TAIL", "1"),
                          ("DSPEED_HIP_NO_WALKS_BEHIND", "1"), ("DSPEED_HIP_FITS_BESIDE", "0")):
        monkeypatch.setenv(switch, value)
        if switch == "DSPEED_HIP_FITS_BESIDE":
            from dspeed_amd.processing_chain import ProcessingChain

            monkeypatch.setattr(ProcessingChain, "fits_beside_stages", False)  # (the class attribute was read from the environment at import)
        other, _, out2 = build_processing_chain(recipes.ICPC_REF, tb)
        for _ in range(2):  # (twice: the second pass's fits start behind the first pass's readers of their columns)
            other.execute()
        monkeypatch.delenv(switch)
        if switch == "DSPEED_HIP_NO_SCALAR_HEAD":
            assert "per-event arithmetic ahead of the program" not in [st["what"] for st in other._stages]
        for k in ref:
            assert np.array_equal(np.asarray(out2[k]), ref[k], equal_nan=True), (switch, k)


def test_a_team_of_wavefronts_per_row_gives_the_same_results(monkeypatch):
    """the recipe's program loads the pole-zero rows and then only reads them: its ops fall into three groups that share no register (a trapezoid
    with its five walks, a trapezoid with a pick-off, a pick-off of a third), and a team of three wavefronts per row runs them on the one LDS image,
    each loading a third of the row (dsp_chain_create, DevProgram.team); DSPEED_HIP_TEAM_MAX=2 makes it a team of two, DSPEED_HIP_NO_TEAMS=1 keeps
    a wavefront per row"""
    from dspeed_amd.processing_chain import WaveformInput, build_processing_chain

    rng = np.random.default_rng(23)
    n = 1000  # (not a multiple of the rows a workgroup takes: the last workgroup has idle row slots that still meet its barriers)
    wf, bl = _synth(rng, n)
    tb = {"waveform": WaveformInput(wf.astype(np.uint16), 16.0, 48000.0), "baseline": bl}
    chain, _, out = build_processing_chain(recipes.ICPC, tb)
    chain.execute()
    g = chain._chain.geometry(n)
    assert chain._chain.kernel_name.startswith("dsp_vm") and g["waves_per_block"] == 3, g   # (a workgroup per team)
    monkeypatch.setenv("DSPEED_HIP_TEAM_WPB", "4")  # four teams to a workgroup: the last one has idle row slots that still meet its barriers
    four, _, out4 = build_processing_chain(recipes.ICPC, tb)
    four.execute()
    assert four._chain.geometry(n)["waves_per_block"] == 12
    for k in out4:
        assert np.array_equal(np.asarray(out[k]), np.asarray(out4[k]), equal_nan=True), k
    monkeypatch.delenv("DSPEED_HIP_TEAM_WPB")
    monkeypatch.setenv("DSPEED_HIP_TEAM_MAX", "2")
    two, _, out2 = build_processing_chain(recipes.ICPC, tb)
    two.execute()
    assert two._chain.geometry(n)["waves_per_block"] == 2
    for k in out2:
        assert np.array_equal(np.asarray(out[k]), np.asarray(out2[k]), equal_nan=True), k
    monkeypatch.delenv("DSPEED_HIP_TEAM_MAX")
    monkeypatch.setenv("DSPEED_HIP_NO_TEAMS", "1")
    single, _, ref = build_processing_chain(recipes.ICPC, tb)
    single.execute()
    assert single._chain.geometry(n)["waves_per_block"] == 4
    for k in ref:
        assert np.array_equal(np.asarray(out[k]), np.asarray(ref[k]), equal_nan=True), k
    for _ in range(2):  # (again on the same handle: the image and the registers are reused row after row)
        chain.execute()
        for k in ref:
            assert np.array_equal(np.asarray(out[k]), np.asarray(ref[k]), equal_nan=True), k
