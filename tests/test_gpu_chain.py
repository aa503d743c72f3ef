"""GPU parity of the fused chains (one launch, intermediates in LDS) against the golden chain fixtures and the oracle."""
import json

import numpy as np
import pytest

import oracle
from golden_util import assert_rel_to_peak, cases

pytestmark = pytest.mark.gpu

TOL = 1e-6


def _run_energy(wf, bl, tp, tau, rise, flat, mode="l", fused=True, trap="trap_filter"):
    """fused=True: the specialised energy kernel (dsp_energy.hip) when the shape allows; False: the generic waveform VM."""
    from dspeed_amd.chain import Chain, energy_chain_program
    from dspeed_amd.device import DeviceArray

    n_wf, wf_len = wf.shape
    ch = Chain(energy_chain_program(wf_len, tau, rise, flat, mode, wf_dtype=wf.dtype, trap=trap), "energy")
    ch.set_fused(fused)
    bufs = {"waveform": DeviceArray.from_numpy(wf), "baseline": DeviceArray.from_numpy(bl), "t_pick": DeviceArray.from_numpy(tp),
            "trapEftp": DeviceArray((n_wf,), np.float32)}
    ch.execute(bufs, n_wf)
    ch.check()
    return bufs["trapEftp"].to_numpy()


@pytest.mark.parametrize("fused", [1, 13, 15, 0])
def test_energy_chain_golden(fused):
    c2 = cases("chains")[1]
    p = c2.params
    got = _run_energy(c2["waveform"], c2["baseline"], c2["t_pick"], p["tau"], p["rise"], p["flat"], p["mode"], fused=fused)
    want = c2["trapEftp"]
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    rel = np.abs(got[ok] - want[ok]) / np.abs(want[ok])
    print("fused energy chain vs golden: max rel", rel.max())
    assert rel.max() <= TOL


@pytest.mark.parametrize("fused", [1, 13, 15, 0])
@pytest.mark.parametrize("wf_len,rise,flat", [(4096, 625, 188), (1024, 64, 16), (8192, 1250, 376), (6092, 500, 100), (2048, 300, 7),
                                              (3000, 128, 0), (200, 10, 3)])
def test_energy_chain_vs_oracle(wf_len, rise, flat, fused):
    rng = np.random.default_rng(wf_len + rise)
    n_wf = 300
    i = np.arange(wf_len, dtype=np.float64)[None, :]
    B = rng.uniform(9000, 11000, (n_wf, 1))
    A = rng.uniform(500, 15000, (n_wf, 1))
    t0 = np.floor(rng.uniform(0.45, 0.55, (n_wf, 1)) * wf_len)
    wf = (B + A * np.exp(-(i - t0) / 1716.28) * (i >= t0) + 5.0 * rng.standard_normal((n_wf, wf_len))).astype(np.float32)
    bl = B[:, 0].astype(np.float32)
    tp = (t0[:, 0] + rise + 0.8 * flat).astype(np.float32)
    wf[11, 17] = np.nan
    bl[14] = np.nan
    tp[15] = np.nan
    tp[12] = np.float32(np.floor(tp[12]))
    tp[13] = np.float32(wf_len + 3)
    for mode in "lnh":
        got = _run_energy(wf, bl, tp, 1716.28, rise, flat, mode, fused=fused)
        want, rc = oracle.chain_energy(wf, bl, tp, 1716.28, rise, flat, mode)
        assert rc == 0
        assert np.array_equal(np.isnan(got), np.isnan(want))
        ok = ~np.isnan(want)
        rel = np.abs(got[ok] - want[ok]) / np.abs(want[ok])
        print(f"len={wf_len} mode={mode}: max rel {rel.max():.2e}, median {np.median(rel):.2e}, bit-exact {np.mean(got[ok] == want[ok]):.2f}")
        assert rel.max() <= TOL


@pytest.mark.parametrize("fused", [1, 15])
@pytest.mark.parametrize("wf_len,rise,flat", [(4096, 300, 50), (2048, 100, 31), (1024, 40, 9), (8192, 600, 100)])
def test_energy_pickoff_position_sweep(wf_len, rise, flat, fused):
    """the picked-off samples are caught at run-time positions: sweep the time point over lane-chunk boundaries (C = len/64 + 1
    samples per lane in the default kernel), capture-block boundaries, both ends of the waveform, integer and fractional times"""
    C = wf_len // 64 + 1
    pos = sorted(set([0, 1, 2, 3, 14, 15, 16, 17, 31, 32, 33, C - 2, C - 1, C, C + 1, 2 * C - 1, 2 * C, 2 * C + 15, 2 * C + 16, 5 * C + 47,
                      5 * C + 48, 31 * C + C // 2, 62 * C, 63 * C - 1, 63 * C, wf_len - 3, wf_len - 2, wf_len - 1]))
    tp = np.array([p + f for p in pos for f in (0.0, 0.3, 0.5, 0.75)], dtype=np.float32)
    n_wf = tp.size
    rng = np.random.default_rng(wf_len)
    i = np.arange(wf_len, dtype=np.float64)[None, :]
    t0 = np.floor(rng.uniform(0.1, 0.9, (n_wf, 1)) * wf_len)
    wf = (10000 + 8000 * np.exp(-(i - t0) / 1716.28) * (i >= t0) + 5.0 * rng.standard_normal((n_wf, wf_len))).astype(np.float32)
    bl = np.full(n_wf, 10000, dtype=np.float32)
    # the filter-output tolerance is relative to the trapezoid's peak in that waveform (its rounding noise does not shrink where the
    # output is near zero)
    trap, rc = oracle.chain_pz_trap(wf - bl[:, None], 1716.28, rise, flat)
    assert rc == 0
    peak = np.max(np.abs(trap), axis=1)
    for mode in "lnhfc":
        got = _run_energy(wf, bl, tp, 1716.28, rise, flat, mode, fused=fused)
        want, rc = oracle.chain_energy(wf, bl, tp, 1716.28, rise, flat, mode)
        assert rc == 0
        assert np.array_equal(np.isnan(got), np.isnan(want)), mode
        ok = ~np.isnan(want)
        assert np.max(np.abs(got[ok] - want[ok]) / peak[ok]) <= TOL, mode


@pytest.mark.parametrize("dtype", [np.int16, np.uint16])
@pytest.mark.parametrize("wf_len,rise,flat", [(4096, 625, 188), (2048, 300, 7), (1024, 64, 16), (8192, 1250, 376)])
def test_energy_chain_on_digitiser_samples(dtype, wf_len, rise, flat):
    """16-bit rows (what the digitisers write) take the float32 loop like in the reference (ufunc casting, processing_chain.py:1565-1572):
    the default kernel widens them while staging.  Same results as the float32 copy of the same samples, the VM and the oracle."""
    rng = np.random.default_rng(wf_len)
    n_wf = 257
    i = np.arange(wf_len, dtype=np.float64)[None, :]
    off = 0 if dtype == np.int16 else 20000
    B = rng.uniform(-3000, 3000, (n_wf, 1)) + off
    A = rng.uniform(500, 15000, (n_wf, 1))
    t0 = np.floor(rng.uniform(0.45, 0.55, (n_wf, 1)) * wf_len)
    wf = np.rint(B + A * np.exp(-(i - t0) / 1716.28) * (i >= t0) + 5.0 * rng.standard_normal((n_wf, wf_len))).astype(dtype)
    if dtype == np.int16:
        wf[3, 5] = -32768
        wf[4, 6] = 32767
    else:
        wf[3, 5] = 0
        wf[4, 6] = 65535
    bl = B[:, 0].astype(np.float32)
    tp = (t0[:, 0] + rise + 0.8 * flat).astype(np.float32)
    want, rc = oracle.chain_energy(wf.astype(np.float32), bl, tp, 1716.28, rise, flat, "l")
    assert rc == 0
    from dspeed_amd.chain import Chain, energy_chain_program

    ch = Chain(energy_chain_program(wf_len, 1716.28, rise, flat, "l", wf_dtype=dtype), "k")
    assert ch.kernel_name == "dsp_energy_rr_kernel"  # the specialised kernel takes these rows directly
    got = _run_energy(wf, bl, tp, 1716.28, rise, flat, "l", fused=1)
    assert np.max(np.abs(got - want) / np.abs(want)) <= TOL
    assert np.array_equal(got, _run_energy(wf.astype(np.float32), bl, tp, 1716.28, rise, flat, "l", fused=1))  # widening is exact
    vm = _run_energy(wf, bl, tp, 1716.28, rise, flat, "l", fused=0)
    assert np.max(np.abs(vm - want) / np.abs(want)) <= TOL
    # the classic kernel reads float32 rows only: asking for it on 16-bit rows runs the VM, not something wrong
    assert np.array_equal(_run_energy(wf, bl, tp, 1716.28, rise, flat, "l", fused=15), vm)


@pytest.mark.parametrize("row_stride,offset_rows", [(4096, 0), (4100, 0), (4097, 0), (4104, 3)])
def test_energy_chain_on_strided_and_offset_rows(row_stride, offset_rows):
    """rows inside a wider allocation (row_stride > wf_len) and a buffer that starts mid-allocation: 16-byte aligned rows take the
    specialised kernel, anything else the VM's element-wise loads; same energies either way"""
    from dspeed_amd.chain import Chain, energy_chain_program
    from dspeed_amd.device import DeviceArray

    rng = np.random.default_rng(row_stride)
    n_wf, wf_len = 70, 4096
    i = np.arange(wf_len, dtype=np.float64)[None, :]
    t0 = np.floor(rng.uniform(0.45, 0.55, (n_wf, 1)) * wf_len)
    wf = (10000 + 7000 * np.exp(-(i - t0) / 1716.28) * (i >= t0) + 5.0 * rng.standard_normal((n_wf, wf_len))).astype(np.float32)
    bl = np.full(n_wf, 10000, dtype=np.float32)
    tp = (t0[:, 0] + 625 + 150.4).astype(np.float32)
    want, _ = oracle.chain_energy(wf, bl, tp, 1716.28, 625, 188, "l")
    big = np.full((n_wf + offset_rows, row_stride), np.nan, dtype=np.float32)  # NaN everywhere the chain must not look
    big[offset_rows:, :wf_len] = wf
    dev = DeviceArray.from_numpy(big)
    view = dev.view_rows(offset_rows, offset_rows + n_wf)
    ch = Chain(energy_chain_program(wf_len, 1716.28, 625, 188, "l", row_stride=row_stride), "strided")
    out = DeviceArray((n_wf,), np.float32)
    ch.execute({"waveform": view, "baseline": DeviceArray.from_numpy(bl), "t_pick": DeviceArray.from_numpy(tp), "trapEftp": out}, n_wf)
    ch.check()
    got = out.to_numpy()
    assert np.max(np.abs(got - want) / np.abs(want)) <= TOL
    aligned = row_stride % 4 == 0
    assert ch.kernel_name == ("dsp_energy_rr_kernel" if aligned else "dsp_vm_kernel<float>")


def test_energy_chain_matches_unfused_processors():
    """fused chain == the same processors called one by one on the device (the ProcessingChain way)"""
    from dspeed_amd import processors as P

    rng = np.random.default_rng(99)
    n_wf, wf_len = 64, 4096
    wf = (10000 + 3000 * (np.arange(wf_len)[None, :] > 2000) + 5 * rng.standard_normal((n_wf, wf_len))).astype(np.float32)
    bl = np.full(n_wf, 10000, dtype=np.float32)
    tp = np.full(n_wf, 2000 + 625 + 150.4, dtype=np.float32)
    # the classic kernel (15) and the VM (0) use the same chunking as the single processors -> identical bits; the default
    # register-resident kernel (1 = 13) replays the rounding sequence over 65-sample chunks -> equal within the filter tolerance
    step = P.fixed_time_pickoff(P.trap_filter(P.pole_zero(P.bl_subtract(wf, bl), 1716.28), 625, 188), tp, ord("l"))
    for fused in (15, 0):
        assert np.array_equal(_run_energy(wf, bl, tp, 1716.28, 625, 188, fused=fused), step)
    for fused in (1, 13):
        assert np.max(np.abs(_run_energy(wf, bl, tp, 1716.28, 625, 188, fused=fused) - step) / np.abs(step)) <= TOL
    step = P.fixed_time_pickoff(P.trap_norm(P.pole_zero(P.bl_subtract(wf, bl), 1716.28), 625, 188), tp, ord("h"))
    for fused in (15, 0):
        assert np.array_equal(_run_energy(wf, bl, tp, 1716.28, 625, 188, mode="h", fused=fused, trap="trap_norm"), step)
    for fused in (1, 13):
        assert np.max(np.abs(_run_energy(wf, bl, tp, 1716.28, 625, 188, mode="h", fused=fused, trap="trap_norm") - step) / np.abs(step)) <= TOL


def test_data_dependent_fatal_reports_row():
    from dspeed_amd.errors import DSPFatal

    wf = np.ones((10, 256), dtype=np.float32)
    bl = np.zeros(10, dtype=np.float32)
    tp = np.full(10, 100.0, dtype=np.float32)
    tp[6] = 100.5
    for fused in (1, 13, 15, 0):
        with pytest.raises(DSPFatal) as ei:
            _run_energy(wf, bl, tp, 100.0, 16, 8, "i", fused=fused)
        assert ei.value.wf_range == range(6, 7)
        assert "integer t_in" in str(ei.value)


def test_a_chain_that_misses_a_specialised_kernel_says_why(caplog):
    """dsp_chain_kernel_note: the energy chain with a time constant per event, on 3000-sample rows, with a short trapezoid; a 40-tap FIR -- each
    runs on the interpreter (same results as ever) and the chain says why, in the log of the recipe's author"""
    import logging

    import recipes
    from dspeed_amd.processing_chain import build_processing_chain

    rng = np.random.default_rng(5)
    n = 64
    M = "dspeed.processors"

    def notes(rec, tb):
        with caplog.at_level(logging.WARNING, logger="dspeed"):
            caplog.clear()
            chain, _, out = build_processing_chain(rec, tb)
            chain.execute()
        return chain, [r.getMessage() for r in caplog.records]

    wf = (1000 + 50 * rng.standard_normal((n, 4096))).astype(np.float32)
    tb = {"waveform": wf, "baseline": np.full(n, 1000, np.float32), "t_pick": np.full(n, 3000, np.float32)}
    chain, msgs = notes(recipes.C2, tb)
    assert chain.kernel_notes() == [] and msgs == [] and chain._chain.kernel_name == "dsp_energy_rr_kernel"
    # a time constant per event: the register-resident kernel's own build since round 3 -- nothing to say
    rec_tau = json.loads(json.dumps(recipes.C2))
    rec_tau["processors"]["wf_pz"] = "dspeed.processors.pole_zero(wf_blsub, tau, wf_pz)"
    chain, msgs = notes(rec_tau, dict(tb, tau=np.full(n, 1716.28, np.float32)))
    assert chain._chain.kernel_name == "dsp_energy_rr_kernel" and chain.kernel_notes() == [] and msgs == []
    # 3000-sample rows
    tb3 = {"waveform": wf[:, :3000].copy(), "baseline": tb["baseline"], "t_pick": np.full(n, 2000, np.float32)}
    chain, msgs = notes(recipes.C2, tb3)
    assert chain._chain.kernel_name.startswith("dsp_vm_kernel") and "3000 samples" in chain.kernel_notes()[0][1] and len(msgs) == 1
    # a 40-tap FIR kept as a waveform: a piecewise-constant kernel (t0_filter) has the run-length FIR kernel whatever its length ...
    rec_f = {"outputs": ["wf_f"], "processors": {
             "kern": {"function": "t0_filter", "module": M, "args": [8, 32, "kern(40, 'f')"]},
             "wf_f": {"function": "convolve_wf", "module": M, "args": ["waveform", "kern", "'s'", "wf_f(4096, 'f')"]}}}
    chain, msgs = notes(rec_f, {"waveform": wf})
    assert chain.kernels() == [("program", "dsp_fir_runs_kernel")] and chain.kernel_notes() == [] and msgs == []
    # ... any other one of fewer than 64 taps stays with the interpreter, and the chain says so
    rec_f["processors"]["kern"] = {"function": "moving_slope", "module": M, "args": ["kern(40, 'f')"]}
    chain, msgs = notes(rec_f, {"waveform": wf})
    assert any("40-tap" in note for _w, note in chain.kernel_notes()), (chain.kernels(), chain.kernel_notes())


@pytest.mark.parametrize("dtype,L", [(np.float32, 4096), (np.int16, 8192), (np.uint16, 2048), (np.float32, 1024)])
@pytest.mark.parametrize("trap", ["trap_filter", "trap_norm", "asym_trap_filter"])
def test_energy_chain_with_a_time_constant_per_event(dtype, L, trap):
    """pole_zero's tau as a per-event column (pole_zero.py:24-30, the gufunc's "()" slot): the register-resident kernel's TAU build forms
    exp(-1/tau) per row as the interpreter's op does -- the oracle's and the interpreter's energies to 1e-6 relative (as with a constant), a
    NaN time constant -> a NaN energy"""
    import recipes
    from dspeed_amd.processing_chain import build_processing_chain

    rng = np.random.default_rng(L + len(trap))
    n = 150
    i = np.arange(L)[None, :]
    t0 = np.floor(0.4 * L)
    tau = rng.uniform(800, 2500, n).astype(np.float32)
    x = 1000 + rng.uniform(500, 15000, (n, 1)) * np.exp(-np.clip(i - t0, 0, None) / tau[:, None].astype(np.float64)) * (i >= t0) + 5 * rng.standard_normal((n, L))
    wf = (np.rint(x) if np.dtype(dtype).kind in "iu" else x).astype(dtype)
    tau[7] = np.nan
    bl = np.full(n, 1000, np.float32)
    rise, flat = L // 16, L // 32
    tp = np.full(n, t0 + rise + flat // 2, np.float32)
    args = ["wf_pz", str(rise), str(flat), "wf_trap"] if trap != "asym_trap_filter" else ["wf_pz", str(rise), str(flat), str(rise // 2), "wf_trap"]
    rec = {"outputs": ["trapEftp"], "processors": {
        "wf_blsub": "dspeed.processors.bl_subtract(waveform, baseline, wf_blsub)",
        "wf_pz": "dspeed.processors.pole_zero(wf_blsub, tau, wf_pz)",
        "wf_trap": {"function": trap, "module": "dspeed.processors", "args": args},
        "trapEftp": {"function": "fixed_time_pickoff", "module": "dspeed.processors", "args": ["wf_trap", "t_pick", "'l'", "trapEftp"]}}}
    tb = {"waveform": wf, "baseline": bl, "t_pick": tp, "tau": tau}
    got = {}
    for fused in (1, 0):
        chain, _, out = build_processing_chain(rec, tb)
        chain._ensure()
        chain._chain.set_fused(fused)
        assert chain._chain.kernel_name == ("dsp_energy_rr_kernel" if fused else "dsp_vm_kernel<float>")
        chain.execute()
        got[fused] = np.array(out["trapEftp"])
    assert np.isnan(got[1][7]) and not np.isnan(np.delete(got[1], 7)).any()
    # (the kernel walks the recurrence in the reference's order, the interpreter sums prefix sums: last places)
    assert np.isnan(got[0][7]) and np.max(np.abs(np.delete(got[1], 7) - np.delete(got[0], 7)) / np.abs(np.delete(got[0], 7))) <= 1e-6
    xs = wf.astype(np.float32) - bl[:, None]
    want = np.empty(n, np.float32)
    for r in range(n):
        if r == 7:
            continue
        pz = oracle.pole_zero(xs[r:r + 1], float(tau[r]))[0]
        tr = getattr(oracle, trap)(pz, *[int(a) for a in args[1:-1]])[0]
        want[r] = oracle.fixed_time_pickoff(tr, tp[r:r + 1], "l")[0][0]
    ok = np.arange(n) != 7
    assert np.max(np.abs(got[1][ok] - want[ok]) / np.abs(want[ok])) <= 1e-6


def test_current_branch_outside_its_kernel_says_why():
    """the current branch with 40-sample moving windows (not a multiple of 16): on the interpreter, and the stage says so"""
    from dspeed_amd.processing_chain import build_processing_chain

    rng = np.random.default_rng(8)
    n, L = 40, 2048
    M = "dspeed.processors"
    wf = np.cumsum(rng.standard_normal((n, L)), axis=1).astype(np.float32)
    t0 = np.full(n, 700.0, np.float32)
    for ma, kernel, word in ((48, "dsp_current_kernel", None), (40, "dsp_vm_kernel<float>", "40 samples")):
        rec = {"outputs": ["A_max"], "processors": {
            "wf_le": {"function": "windower", "module": M, "args": ["waveform", "t0", "wf_le(301, 'f')"]},
            "curr": {"function": "avg_current", "module": M, "args": ["wf_le", 1, "curr(300, 'f')"]},
            "curr_up": {"function": "upsampler", "module": M, "args": ["curr", "16", "curr_up(4784, 'f')"]},
            "curr_av": {"function": "moving_window_multi", "module": M, "args": ["curr_up", str(ma), 3, 0, "curr_av"]},
            "t_min, t_max, A_min, A_max": f"{M}.min_max(curr_av, t_min, t_max, A_min, A_max)"}}
        chain, _, out = build_processing_chain(rec, {"waveform": wf, "t0": t0})
        chain.execute()
        names = [k for _w, k in chain.kernels()]
        assert kernel in names, names
        notes = [t for _w, t in chain.kernel_notes()]
        assert (notes == []) if word is None else any(word in t for t in notes), notes
        up = oracle.upsampler(oracle.avg_current(oracle.windower(wf, t0, 301)[0], 1)[0], 16, 4784)[0]
        av = oracle.moving_window_multi(up, ma, 3, 0)[0]
        want = oracle.min_max(av)[3]
        assert np.max(np.abs(out["A_max"] - want) / np.abs(want)) <= 1e-6
