"""The lane-per-waveform current-branch kernel (dsp_current.hip): windower -> avg_current -> upsampler -> moving_window_multi (three
alternating windows) -> min_max on float32 rows.  Every lane runs the reference's loops in their own order, so all four outputs are
bit-identical to the oracle's five processors run one after the other (reference windower.py:12-54, moving_windows.py:117-249,
upsampler.py:13-56, min_max.py:11-82), for any window start -- fractional, negative, beyond the end, NaN -- and for rows with NaN or
infinite samples."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
M = "dspeed.processors"


def _recipe(n_win=301, ac=1, up=16, n_up=4784, ma=48, outputs=("t_lo", "t_hi", "a_lo", "a_hi")):
    return {"outputs": list(outputs), "processors": {
        "wf_le": f"{M}.windower(wf, t_start, wf_le({n_win}, 'f'))",
        "curr": f"{M}.avg_current(wf_le, {ac}, curr({n_win - ac}, 'f'))",
        "curr_up": f"{M}.upsampler(curr, {up}, curr_up({n_up}, 'f'))",
        "curr_av": f"{M}.moving_window_multi(curr_up, {ma}, 3, 0, curr_av)",
        "t_lo, t_hi, a_lo, a_hi": f"{M}.min_max(curr_av, t_lo, t_hi, a_lo, a_hi)"}}


def _oracle(wf, t0, n_win, ac, up, n_up, ma):
    w, rc = oracle.windower(wf, t0, n_win)
    assert rc == 0
    c, rc = oracle.avg_current(w, ac)
    assert rc == 0
    u, rc = oracle.upsampler(c, up, n_up)
    assert rc == 0
    a, rc = oracle.moving_window_multi(u, ma, 3, 0)
    assert rc == 0
    *mm, rc = oracle.min_max(a)
    assert rc == 0
    return mm


def _rows(rng, n, L):
    i = np.arange(L, dtype=np.float64)[None, :]
    t0 = np.floor(rng.uniform(0.3, 0.6, (n, 1)) * L)
    rise = rng.uniform(3, 40, (n, 1))
    x = rng.uniform(500, 15000, (n, 1)) * (1 - np.exp(-np.clip(i - t0, 0, None) / rise)) * np.exp(-np.clip(i - t0, 0, None) / 30000.0)
    x += 5 * rng.standard_normal((n, L))
    return x.astype(np.float32), t0[:, 0]


def _run(recipe, tb, fused):
    from dspeed_amd.processing_chain import build_processing_chain

    chain, _, out = build_processing_chain(recipe, tb)
    chain._ensure()
    got = chain._chain.set_fused(1 if fused else 0)
    chain.execute()
    return chain, out, got


@pytest.mark.parametrize("cfg", [dict(), dict(n_win=200, ac=2, up=8, n_up=1568, ma=32), dict(n_win=150, ac=1, up=1, n_up=144, ma=16),
                                 dict(n_win=301, ac=1, up=16, n_up=4784, ma=112), dict(n_win=90, ac=3, up=4, n_up=336, ma=16)])
def test_current_branch_is_bit_exact(cfg):
    p = dict(n_win=301, ac=1, up=16, n_up=4784, ma=48)
    p.update(cfg)
    rng = np.random.default_rng(sum(p.values()))
    n, L = 333, 2048
    wf, t0 = _rows(rng, n, L)
    start = (t0 - rng.integers(20, 60, n)).astype(np.float32)
    start[5] += 0.37          # a fractional start truncates
    start[6] = -0.5           # int(-0.5) == 0: a window from sample 0
    start[7] = -1.0           # one sample before the waveform: NaN
    start[8] = L - p["n_win"]  # the last window that fits
    start[9] = L - p["n_win"] + 1  # one past: NaN
    start[10] = np.nan
    start[11] = 1e9
    wf[20, 100] = np.nan      # far from the window: still a NaN waveform
    wf[21, int(start[21]) + 50] = np.inf  # inf - inf inside the window
    wf[22, 0] = -np.inf       # an infinity outside the window changes nothing
    tb = {"wf": wf, "t_start": start}
    chain, out, fused = _run(_recipe(**p), tb, True)
    assert fused and chain._chain.kernel_name == "dsp_current_kernel"
    want = _oracle(wf, start, **p)
    for name, w in zip(("t_lo", "t_hi", "a_lo", "a_hi"), want):
        assert np.array_equal(out[name], w, equal_nan=True), (name, np.flatnonzero(~((out[name] == w) | (np.isnan(out[name]) & np.isnan(w))))[:10])
    assert np.isnan(out["a_hi"][[7, 9, 10, 11, 20, 21]]).all() and not np.isnan(out["a_hi"][[5, 6, 8, 22]]).any()
    # the same program on the waveform VM: its moving averages replay the rounding (not bit-exact), so values agree to the filter bar
    _c2, vm, fused2 = _run(_recipe(**p), tb, False)
    assert not fused2
    ok = ~np.isnan(want[3])
    assert np.array_equal(np.isnan(vm["a_hi"]), ~ok)
    assert np.all(np.abs(vm["a_hi"][ok] - want[3][ok]) <= 2e-6 * np.abs(want[3][ok]).max())


def test_constant_start_a_subset_of_outputs_and_many_rows():
    rng = np.random.default_rng(99)
    n, L = 9000, 1024  # more groups of 64 than resident wavefronts on a small grid are walked by the same wavefront
    wf, _t0 = _rows(rng, n, L)
    rec = _recipe(n_win=101, ac=1, up=16, n_up=1584, ma=48, outputs=("t_hi", "a_hi"))
    rec["processors"]["wf_le"] = f"{M}.windower(wf, 400, wf_le(101, 'f'))"
    chain, out, fused = _run(rec, {"wf": wf}, True)
    assert fused and chain._chain.kernel_name == "dsp_current_kernel" and sorted(out) == ["a_hi", "t_hi"]
    want = _oracle(wf, 400.0, 101, 1, 16, 1584, 48)
    assert np.array_equal(out["t_hi"], want[1]) and np.array_equal(out["a_hi"], want[3])


def test_shapes_the_kernel_does_not_take_run_on_the_vm():
    rng = np.random.default_rng(5)
    wf, t0 = _rows(rng, 64, 1024)
    start = (t0 - 30).astype(np.float32)
    for cfg in (dict(up=3, n_up=288, ma=16, n_win=100), dict(up=16, n_up=1584, ma=40, n_win=101), dict(up=16, n_up=1590, ma=48, n_win=101)):
        p = dict(n_win=301, ac=1, up=16, n_up=4784, ma=48)
        p.update(cfg)
        chain, out, fused = _run(_recipe(**p), {"wf": wf, "t_start": start}, True)
        assert "vm" in chain._chain.kernel_name
        want = _oracle(wf, start, **p)
        assert np.all(np.abs(out["a_hi"] - want[3]) <= 2e-6 * np.abs(want[3]).max()), cfg
