"""Edge shapes of the gufunc and chain entry points: empty batches, a single 1-D waveform, ragged lengths (not a multiple of the
wavefront), the longest waveform one wavefront can hold and the first length it cannot, the shortest waveforms."""
import os

import numpy as np
import pytest

import oracle
from golden_util import assert_rel_to_peak

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    from dspeed_amd import processors

    return processors


def test_empty_batch_is_a_no_op(P):
    w = np.zeros((0, 1024), dtype=np.float32)
    assert P.trap_filter(w, 10, 5).shape == (0, 1024)
    assert P.fixed_time_pickoff(w, np.zeros(0, dtype=np.float32), ord("l")).shape == (0,)
    from dspeed_amd.processing_chain import build_processing_chain
    import recipes

    chain, _, out = build_processing_chain(recipes.C1, {"waveform": w})
    chain.execute()
    assert out["wf_trap"].shape == (0, 1024)


def test_one_dimensional_waveform_in_one_dimensional_result_out(P):
    rng = np.random.default_rng(2)
    w = (1000 * rng.standard_normal(777)).astype(np.float32)
    got = P.pole_zero(w, 50.0)
    assert got.shape == (777,)
    assert_rel_to_peak(got[None, :], oracle.pole_zero(w[None, :], 50.0)[0], 1e-6, "pz 1-D")
    tmin, tmax, amin, amax = P.min_max(w)
    assert (tmin, tmax) == (float(np.argmin(w)), float(np.argmax(w))) and amin == w.min() and amax == w.max()


@pytest.mark.parametrize("wf_len", [1, 2, 5, 63, 64, 65, 127, 129, 1000, 4097])
def test_ragged_lengths(P, wf_len):
    rng = np.random.default_rng(wf_len)
    w = (100 * rng.standard_normal((9, wf_len))).astype(np.float32)
    assert np.array_equal(P.bl_subtract(w, 3.5), oracle.bl_subtract(w, 3.5)[0])
    mm = P.min_max(w)
    ref = oracle.min_max(w)
    for g, r in zip(mm, ref[:4]):
        assert np.array_equal(g, r)
    assert_rel_to_peak(P.pole_zero(w, 20.0), oracle.pole_zero(w, 20.0)[0], 1e-6, "pz ragged")
    if wf_len >= 5:
        assert_rel_to_peak(P.trap_filter(w, 2, 1), oracle.trap_filter(w, 2, 1)[0], 1e-6, "trap ragged")


def test_longest_waveform_and_the_first_that_does_not_fit(P):
    """one wavefront holds a waveform in LDS (160 KB per CU): 32768 float32 samples fit for the in-place filters and the
    waveform -> scalar processors (about 19 k for filters that keep source and destination side by side); 65536 do not -> a clear error"""
    rng = np.random.default_rng(4)
    w = (100 * rng.standard_normal((3, 32768))).astype(np.float32)
    assert np.array_equal(P.bl_subtract(w, 1.0), w - np.float32(1.0))
    t = np.array([0.0, 16000.5, 32767.0], dtype=np.float32)
    assert np.array_equal(P.fixed_time_pickoff(w, t, ord("l")), oracle.fixed_time_pickoff(w, t, "l")[0])
    assert_rel_to_peak(P.pole_zero(w, 500.0), oracle.pole_zero(w, 500.0)[0], 1e-6, "pz 32768")
    w16 = w[:, :16384].copy()
    assert_rel_to_peak(P.trap_filter(w16, 100, 30), oracle.trap_filter(w16, 100, 30)[0], 1e-6, "trap 16384")
    with pytest.raises(ValueError, match="LDS"):
        P.bl_subtract(np.zeros((1, 65536), dtype=np.float32), 1.0)
    with pytest.raises(ValueError, match="LDS"):
        P.trap_filter(w, 100, 30)


def test_shortest_waveforms_follow_the_reference(P):
    from dspeed_amd.errors import DSPFatal

    w3 = np.ones((2, 3), dtype=np.float32)
    with pytest.raises(DSPFatal):  # pole_zero.py:163-166: double_pole_zero needs more than 3 samples
        P.double_pole_zero(w3, 10.0, 5.0, 0.1)
    with pytest.raises(DSPFatal):  # trap_filters.py:59-60: 2*rise + flat > len
        P.trap_filter(w3, 2, 0)
    assert np.array_equal(P.pole_zero(np.ones((1, 1), dtype=np.float32), 10.0), np.ones((1, 1), dtype=np.float32))


@pytest.mark.parametrize("wf_len,m", [(8192, 133), (4096, 133), (4096, 16), (8192, 300), (6092, 133), (2048, 1000)])
def test_fir_edges_same_and_full_modes(wf_len, m):
    """'same' and 'full' convolutions reach past both ends of the waveform, where the reference pads with zeros (np.convolve,
    convolutions.py:72): the blocked FIR path reads those zeros from the slot's guard and tail instead of checking bounds.  Compared
    sample by sample with the oracle -- the edges are the point --, as a processor (input alone in its slot: linear layout) and inside
    a recipe where the pole-zero filter shares the input (chunk-padded layout); an infinite tap must not meet a padding zero."""
    import oracle
    from dspeed_amd import processors as P
    from dspeed_amd.processing_chain import build_processing_chain

    rng = np.random.default_rng(wf_len + m)
    n_wf = 12
    w = (1000 + 100 * rng.standard_normal((n_wf, wf_len))).astype(np.float32)
    k = rng.standard_normal(m).astype(np.float32)
    for mode, out_len in (("s", wf_len), ("f", wf_len + m - 1), ("v", wf_len - m + 1)):
        want = oracle.convolve_wf(w, k, mode, out_len)[0]
        got = P.convolve_wf(w, k, ord(mode), np.empty((n_wf, out_len), dtype=np.float32))
        # (random zero-mean taps cancel: the error bound of a dot product is relative to sum |k| * max |w|, not to the output)
        scale = np.sum(np.abs(k)) * np.max(np.abs(w), axis=1, keepdims=True)
        assert np.max(np.abs(got - want) / scale) <= 2e-7, (mode, "processor")
    # inside a chain: the FIR input is the pole-zero output (padded layout), 'same' mode, output stored
    rec = {"outputs": ["wf_f", "wf_pz"], "processors": {
        "wf_pz": "dspeed.processors.pole_zero(waveform, 500.5, wf_pz)",
        "kern": {"function": "t0_filter", "module": "dspeed.processors", "args": [8, m - 8, f"kern({m}, 'f')"]},
        "wf_f": {"function": "convolve_wf", "module": "dspeed.processors", "args": ["wf_pz", "kern", "'s'", f"wf_f({wf_len}, 'f')"]}}}
    chain, _, out = build_processing_chain(rec, {"waveform": w})
    chain.execute()
    kern = np.zeros(m, dtype=np.float32)
    P.t0_filter(8, m - 8, kern)
    pz = oracle.pole_zero(w, 500.5)[0]
    want = oracle.convolve_wf(pz, kern, "s", wf_len)[0]
    scale = np.max(np.abs(want), axis=1, keepdims=True)
    assert np.max(np.abs(out["wf_f"] - want) / scale) <= 2e-6, "recipe, padded input"
    # the recipe's tap binding holds zeros up to a multiple of 16 taps (the blocked loop then covers every tap): an infinite SAMPLE must
    # not meet one of them (0 * inf) -- the op falls back to the true length for such a waveform
    w_inf = w.copy()
    w_inf[1, wf_len // 3], w_inf[2, 5] = np.inf, -np.inf
    rec2 = {"outputs": ["wf_f"], "processors": {"kern": rec["processors"]["kern"], "wf_c": "waveform + 0",
                                               "wf_f": {"function": "convolve_wf", "module": "dspeed.processors",
                                                        "args": ["wf_c", "kern", "'s'", f"wf_f({wf_len}, 'f')"]}}}
    chain, _, out = build_processing_chain(rec2, {"waveform": w_inf})
    chain.execute()
    want = oracle.convolve_wf(w_inf, kern, "s", wf_len)[0]
    assert np.array_equal(np.isnan(out["wf_f"]), np.isnan(want)) and np.array_equal(np.isinf(out["wf_f"]), np.isinf(want))
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(out["wf_f"]), fin)
    scale = np.max(np.abs(np.where(fin, want, 0)), axis=1, keepdims=True)
    with np.errstate(invalid="ignore"):
        assert np.max(np.abs(np.where(fin, out["wf_f"] - want, 0)) / scale) <= 2e-6, "recipe, infinite samples"
    # an infinite tap: no 0 * inf from the padding
    k2 = k.copy()
    k2[m // 2] = np.inf
    want = oracle.convolve_wf(w, k2, "s", wf_len)[0]
    got = P.convolve_wf(w, k2, ord("s"), np.empty((n_wf, wf_len), dtype=np.float32))
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.isinf(got), np.isinf(want))


@pytest.mark.skipif(os.environ.get("DSPEED_TEST_PIN_IN_PLACE", "0") != "1",
                    reason="in-place page-locking of NumPy memory is opt-in (profiles/design_diary_r01_r03.md: Host memory and the runtime); set DSPEED_TEST_PIN_IN_PLACE=1")
def test_page_locked_ranges_never_overlap():
    """HostPin: one record per range at the runtime -- the same array twice shares it, a range that shares pages with a live one is
    refused (the chain then copies it unpinned), and everything can be locked again once released."""
    from dspeed_amd.device import _PINNED, HostPin

    a = np.zeros((512, 4096), dtype=np.float32)
    before = dict(_PINNED)
    p1 = HostPin(a)
    p2 = HostPin(a)
    assert _PINNED[(a.ctypes.data, a.nbytes)] == 2
    for view in (a[:100], a[100:200], a[511:]):
        with pytest.raises(RuntimeError):
            HostPin(view)
    p1.close()
    with pytest.raises(RuntimeError):
        HostPin(a[:100])
    p2.close()
    assert dict(_PINNED) == before
    p3 = HostPin(a[:100])
    with pytest.raises(RuntimeError):
        HostPin(a)
    p4 = HostPin(a[300:400])  # (no page in common with rows 0..99)
    p3.close()
    p4.close()
    assert dict(_PINNED) == before


@pytest.mark.skipif(os.environ.get("DSPEED_TEST_PIN_IN_PLACE", "0") != "1",
                    reason="in-place page-locking of NumPy memory is opt-in (profiles/design_diary_r01_r03.md: Host memory and the runtime); set DSPEED_TEST_PIN_IN_PLACE=1")
def test_columns_page_locked_in_place_give_the_same_results():
    import recipes
    from dspeed_amd.processing_chain import build_processing_chain

    rng = np.random.default_rng(21)
    n = 600
    wf = (10000 + 50 * rng.standard_normal((n, 4096))).astype(np.float32)
    tb = {"waveform": wf, "baseline": np.full(n, 10000, np.float32), "t_pick": np.full(n, 2823.4, np.float32)}
    staged, _, o1 = build_processing_chain(recipes.C2, tb)
    staged.pipeline_bytes = 100 * 4096 * 4
    staged.execute()
    ref = o1["trapEftp"].copy()
    pinned, _, o2 = build_processing_chain(recipes.C2, tb)
    pinned.pin_in_place = True
    pinned.pipeline_bytes = 100 * 4096 * 4
    pinned.execute()
    assert np.array_equal(o2["trapEftp"], ref) and len(pinned._pins) >= 1
