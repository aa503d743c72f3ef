"""BASELINE.json's full sizes on the device, checked through properties that do not need the oracle to chew through the whole batch:
  * partition invariance -- one launch over N rows == two launches over the halves, bit for bit (rows are independent; the row ->
    wavefront assignment differs between the two);
  * variant agreement    -- the default energy kernel, the classic one and the generic waveform VM agree within the filter tolerance
    on every row, and the classic kernel equals the VM bit for bit;
  * sampled parity       -- a strided sample of rows (and the very first / very last) is pulled to the host and compared with the oracle;
  * NaN containment      -- NaN poked into a few rows (waveform, baseline, pick-off time) makes exactly those outputs NaN;
  * shard consistency    -- a rank's 1.25 M-row shard of the 10 M-row batch is the same data and the same results as that slice of a
    larger launch (C4: event-axis sharding, no collective)."""
import numpy as np
import pytest

import oracle
import recipes

pytestmark = pytest.mark.gpu
TOL = 1e-6
TAU, RISE, FLAT, SIGMA, SEED = 1716.28, 625, 188, 5.0, 0xD5BEED


def _synth(rows, wf_len, dtype=np.float32, first_row=0, bl=(9000.0, 11000.0)):
    from dspeed_amd import _lib
    from dspeed_amd.device import DeviceArray, sync

    wf = DeviceArray((rows, wf_len), dtype)
    b, t = DeviceArray((rows,), np.float32), DeviceArray((rows,), np.float32)
    code = _lib.I16 if np.dtype(dtype) == np.int16 else _lib.F32
    _lib.check(_lib.lib().dsp_synth_waveforms(wf.ptr, code, rows, wf_len, wf_len, b.ptr, t.ptr, SEED, first_row, TAU, SIGMA,
                                              RISE + 0.8 * FLAT, bl[0], bl[1], 500.0, 15000.0, None), what="synth")
    sync()
    return wf, b, t


def _energy_chain(fused=1):
    from dspeed_amd.chain import Chain, energy_chain_program

    ch = Chain(energy_chain_program(4096, TAU, RISE, FLAT, "l"), "fullsize")
    ch.set_fused(fused)
    return ch


def _run_energy(ch, wf, bl, tp, lo, hi, out):
    bufs = {"waveform": wf.view_rows(lo, hi), "baseline": bl.view_rows(lo, hi), "t_pick": tp.view_rows(lo, hi), "trapEftp": out.view_rows(lo, hi)}
    ch.execute(bufs, hi - lo)
    ch.check()


def _rows(arr, idx):
    return np.stack([arr.view_rows(int(i), int(i) + 1).to_numpy()[0] for i in idx])


def test_c2_one_million_rows():
    from dspeed_amd.device import DeviceArray

    n = 1_000_000
    wf, bl, tp = _synth(n, 4096)
    ch = _energy_chain(1)
    assert ch.kernel_name == "dsp_energy_rr_kernel"
    out = DeviceArray((n,), np.float32)
    _run_energy(ch, wf, bl, tp, 0, n, out)
    whole = out.to_numpy()
    assert np.all(np.isfinite(whole))
    # partition invariance
    out2 = DeviceArray((n,), np.float32)
    cut = 437_911
    _run_energy(ch, wf, bl, tp, 0, cut, out2)
    _run_energy(ch, wf, bl, tp, cut, n, out2)
    assert np.array_equal(out2.to_numpy(), whole)
    # variant agreement
    res = {}
    for fused in (15, 0):
        c = _energy_chain(fused)
        o = DeviceArray((n,), np.float32)
        _run_energy(c, wf, bl, tp, 0, n, o)
        res[fused] = o.to_numpy()
    assert np.array_equal(res[15], res[0])  # classic kernel == VM, bit for bit
    assert np.max(np.abs(whole - res[0]) / np.abs(res[0])) <= TOL
    # sampled parity with the oracle
    idx = np.unique(np.concatenate([[0, 1, n - 2, n - 1], np.arange(0, n, 4099)]))
    w_s, b_s, t_s = _rows(wf, idx), _rows(bl, idx), _rows(tp, idx)
    want, rc = oracle.chain_energy(w_s, b_s, t_s, TAU, RISE, FLAT, "l")
    assert rc == 0
    assert np.max(np.abs(whole[idx] - want) / np.abs(want)) <= TOL
    # NaN containment
    nan = np.array([np.nan], dtype=np.float32)
    poke = {"wf": (123_456, 777), "bl": 654_321, "tp": 999_999}
    wf.view_rows(poke["wf"][0], poke["wf"][0] + 1).to_numpy()  # (row exists)
    from dspeed_amd import _lib

    L = _lib.lib()
    _lib.check(L.dsp_h2d(wf.ptr + (poke["wf"][0] * 4096 + poke["wf"][1]) * 4, nan.ctypes.data, 4))
    _lib.check(L.dsp_h2d(bl.ptr + poke["bl"] * 4, nan.ctypes.data, 4))
    _lib.check(L.dsp_h2d(tp.ptr + poke["tp"] * 4, nan.ctypes.data, 4))
    _run_energy(ch, wf, bl, tp, 0, n, out)
    got = out.to_numpy()
    bad = np.flatnonzero(np.isnan(got))
    assert sorted(bad.tolist()) == sorted([poke["wf"][0], poke["bl"], poke["tp"]])
    keep = np.ones(n, dtype=bool)
    keep[bad] = False
    assert np.array_equal(got[keep], whole[keep])


def test_c4_shard_of_the_ten_million_row_batch():
    """rank 3 of 8: its shard generated on its own equals rows [3.75 M, 5 M) of the global synthetic batch, and gives the same energies"""
    from dspeed_amd.device import DeviceArray
    from dspeed_amd.processing_chain import shard_rows

    total, world, rank = 10_000_000, 8, 3
    lo, hi = shard_rows(total, world, rank)
    assert (lo, hi) == (3_750_000, 5_000_000)
    n = hi - lo
    wf, bl, tp = _synth(n, 4096, first_row=lo)
    ch = _energy_chain(1)
    out = DeviceArray((n,), np.float32)
    _run_energy(ch, wf, bl, tp, 0, n, out)
    mine = out.to_numpy()
    # the neighbouring window generated with another origin overlaps this shard: same rows, same results
    off = 500_000
    wf2, bl2, tp2 = _synth(500_000, 4096, first_row=lo + off)
    out2 = DeviceArray((500_000,), np.float32)
    _run_energy(ch, wf2, bl2, tp2, 0, 500_000, out2)
    assert np.array_equal(out2.to_numpy(), mine[off:off + 500_000])
    assert np.array_equal(wf2.view_rows(17, 18).to_numpy(), wf.view_rows(off + 17, off + 18).to_numpy())


def test_c3_one_million_rows_long_fir():
    from dspeed_amd.device import DeviceArray
    from dspeed_amd.processing_chain import build_processing_chain

    n = 1_000_000
    wf, bl, _ = _synth(n, 8192)
    tb = {"waveform": wf, "baseline": bl}
    chain, _, _ = build_processing_chain(recipes.C3, tb)
    o1 = {"cuspEmax": DeviceArray((n,), np.float32), "zacEmax": DeviceArray((n,), np.float32)}
    chain.link(tb, o1)
    chain.execute()
    whole = {k: v.to_numpy() for k, v in o1.items()}
    o2 = {"cuspEmax": DeviceArray((n,), np.float32), "zacEmax": DeviceArray((n,), np.float32)}
    chain.link(tb, o2)
    chain.execute(0, 333_333)
    chain.execute(333_333, n)
    for k in whole:
        assert np.all(np.isfinite(whole[k]))
        assert np.array_equal(o2[k].to_numpy(), whole[k]), k
    # sampled parity: the oracle's FIR on a few rows (3.5 M multiply-adds each)
    idx = np.array([0, 1, 499_999, n - 1] + list(range(7, n, 99_991)))
    w_s, b_s = _rows(wf, idx), _rows(bl, idx)
    taps = {k: chain._consts[f"taps:{k}_kernel"] for k in ("cusp", "zac")}
    xb = oracle.bl_subtract(w_s, b_s)[0]
    for k in ("cusp", "zac"):
        conv, rc = oracle.convolve_wf(xb, taps[k], "v", 301, in_len=6092)
        assert rc == 0
        want = conv.max(axis=1)
        assert np.max(np.abs(whole[f"{k}Emax"][idx] - want) / np.max(np.abs(conv), axis=1)) <= TOL, k


def test_c5_one_million_int16_rows():
    from dspeed_amd.device import DeviceArray
    from dspeed_amd.processing_chain import build_processing_chain

    n = 1_000_000
    wf, _, _ = _synth(n, 8192, dtype=np.int16, bl=(-3000.0, 3000.0))
    thr = DeviceArray.from_numpy(np.full(n, 20.0, dtype=np.float32))
    tb = {"waveform": wf, "thr": thr}
    chain, _, _ = build_processing_chain(recipes.C5, tb)

    def outs():
        o = {k: DeviceArray((n,), np.float32) for k in ("tp_0", "tp_min", "tp_max", "wf_min", "wf_max")}
        o["dwt_haar"] = DeviceArray((n, 256), np.float32)
        return o

    o1 = outs()
    chain.link(tb, o1)
    chain.execute()
    o2 = outs()
    chain.link(tb, o2)
    chain.execute(0, 600_001)
    chain.execute(600_001, n)
    for k in ("tp_0", "tp_min", "tp_max", "wf_min", "wf_max"):
        assert np.array_equal(o1[k].to_numpy(), o2[k].to_numpy(), equal_nan=True), k
    idx = np.array([0, 1, n - 1] + list(range(11, n, 49_999)))
    assert np.array_equal(_rows(o1["dwt_haar"], idx), _rows(o2["dwt_haar"], idx))
    # sampled parity, stage by stage as the reference's ProcessingChain would run it.  The lane-per-waveform kernel evaluates every
    # recursion in the reference's own order, so all of it is bit-exact: extremes, their indices, the threshold time point, the coefficients
    assert chain._chain.kernel_name == "dsp_rows_kernel"
    w = _rows(wf, idx).astype(np.float32)
    dpz = oracle.double_pole_zero(w, 1716.28, 62.5, 0.02)[0]
    at = oracle.asym_trap_filter(dpz, 8, 4, 125)[0]
    tmin, tmax, amin, amax, rc = oracle.min_max(at)
    assert rc == 0
    for k, want in (("tp_min", tmin), ("tp_max", tmax), ("wf_min", amin), ("wf_max", amax)):
        assert np.array_equal(o1[k].to_numpy()[idx], want), k
    tp0, rc = oracle.time_point_thresh(at, np.float32(20.0), tmax, 0)
    assert rc == 0 and np.array_equal(o1["tp_0"].to_numpy()[idx], tp0, equal_nan=True) and np.isfinite(tp0).mean() > 0.2
    dwt = oracle.dwt_haar(dpz, 5, "a", 256)[0]
    assert np.array_equal(_rows(o1["dwt_haar"], idx), dwt)


def test_ge_recipe_on_the_measured_batch():
    """the whole Ge recipe on the batch its rate is quoted on (131 072 int16 rows of 8192 samples, device-resident): every kernel of the pass
    walks the batch in rounds of persistent workgroups whose count follows from the batch, so the same rows are run (a) in one launch, (b) in
    two uneven pieces, (c) a sample of them -- first, last, the rows either side of the pieces' seam, a stride through the rest -- as a batch of
    their own; all outputs bit for bit the same.  What the outputs ARE is checked against the oracle on small batches
    (test_gpu_icpc_recipe.py); this ties the full batch to those."""
    from dspeed_amd.device import DeviceArray
    from dspeed_amd.processing_chain import WaveformInput, build_processing_chain

    n, cut = 131_072, 70_001
    wf, bl, _ = _synth(n, 8192, dtype=np.int16, bl=(-3000.0, 3000.0))
    tb = {"waveform": WaveformInput(wf, 16.0, 48000.0), "baseline": bl}
    names = list(recipes.ICPC["outputs"])
    chain, _, _ = build_processing_chain(recipes.ICPC, tb)
    o1 = {k: DeviceArray((n,), np.float32) for k in names}
    chain.link(tb, o1)
    chain.execute()
    one = {k: o1[k].to_numpy() for k in names}
    o2 = {k: DeviceArray((n,), np.float32) for k in names}
    chain.link(tb, o2)
    chain.execute(0, cut)
    chain.execute(cut, n)
    for k in names:
        assert np.array_equal(one[k], o2[k].to_numpy(), equal_nan=True), k
    idx = np.array(sorted({0, 1, 3, 4, 63, 64, cut - 1, cut, n - 2, n - 1} | set(range(17, n, 1021))))
    small = {"waveform": WaveformInput(DeviceArray.from_numpy(_rows(wf, idx)), 16.0, 48000.0), "baseline": DeviceArray.from_numpy(bl.to_numpy()[idx])}
    chain_s, _, _ = build_processing_chain(recipes.ICPC, small)
    o3 = {k: DeviceArray((len(idx),), np.float32) for k in names}
    chain_s.link(small, o3)
    chain_s.execute()
    for k in names:
        assert np.array_equal(one[k][idx], o3[k].to_numpy(), equal_nan=True), k
    # the batch is pulses (one-sample steps: most rise-time walks find no crossing and say NaN): the energies are there, and where two
    # rise-time points exist they are ordered
    assert np.isfinite(one["trapEmax"]).all() and (one["trapEmax"] > 100.0).mean() > 0.9
    ok = np.isfinite(one["tp_10"]) & np.isfinite(one["tp_90"])
    assert (one["tp_10"][ok] <= one["tp_90"][ok]).all()
