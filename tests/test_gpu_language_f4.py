"""What closed SURVEY 8(f4) in round 4, on the device against NumPy's own ufuncs (the reference adds the NumPy ufunc as a processor for every
operator of the language, processing_chain.py:832-947, and picks its loop by ``np.can_cast`` on the variables' types, :1565-1572):

* truth values alone select NumPy's '??' loops -- ``+`` is logical or, ``*`` logical and, ``//`` the int8 loop, ``-`` an error;
* 64-bit integer loops ('ll->l', 'QQ->Q': int64 / uint64 columns, int32 beside uint32) between per-event values run in an *integer
  program* (64-bit integer registers: wrap-around at 2^63 / 2^64 bit for bit), on waveforms in the float64 chain where the operands' types
  bound the result below 2^53;
* a per-event variable times / over a time (``t * (2*us)``: the time counts periods of the variable's grid, :1747-1764);
* a processor's INTEGER parameter given as a per-event column (``trap_filter(wf, rise_column, ...)``, :1702-1745): rows grouped by value;
* a slice whose bound is a variable is the reference's ProcessingChainError (:1016-1022)."""
import numpy as np
import pytest

import oracle
from dspeed_amd import build_dsp, build_processing_chain
from dspeed_amd.errors import DSPFatal, ProcessingChainError
from dspeed_amd.processing_chain import GroupedProcessingChain, WaveformInput

pytestmark = pytest.mark.gpu
M = "dspeed.processors"


def _run(processors, outputs, tb):
    chain, _, out = build_processing_chain({"outputs": outputs, "processors": processors}, tb)
    chain.execute()
    return chain, out


def test_truth_values_alone_run_numpys_logical_loops():
    rng = np.random.default_rng(5)
    n = 70
    wf = rng.normal(1000, 300, (n, 200)).astype(np.float32)
    tb = {"waveform": wf, "b1": rng.random(n) < 0.5, "b2": rng.random(n) < 0.5, "ev": np.arange(n, dtype=np.int32)}
    procs = {"lo": "waveform > 900", "hi": "waveform < 1200", "band": "lo * hi", "any": "lo + hi", "first": "ev == 0", "lo_or_first": "lo + first",
             "o": "b1 + b2", "a": "b1 * b2", "q": "b1 // b2", "o1": "b1 + 1", "a0": "b1 * 0", "o0": "b1 + 0", "mixed": "b1 + ev", "band_q": "lo // hi"}
    outs = ["band", "any", "lo_or_first", "o", "a", "q", "o1", "a0", "o0", "mixed", "band_q"]
    _, out = _run(procs, outs, tb)
    lo, hi, b1, b2 = wf > 900, wf < 1200, tb["b1"], tb["b2"]
    with np.errstate(all="ignore"):
        want = {"band": np.multiply(lo, hi), "any": np.add(lo, hi), "lo_or_first": np.add(lo, (tb["ev"] == 0)[:, None]), "o": np.add(b1, b2), "a": np.multiply(b1, b2),
                "q": np.floor_divide(b1, b2), "o1": np.add(b1, np.bool_(1)), "a0": np.multiply(b1, np.bool_(0)), "o0": np.add(b1, np.bool_(0)),
                "mixed": np.add(b1, tb["ev"]), "band_q": np.floor_divide(lo, hi)}
    for k, w in want.items():
        assert out[k].dtype == w.dtype and np.array_equal(out[k], w), (k, out[k].dtype, w.dtype)
    assert want["band"].dtype == np.bool_ and want["q"].dtype == np.int8 and want["mixed"].dtype == np.int32  # (what NumPy's loops return)
    for bad in ("b1 - b2", "-b1", "lo - hi", "-lo"):  # numpy.subtract / numpy.negative refuse truth values
        with pytest.raises(ProcessingChainError, match="numpy boolean"):
            build_processing_chain({"outputs": ["x"], "processors": dict(procs, x=bad)}, tb)


def _wide_table(n=96, seed=17):
    rng = np.random.default_rng(seed)
    big = np.iinfo(np.int64)
    q = rng.integers(big.min, big.max, n, dtype=np.int64)
    q[:6] = [big.min, big.max, -1, 0, 1, big.min + 1]
    u = rng.integers(0, np.iinfo(np.uint64).max, n, dtype=np.uint64)
    u[:4] = [0, np.iinfo(np.uint64).max, 1, 2 ** 63]
    return {"q": q, "r": rng.integers(-2 ** 40, 2 ** 40, n, dtype=np.int64), "u": u, "v": rng.integers(0, 2 ** 33, n, dtype=np.uint64),
            "i": rng.integers(-2 ** 31, 2 ** 31, n, dtype=np.int32), "w": rng.integers(0, 2 ** 32, n, dtype=np.uint32), "h": rng.integers(-300, 300, n).astype(np.int16),
            "flag": rng.random(n) < 0.5, "waveform": rng.normal(0, 100, (n, 64)).astype(np.float32)}


def test_64_bit_integer_loops_wrap_bit_for_bit():
    tb = _wide_table()
    q, r, u, v, i, w, h = (tb[k] for k in "qruviwh")
    procs = {"a": "q + 1", "b": "q * r", "c": "q - r", "d": "q // h", "e": "-q", "f": "i + w", "g": "i * w", "k": "u * 3", "l": "u // v", "m": "u + v", "n": "-u",
             "o": "q // 0", "p": "r // -1", "s": "i // w", "t": "w - i", "lt": "q < r", "ge": "u >= v", "eq": "q == -1", "big": f"r > {2 ** 39}",
             "sel": "where(lt, q, r)", "selc": "where(flag, u, 7)", "n16": "astype(q, 'int16')", "n32": "astype(r, 'uint32')", "asu": "astype(q, 'uint64')",
             "asq": "astype(u, 'int64')", "tv": "astype(r, '?')", "chain": "(q + r) * 3 - i", "up": "astype(h, 'int64') * r"}
    _, out = _run(procs, list(procs), tb)
    i64 = np.int64
    with np.errstate(all="ignore"):
        want = {"a": q + i64(1), "b": q * r, "c": q - r, "d": q // h, "e": -q, "f": i + w, "g": i * w, "k": u * np.uint64(3), "l": u // v, "m": u + v, "n": -u,
                "o": q // i64(0), "p": r // i64(-1), "s": i // w, "t": w - i, "lt": q < r, "ge": u >= v, "eq": q == i64(-1), "big": r > i64(2 ** 39),
                "sel": np.where(q < r, q, r), "selc": np.where(tb["flag"], u, np.uint64(7)), "n16": q.astype(np.int16), "n32": r.astype(np.uint32),
                "asu": q.astype(np.uint64), "asq": u.astype(np.int64), "tv": r != 0, "chain": (q + r) * i64(3) - i, "up": h.astype(np.int64) * r}
    for k, x in want.items():
        assert out[k].dtype == x.dtype and np.array_equal(out[k], x), (k, procs[k], out[k].dtype, x.dtype)
    assert want["f"].dtype == np.int64 and want["k"].dtype == np.uint64  # (int32 beside uint32 IS the int64 loop)
    assert (want["b"] != (q.astype(object) * r.astype(object))).any() and (want["k"] != u.astype(object) * 3).any()  # the inputs do wrap
    for bad, exc in (("q + u", NotImplementedError), ("astype(waveform, 'int64')", NotImplementedError)):  # (NumPy's loop for int64 beside uint64 is the float64 one)
        with pytest.raises(exc):
            build_processing_chain({"outputs": ["x"], "processors": {"x": bad}}, tb)


def test_random_64_bit_expressions_against_numpy():
    """seeded random expression trees over int64 / uint64 / int32 / uint32 columns: NumPy's own integer ufuncs on wrapping inputs are the oracle,
    bit for bit and dtype for dtype (the scheme of test_gpu_expressions.py::test_random_integer_expressions_against_numpy)"""
    for names, seed in ((("q", "r"), 1), (("u", "v"), 2), (("i", "w"), 3), (("q", "i"), 4), (("v", "w"), 5)):
        rng = np.random.default_rng(100 + seed)
        tb = _wide_table(n=64, seed=seed)

        def tree(depth):
            if depth == 0 or rng.random() < 0.25:
                return str(rng.choice([names[0], names[1], str(int(rng.integers(1, 9)))]))
            op = rng.choice(["+", "-", "*", "//", "neg"])
            if op == "neg":
                return f"(-{tree(depth - 1)})"
            l, r = tree(depth - 1), tree(depth - 1)
            if l.isdigit() and r.isdigit():
                l = names[0]
            return f"({l} {op} {r})"

        procs, want = {}, {}
        env = {nm: tb[nm] for nm in names}
        k = 0
        while k < 14:
            e = tree(3)
            if not any(nm in e for nm in names) or e.isidentifier():
                continue
            try:
                with np.errstate(all="ignore"):
                    val = np.asarray(eval(e, {"__builtins__": {}}, env))
            except OverflowError:
                continue
            if val.dtype.kind == "f":  # (a mix NumPy sends to its float64 loop: refused by name, covered above)
                continue
            procs[f"x{k}"], want[f"x{k}"] = e, val
            k += 1
        _, out = _run(procs, list(procs), tb)
        for k, w in want.items():
            assert out[k].dtype == w.dtype and np.array_equal(out[k], w), (names, k, procs[k], out[k].dtype, w.dtype)


def test_integer_program_results_feed_the_processors():
    """what the integer program computed reaches the waveform processors as a column: a threshold scaled in integers, a start index"""
    rng = np.random.default_rng(3)
    n = 40
    wf = np.cumsum(rng.normal(0.5, 1.0, (n, 512)), axis=1).astype(np.float32)
    tb = {"waveform": wf, "start": rng.integers(100, 400, n).astype(np.int64), "k": rng.integers(1, 5, n).astype(np.int32), "big": np.full(n, 2 ** 40, np.int64)}
    procs = {"t_from": "start + k", "thr": "(big // 1099511627776) * 20",  # (2^40 // 2^40 = 1, then the int64 loop's 20)
             "tp": {"function": "time_point_thresh", "module": M, "args": ["waveform", "thr", "t_from", 1, "tp"]}}
    _, out = _run(procs, ["tp", "t_from", "thr"], tb)
    t_from = tb["start"] + tb["k"]
    assert out["t_from"].dtype == np.int64 and np.array_equal(out["t_from"], t_from) and np.array_equal(out["thr"], np.full(n, 20, np.int64))
    want = np.empty(n, np.float32)
    for r in range(n):
        want[r] = oracle.time_point_thresh(wf[r:r + 1], np.float32(20), np.float32(t_from[r]), np.float32(1))[0][0]
    assert np.array_equal(out["tp"], want, equal_nan=True)


def test_wide_loops_on_waveforms_are_exact_in_the_float64_chain():
    rng = np.random.default_rng(9)
    n = 24
    a = rng.integers(-2 ** 31, 2 ** 31, (n, 96), dtype=np.int32)
    b = rng.integers(0, 2 ** 32, (n, 96), dtype=np.uint32)
    a[0, :4] = [-2 ** 31, 2 ** 31 - 1, -1, 0]
    b[0, :4] = [2 ** 32 - 1, 2 ** 32 - 1, 0, 1]
    tb = {"a": a, "b": b, "sc": rng.integers(0, 2 ** 32, n, dtype=np.uint32), "h": rng.integers(1, 1000, (n, 96)).astype(np.int16)}
    procs = {"x": "a + b", "y": "a - b", "z": "b - a", "q": "a // b", "t": "a + sc", "m": "(a + b) * h", "u": "(a + b) // h", "neg": "-(a + b)"}
    chain, out = _run(procs, list(procs), tb)
    assert chain.loop_dtype == np.float64
    with np.errstate(all="ignore"):
        want = {"x": a + b, "y": a - b, "z": b - a, "q": a // b, "t": a + tb["sc"][:, None], "m": (a + b) * tb["h"], "u": (a + b) // tb["h"], "neg": -(a + b)}
    for k, w in want.items():
        assert w.dtype == np.int64 and out[k].dtype == np.int64 and np.array_equal(out[k], w), k
    with pytest.raises(NotImplementedError, match="2\\^53"):  # 32 + 32 bits: the product can exceed what a float64 holds
        build_processing_chain({"outputs": ["x"], "processors": {"x": "a * b"}}, tb)


def test_a_variable_times_a_time_counts_periods_of_its_grid():
    rng = np.random.default_rng(21)
    n = 30
    wf = (1000 + 50 * rng.standard_normal((n, 1000))).astype(np.float32)
    wf[:, 500:] += 900
    tb = {"waveform": WaveformInput(wf, 16.0, 0.0)}
    mm = {"t_a, t_b, lo, hi": {"function": "min_max", "module": M, "args": ["waveform", "t_a", "t_b", "lo", "hi"], "unit": ["ns", "ns", "ADC", "ADC"]}}
    procs = dict(mm, x="t_b * (32*ns)", y="t_b / (8*ns)", z=f"{M}.fixed_time_pickoff(waveform, t_a * (4*ns) + 5, 'n', z)")
    _, out = _run(procs, ["x", "y", "z", "t_b", "t_a"], tb)
    t_a, t_b = np.argmin(wf, axis=1).astype(np.float32), np.argmax(wf, axis=1).astype(np.float32)
    # a coordinate holds samples of its grid; 32 ns are 2 periods of 16 ns and 8 ns half a period; the column leaves in ns (x 16)
    assert np.array_equal(out["t_b"], t_b * 16) and np.array_equal(out["x"], (t_b * np.float32(2.0)) * 16) and np.array_equal(out["y"], (t_b / np.float32(0.5)) * 16)
    want = np.array([wf[r, int(np.rint(t_a[r] * np.float32(0.25) + np.float32(5)))] for r in range(n)], np.float32)
    pick = np.array([oracle.fixed_time_pickoff(wf[r:r + 1], np.float32(t_a[r] * np.float32(0.25) + np.float32(5)), "n")[0][0] for r in range(n)], np.float32)
    assert np.array_equal(out["z"], pick) and want.shape == pick.shape


def test_integer_parameters_given_per_event_group_the_rows():
    """trap_filter(wf, rise_column, flat_column, out): the reference broadcasts the columns into the gufunc's "()" slots (:1702-1745); here the
    rows are grouped by value.  Oracle: the C restatement row by row with each row's own parameters."""
    rng = np.random.default_rng(8)
    n = 90
    wf = np.cumsum(rng.normal(0, 3, (n, 512)), axis=1).astype(np.float32)
    rise = rng.choice([4, 8, 16, 40], n).astype(np.int32)
    flat = rng.choice([2, 10], n).astype(np.int16)
    level = rng.choice([1, 2, 3], n).astype(np.uint16)
    tb = {"waveform": wf, "rise": rise, "flat": flat, "level": level, "baseline": rng.normal(0, 1, n).astype(np.float32)}
    procs = {"wf_blsub": f"{M}.bl_subtract(waveform, baseline, wf_blsub)",
             "wf_trap": {"function": "trap_norm", "module": M, "args": ["wf_blsub", "rise", "flat", "wf_trap"]},
             "e": {"function": "fixed_time_pickoff", "module": M, "args": ["wf_trap", "300", "'i'", "e"]},
             "tp": {"function": "trap_pickoff", "module": M, "args": ["wf_blsub", "rise", "flat", "300", "tp"]}}
    chain, mask, out = build_processing_chain({"outputs": ["wf_trap", "e", "tp"], "processors": procs}, tb)
    assert isinstance(chain, GroupedProcessingChain) and chain.group_columns == ["rise", "flat"] and {"rise", "flat"} <= set(mask)
    chain.execute()
    assert len(chain._group_chains) == len({(r, f) for r, f in zip(rise, flat)}) == 8
    bl = wf - tb["baseline"][:, None]
    for r in range(n):
        t, rc = oracle.trap_norm(bl[r:r + 1], int(rise[r]), int(flat[r]))
        assert rc == 0
        peak = np.abs(t).max()
        assert np.abs(out["wf_trap"][r] - t[0]).max() <= 1e-6 * peak, r
        assert abs(out["e"][r] - t[0, 300]) <= 1e-6 * peak
        tpo, rc = oracle.trap_pickoff(bl[r:r + 1], int(rise[r]), int(flat[r]), np.float32(300))
        assert rc == 0 and abs(out["tp"][r] - tpo[0]) <= 1e-6 * max(abs(tpo[0]), 1.0)
    # a second pass over other rows reuses the groups' chains; through build_dsp the same
    tb2 = {k: v[::-1].copy() for k, v in tb.items()}
    out2 = chain(tb2, {k: np.empty_like(v) for k, v in out.items()})
    assert len(chain._group_chains) == 8 and np.array_equal(out2["e"], out["e"][::-1])
    dsp = build_dsp(tb, dsp_config={"outputs": ["e", "tp"], "processors": procs})
    assert np.array_equal(dsp["e"], out["e"]) and np.array_equal(dsp["tp"], out["tp"])
    # a DSPFatal names the first row, in table order, whose parameters the reference refuses (trap_filters.py:53-60)
    bad = dict(tb, rise=rise.copy())
    bad["rise"][[17, 60]] = -3
    chain_b, _, _ = build_processing_chain({"outputs": ["e"], "processors": procs}, bad)
    with pytest.raises(DSPFatal, match="rise section must be positive") as ei:
        chain_b.execute()
    assert ei.value.wf_range == range(17, 18)
    # a float column in an integer slot matches no signature of the gufunc (:1565-1572); an integer computed inside the recipe is refused by name
    with pytest.raises(ProcessingChainError, match="type signature"):
        build_processing_chain({"outputs": ["e"], "processors": dict(procs, wf_trap={"function": "trap_norm", "module": M, "args": ["wf_blsub", "baseline", "flat", "wf_trap"]})}, tb)
    with pytest.raises(NotImplementedError, match="computed per event"):
        build_processing_chain({"outputs": ["e"], "processors": dict(procs, wf_trap={"function": "trap_norm", "module": M, "args": ["wf_blsub", "rise + 1", "flat", "wf_trap"]})}, tb)


def test_a_variable_slice_bound_is_the_references_error():
    tb = {"waveform": np.zeros((4, 100), np.float32), "k": np.arange(4, dtype=np.int32)}
    for expr in ("waveform[k:k+10]", "waveform[:k]", "waveform[0:50:k]"):
        with pytest.raises(ProcessingChainError, match="Slice values must be constants"):
            build_processing_chain({"outputs": ["x"], "processors": {"x": expr}}, tb)


def test_rounding_floor_division_and_numpy_ufuncs_on_waveforms():
    """round / floor / ceil / trunc(wf, to_nearest) are the reference's rounding ufuncs sample by sample (processors/round_to_nearest.py:
    to_nearest * f(val / to_nearest) in the loop's type); ``//`` between floats is numpy.floor_divide's float loop (its quotient comes from
    fmod, not from floor(a / b)); numpy.multiply / numpy.divide with waveform operands written as processors are the operators"""
    rng = np.random.default_rng(31)
    n = 40
    wf = rng.normal(0, 50, (n, 300)).astype(np.float32)
    wf[0, :6] = [2.5, 3.5, -2.5, -0.5, 0.5, np.nan]
    tb = {"waveform": wf, "g": rng.uniform(0.5, 4.0, n).astype(np.float32), "d": rng.choice([0.1, 0.3, 7.0, -2.5], n).astype(np.float32)}
    tb["g"][:5] = [0.5, 0.7, 1.0, 1.4, 1.5]  # (multiples of float32(0.1), which lies above 0.1: the true quotients are just below 5, 7, 10 ...)
    procs = {"r1": "round(waveform)", "r5": "round(waveform, 5)", "f2": "floor(waveform, 2)", "c3": "ceil(waveform, 0.5)", "t4": "trunc(waveform, 4)",
             "fd": "waveform // d", "fd3": "waveform // 0.3", "gd": "g // d", "g01": "g // 0.1",
             "scaled": {"function": "multiply", "module": "numpy", "args": ["waveform", "g", "scaled"]},
             "ratio": {"function": "divide", "module": "numpy", "args": ["waveform", "scaled", "ratio"]},
             "back": f"{M}.bl_subtract(scaled, g, back)"}
    _, out = _run(procs, list(procs), tb)
    f = np.float32
    with np.errstate(all="ignore"):
        want = {"r1": np.rint(wf), "r5": f(5) * np.rint(wf / f(5)), "f2": f(2) * np.floor(wf / f(2)), "c3": f(0.5) * np.ceil(wf / f(0.5)),
                "t4": f(4) * np.trunc(wf / f(4)), "fd": np.floor_divide(wf, tb["d"][:, None]), "fd3": np.floor_divide(wf, f(0.3)),
                "gd": np.floor_divide(tb["g"], tb["d"]), "g01": np.floor_divide(tb["g"], f(0.1)), "scaled": wf * tb["g"][:, None]}
        want["ratio"] = wf / want["scaled"]
        want["back"] = want["scaled"] - tb["g"][:, None]
        want["back"][np.isnan(want["scaled"]).any(axis=1)] = np.nan  # (bl_subtract's NaN rule)
    for k, w in want.items():
        assert out[k].dtype == np.float32 and np.array_equal(out[k], w.astype(np.float32), equal_nan=True), k
    # floor(a / b) and numpy.floor_divide do differ on such inputs: the quotient rounds up to an integer the true one never reaches
    assert np.array_equal(want["g01"][:5], [4, 6, 9, 13, 14]) and np.array_equal(np.floor(tb["g"][:5] / f(0.1)), [5, 7, 10, 14, 15])
    for bad in ("waveform % 2", "waveform ** 2", "g % 2"):  # not in the reference's operator table (processing_chain.py:46-59)
        with pytest.raises(ProcessingChainError):
            build_processing_chain({"outputs": ["x"], "processors": {"x": bad}}, tb)


def test_the_round_4_additions_through_the_table_loop_and_the_containers():
    """integer programs, grouped chains and the device-resident fast path behind the entry points a dspeed user calls: build_dsp over arrays, an LGDO
    table in memory and a chunk iterator; ProcessingChain.execute(wait=False) on device-resident columns"""
    from lgdo_standins import Array, LH5Iterator, Table, WaveformTable

    from dspeed_amd.device import DeviceArray

    rng = np.random.default_rng(41)
    n = 700
    wf = np.cumsum(rng.normal(0, 3, (n, 256)), axis=1).astype(np.float32)
    ev = (np.arange(n, dtype=np.int64) * 7 + 2 ** 45)
    rise = rng.choice([4, 8, 16], n).astype(np.int32)
    recipe = {"outputs": ["ev2", "odd", "e", "hi"], "processors": {
        "ev2": "eventnumber * 3 + 1", "odd": "(eventnumber // 7) - (eventnumber // 14) * 2 == 1", "hi": "astype(eventnumber // 1024, 'uint32')",
        "wf_trap": {"function": "trap_norm", "module": M, "args": ["waveform", "rise", "2", "wf_trap"]},
        "e": {"function": "fixed_time_pickoff", "module": M, "args": ["wf_trap", "200", "'i'", "e"]}}}
    want_e = np.array([oracle.trap_norm(wf[r:r + 1], int(rise[r]), 2)[0][0, 200] for r in range(n)], np.float32)

    def check(out):
        assert np.asarray(out["ev2"]).dtype == np.int64 and np.array_equal(out["ev2"], ev * 3 + 1)
        assert np.array_equal(out["odd"], (ev // 7) - (ev // 14) * 2 == 1) and np.array_equal(out["hi"], (ev // 1024).astype(np.uint32))
        assert np.asarray(out["hi"]).dtype == np.uint32 and np.asarray(out["odd"]).dtype == np.bool_
        peak = np.abs(want_e).max()
        assert np.abs(np.asarray(out["e"]) - want_e).max() <= 1e-6 * peak

    check(build_dsp({"waveform": wf, "eventnumber": ev, "rise": rise}, dsp_config=recipe, buffer_len=256))
    lg = Table(waveform=WaveformTable(wf, 16.0, np.zeros(n)), eventnumber=Array(ev), rise=Array(rise))
    check(build_dsp(lg, dsp_config=recipe))
    check(build_dsp(LH5Iterator(lg, buffer_len=150), dsp_config=recipe))
    # columns resident on the device: a pass is queued and finished later; the integer program writes its int64 column in place
    rec2 = {"outputs": ["ev2", "m"], "processors": {"ev2": "eventnumber * 3 + 1", "a, b, lo, m": {"function": "min_max", "module": M, "args": ["waveform", "a", "b", "lo", "m"]}}}
    tb = {"waveform": DeviceArray.from_numpy(wf), "eventnumber": DeviceArray.from_numpy(ev)}
    chain, _, _ = build_processing_chain(rec2, tb)
    outs = {"ev2": DeviceArray((n,), np.int64), "m": DeviceArray((n,), np.float32)}
    chain.link(tb, outs)
    for _ in range(3):
        chain.execute(0, n, wait=False)
    chain.wait()
    assert np.array_equal(outs["ev2"].to_numpy(), ev * 3 + 1) and np.array_equal(outs["m"].to_numpy(), wf.max(axis=1))
    with pytest.raises(ValueError, match="wait=False"):
        chain2, _, _ = build_processing_chain(rec2, {"waveform": wf, "eventnumber": ev})
        chain2.execute(0, n, wait=False)
