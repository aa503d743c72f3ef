"""CPU tests of the recipe language's unit / coordinate-grid layer and of the scheduling of whole recipes (translation only: the device
chain is created lazily, no GPU needed).  Reference semantics: processing_chain.py:67-144 (CoordinateGrid), :832-891 (operators as
ufunc processors), :1193-1266 (round & co), :1556-1732 (a processor's grid, is_coord), :1806-1908 + unit_conversion.py:16-79."""
import numpy as np
import pytest

import recipes
from dspeed_amd import _lib
from dspeed_amd.chain import plan
from dspeed_amd.errors import ProcessingChainError
from dspeed_amd.processing_chain import Grid, Quantity, WaveformInput, _Builder, build_processing_chain

M = "dspeed.processors"
SCALAR_OPS = (_lib.OP_SCALAR_AFFINE, _lib.OP_SCALAR_DIV, _lib.OP_SCALAR_CONVERT, _lib.OP_STORE_SCALAR, _lib.OP_SCALAR_FUNC)



@pytest.fixture(autouse=True)
def _programs_whole(monkeypatch):
    """these tests read the arithmetic between per-event values where the builder generated it: the all-scalar tail stays in the program
    (the test of the split itself takes the switch off again)"""
    monkeypatch.setenv("DSPEED_HIP_NO_SCALAR_TAIL", "1")
    monkeypatch.setenv("DSPEED_HIP_NO_SCALAR_HEAD", "1")
    monkeypatch.setenv("DSPEED_HIP_NO_WALKS_BEHIND", "1")


def _tb(n=4, wf_len=8192, t0=0.0, dtype=np.uint16):
    return {"waveform": WaveformInput(np.zeros((n, wf_len), dtype=dtype), 16.0, t0), "baseline": np.zeros(n, dtype=np.float32)}


def _slots(op):
    L = _lib
    opcode, dst, src, _io, ip, _sp = op
    if opcode == L.OP_LOAD:
        return (), (dst,)
    if opcode in (L.OP_STORE, L.OP_TRAP_PICKOFF, L.OP_TRAP_REDUCE, L.OP_PICKOFF, L.OP_TIME_POINT_THRESH, L.OP_MEAN_BELOW,
                  L.OP_TRAP_WINDOW_PICKOFF, L.OP_MIN_MAX, L.OP_LINEAR_SLOPE_FIT, L.OP_AMAX, L.OP_CONVOLVE_AMAX):
        return (src,), ()
    if opcode == L.OP_DWT_HAAR:
        return (src,), (dst, ip[2])
    if opcode == L.OP_MOVING_WINDOW_MULTI:
        return (src,), ((dst, ip[2]) if ip[1] > 1 else (dst,))
    if opcode in SCALAR_OPS:
        return (), ()
    if opcode == L.OP_ELEMENTWISE:
        return tuple(x for x in (src, ip[1], ip[2]) if x >= 0), (dst,)
    return (src,), (dst,)


def _regs_written(op):
    L = _lib
    opcode, dst, _src, io, ip, _sp = op
    if opcode in (L.OP_MIN_MAX, L.OP_LINEAR_SLOPE_FIT):
        return list(range(dst, dst + 4))
    if opcode == L.OP_TRAP_REDUCE:
        pick = ((ip[3] >> 16) & 0x3fff) - 1  # (register of a pick-off done in the same pass)
        return ([] if dst < 0 else list(range(dst, dst + 4))) + ([] if io < 0 else [io]) + ([] if pick < 0 else [pick])
    if opcode in (L.OP_PICKOFF, L.OP_TRAP_PICKOFF, L.OP_TIME_POINT_THRESH, L.OP_AMAX, L.OP_CONVOLVE_AMAX, L.OP_MEAN_BELOW,
                  L.OP_TRAP_WINDOW_PICKOFF, L.OP_SCALAR_AFFINE, L.OP_SCALAR_DIV, L.OP_SCALAR_CONVERT, L.OP_SCALAR_FUNC):
        return [dst]
    return []


def _check_program_order(program):
    """every waveform slot and scalar register is written before it is read"""
    have_slot, have_reg = set(), set()
    for i, op in enumerate(program.ops):
        reads, writes = _slots(op)
        for s in reads:
            assert s in have_slot, f"op {i} reads slot {s} before it is written"
        if op[0] == _lib.OP_TRAP_REDUCE:  # its threshold walk may start at the t_max its own min_max part just wrote
            have_reg.update(_regs_written(op))
        for a in op[5]:
            if a.kind == _lib.ARG_REG:
                assert a.index in have_reg, f"op {i} reads register {a.index} before it is written"
        if op[0] == _lib.OP_STORE_SCALAR:
            assert op[4][0] in have_reg
        have_slot.update(writes)
        have_reg.update(_regs_written(op))


def _peak_live_samples(program):
    first, last = {}, {}
    for i, op in enumerate(program.ops):
        reads, writes = _slots(op)
        for s in set(reads) | set(writes):
            first.setdefault(s, i)
            last[s] = i
    return max(sum(program.slots[s] for s in first if first[s] <= i <= last[s]) for i in range(len(program.ops)))


def test_long_filters_of_the_ge_recipe_run_ahead_of_the_program():
    """the t0 filter and the cusp FIR leave the program for the matrix-core FIR kernels (_extract_stages): the pole-zero corrected
    waveform is written to HBM by a small program of its own, the filters read rows, the program reads their results"""
    chain, mask, out = build_processing_chain(recipes.ICPC, _tb())
    P = chain.program
    assert sorted(mask) == ["baseline", "waveform"] and set(out) == set(recipes.ICPC["outputs"])
    _check_program_order(P)
    opcodes = [o[0] for o in P.ops]
    assert _lib.OP_CONVOLVE not in opcodes and _lib.OP_CONVOLVE_AMAX not in opcodes and _lib.OP_POLE_ZERO not in opcodes
    what = [[o[0] for o in st["program"].ops] for st in chain._stages]
    # (min_max of the raw waveform goes along with the pole-zero rows: the kernel that writes them streams the raw rows anyway)
    assert what == [[_lib.OP_LOAD, _lib.OP_MIN_MAX, _lib.OP_BL_SUBTRACT, _lib.OP_POLE_ZERO] + [_lib.OP_STORE_SCALAR] * 4 + [_lib.OP_STORE],
                    # the t0 filter is piecewise constant (a ramp of 8 taps, a plateau of 125): the run-length FIR kernel (dsp_fir_runs.hip),
                    # and with it what the recipe reads off the filtered waveform -- min_max, and tp_0_est's walk from the maximum; nothing
                    # else reads wf_t0_filter, so it is not stored.  Sample indices are handed on (no unit conversion before the stores)
                    [_lib.OP_LOAD, _lib.OP_CONVOLVE, _lib.OP_MIN_MAX, _lib.OP_TIME_POINT_THRESH] + [_lib.OP_STORE_SCALAR] * 5,
                    [_lib.OP_LOAD, _lib.OP_BL_SUBTRACT, _lib.OP_CONVOLVE, _lib.OP_STORE],
                    # the short trapezoid -> threshold walk in the shape of the rows kernel
                    [_lib.OP_LOAD, _lib.OP_TRAP_REDUCE, _lib.OP_STORE_SCALAR],
                    # the current branch in the shape of dsp_current.hip: window at tp_0_est (a column by now) of the pole-zero rows
                    [_lib.OP_LOAD, _lib.OP_WINDOWER, _lib.OP_AVG_CURRENT, _lib.OP_UPSAMPLER, _lib.OP_MOVING_WINDOW_MULTI, _lib.OP_MIN_MAX]
                    + [_lib.OP_STORE_SCALAR] * 4,
                    # per-event values read straight off rows (dsp_reduce.hip): maximum and one sample of the cusp's
                    [_lib.OP_LOAD, _lib.OP_AMAX, _lib.OP_PICKOFF] + [_lib.OP_STORE_SCALAR] * 2]
    pz, t0f, cusp, atrap, current, cusp_values = chain._stages
    assert pz["what"] == "wf_pz -> HBM + min_max of waveform" and [o[1] for o in pz["outs"]] == ["in:tp_min", "in:tp_max", "in:wf_min", "in:wf_max", "in:wf_pz"]
    assert plan(pz["program"])["kernel"] == "dsp_pz_rows_kernel"
    assert cusp_values["alias"] == {"in:wf_cusp": "in:wf_cusp"} and [o[1] for o in cusp_values["outs"]] == ["in:cuspEmax", "in:cuspEftp"]
    assert not {_lib.OP_MIN_MAX, _lib.OP_AMAX, _lib.OP_PICKOFF} & set(opcodes) and "in:wf_cusp" not in [io[0] for io in P.io] and "in:waveform" not in [io[0] for io in P.io]
    assert [o[1] for o in current["outs"]] == ["in:aoe_t_min", "in:tp_aoe_max", "in:A_min", "in:A_max"]
    assert current["alias"] == {"in:wf_pz": "in:wf_pz", "in:tp_0_est": "in:tp_0_est"}
    assert not {_lib.OP_WINDOWER, _lib.OP_UPSAMPLER, _lib.OP_MOVING_WINDOW_MULTI} & set(opcodes), "the program no longer runs the moving averages"
    assert [o[1] for o in t0f["outs"]] == ["in:conv_tmin", "in:tp_start", "in:conv_min", "in:conv_max", "in:tp_0_est"]
    assert t0f["program"].ops[1][4][2] == 1, "CONVOLVE ip[2]: the caller found the kernel piecewise constant"
    assert plan(t0f["program"])["kernel"] == "dsp_fir_runs_kernel"
    assert atrap["alias"] == {"in:wf_pz": "in:wf_pz", "in:bl_std": "aux:0:1", "in:tp_start": "in:tp_start"} and atrap["outs"] == [("out:tp_0_atrap", "in:tp_0_atrap", None)]
    assert _lib.OP_TRAP_REDUCE in opcodes and "in:wf_t0_filter" not in [io[0] for io in P.io], "the program no longer touches the t0-filtered waveform"
    assert pz["outs"][-1] == ("out:wf_pz", "in:wf_pz", 8192) and t0f["alias"] == {"in:wf_pz": "in:wf_pz", "in:bl_std": "aux:0:1"}
    assert cusp["outs"] == [("out:wf_cusp", "in:wf_cusp", 301)]
    assert cusp["program"].ops[0][4] == (0, 8192 - 6092), "bl_subtract's NaN rule covers the whole waveform: the load screens the rest"
    names = [io[0] for io in P.io]
    assert {"in:wf_pz", "in:tp_0_est", "in:tp_0_atrap", "in:cuspEmax", "in:tp_min"} <= set(names) and chain._ext_alias["in:cuspEmax"] == "in:cuspEmax"
    assert "in:waveform[0:6092]" in chain._in_vars, "columns only a stage reads are linked with the program's own"
    for st in chain._stages:
        _check_program_order(st["program"])


@pytest.mark.parametrize("t0", [48000.0, "per_row"])
def test_whole_ge_recipe_translates_into_one_program(t0, monkeypatch):
    monkeypatch.setenv("DSPEED_HIP_NO_STAGES", "1")
    tb = _tb(t0=np.zeros(4, dtype=np.float32) if t0 == "per_row" else t0)
    chain, mask, out = build_processing_chain(recipes.ICPC, tb)
    P = chain.program
    assert sorted(mask) == ["baseline", "waveform"] and set(out) == set(recipes.ICPC["outputs"])
    assert len(P.ops) <= _lib.MAX_OPS and len(P.slots) <= _lib.MAX_SLOTS and P.n_sregs <= _lib.MAX_SREGS and len(P.io) <= _lib.MAX_IO
    _check_program_order(P)
    # the scheduler keeps at most two 8192-sample waveforms alive (pole_zero in place of the baseline-subtracted waveform, the fits
    # on slice views, the cusp FIR before the pole-zero correction): two waveforms per compute unit instead of "does not fit"
    assert _peak_live_samples(P) <= 2 * 8192 + 1024
    opcodes = [o[0] for o in P.ops]
    assert opcodes.count(_lib.OP_LOAD) == 1 and opcodes.count(_lib.OP_POLE_ZERO) == 1
    pz = P.ops[opcodes.index(_lib.OP_POLE_ZERO)]
    assert pz[1] == pz[2], "pole_zero runs in place"
    # the two fits (baseline window of the subtracted waveform, tail of the pole-zero corrected one) run on the rows ahead of the chain,
    # one waveform per lane, in one pass; the chain reads their results as per-event inputs
    assert not [o for o in P.ops if o[0] == _lib.OP_LINEAR_SLOPE_FIT]
    (fit,) = chain._aux
    assert fit["fits"] == [(0, 0, 700), (1, 1600, 6592)] and fit["mode"] == 1 and fit["sub"] == "in:baseline" and fit["wf"] == "in:waveform"
    assert fit["tau"] == 1716.25 and len(fit["names"]) == 8 and all(nm in [io[0] for io in P.io] for nm in fit["names"])
    # time coordinates leave in ns: one conversion per such output, period ratio 16; tp_aoe_max has no grid and leaves as it is
    conv = [o for o in P.ops if o[0] == _lib.OP_SCALAR_CONVERT and o[4][0] == 0 and o[5][3].value == 16.0]
    n_time_outputs = sum(1 for k in recipes.ICPC["outputs"] if k.startswith("tp_") and k != "tp_aoe_max")
    assert len(conv) == n_time_outputs
    names = [io[0] for io in P.io]
    if t0 == "per_row":
        assert "in:waveform.t0" in names
        assert all(o[5][1].kind == _lib.ARG_REG for o in conv), "per-row offsets are per-event operands"
    else:
        assert all(o[5][1].kind == _lib.ARG_CONST and o[5][1].value == 3000.0 for o in conv)
    # round(tp_0_est + 8*us + 2*us*0.8, wf_etrap.grid): two additions, one conversion with rint
    rints = [o for o in P.ops if o[0] == _lib.OP_SCALAR_CONVERT and o[4][0] == 1]
    assert len(rints) == 1 and rints[0][5][3].value == 1.0
    adds = [o[5][2].value for o in P.ops if o[0] == _lib.OP_SCALAR_AFFINE and o[5][2].kind == _lib.ARG_CONST]
    assert 500.0 in adds and 100.0 in adds and 506.0 in adds
    assert opcodes.count(_lib.OP_SCALAR_DIV) == 1  # QDrift / trapTmax


def test_requesting_fewer_outputs_drops_what_they_do_not_need(monkeypatch):
    chain, _, out = build_processing_chain(recipes.ICPC, _tb(), outputs=["bl_std"])
    # the fit runs on the rows; nothing of the waveform is left for the program to do
    assert [o[0] for o in chain.program.ops] == [_lib.OP_SCALAR_FUNC, _lib.OP_STORE_SCALAR] and chain.program.slots == []
    assert chain._aux[0]["fits"] == [(0, 0, 700)] and chain._aux[0]["wf"] in [io[0] for io in chain.program.io]
    monkeypatch.setenv("DSPEED_HIP_FIT_IN_CHAIN", "1")
    chain, _, out = build_processing_chain(recipes.ICPC, _tb(), outputs=["bl_std"])
    assert [o[0] for o in chain.program.ops] == [_lib.OP_LOAD, _lib.OP_BL_SUBTRACT, _lib.OP_LINEAR_SLOPE_FIT, _lib.OP_STORE_SCALAR]
    assert chain.program.slots == [700], "a constant slice of the input is loaded as such"


def test_a_processor_works_on_the_grid_of_its_first_waveform():
    rec = {"outputs": ["t_whole", "t_win", "a"], "processors": {
        "t_a, t_whole, lo, hi": {"function": "min_max", "module": M, "args": ["waveform", "t_a", "t_whole", "lo", "hi"], "unit": ["ns", "us", "ADC", "ADC"]},
        "t_b, t_win, lo2, hi2": {"function": "min_max", "module": M, "args": ["waveform[1000:3000]", "t_b", "t_win", "lo2", "hi2"],
                                 "unit": ["ns", "ns", "ADC", "ADC"]},
        "a": {"function": "fixed_time_pickoff", "module": M, "args": ["waveform", "t_win", "'n'", "a"], "unit": "ADC"}}}
    chain, _, _ = build_processing_chain(rec, _tb(t0=160.0))
    conv = [o for o in chain.program.ops if o[0] == _lib.OP_SCALAR_CONVERT]
    by_ratio = {(round(o[5][3].value, 6), round(o[5][1].value, 6), round(o[5][2].value, 6)) for o in conv}
    # t_whole: grid (16 ns, 160 ns) -> us: (t + 10) * 0.016;  t_win: grid (16 ns, 160 + 16000 ns) -> ns: (t + 1010) * 16
    # t_win read by a processor on the whole waveform: (t + 1010) * 1 - 10
    assert by_ratio == {(0.016, 10.0, 0.0), (16.0, 1010.0, 0.0), (1.0, 1010.0, 10.0)}


def test_argument_language_on_per_event_variables():
    b = _Builder(_tb(), {})
    wf = b.input_var("waveform")
    assert wf.grid == Grid(16.0, 0.0) and b.eval_arg("waveform.grid") == Grid(16.0)
    assert isinstance(b.eval_arg("waveform.offset"), Quantity) and float(b.eval_arg("waveform.offset")) == 0.0
    assert b.eval_arg("waveform[100:200].offset") == Quantity(1600.0)
    assert b.eval_arg("len(waveform[:round(len(waveform)-(33.6*us/waveform.period))])") == 6092
    assert b.eval_arg("len(waveform[16*us:32*us])") == 1000, "slice bounds may be times"
    assert b.eval_arg("round(1*us, waveform.period)") == Quantity(992.0) and b.eval_arg("ceil(1*us, waveform.period)") == Quantity(1008.0)
    assert b.eval_arg("round(187.5)") == 188 and b.eval_arg("round(186.5)") == 186 and b.eval_arg("floor(-0.5)") == -1
    rec = {"outputs": ["x", "y", "z", "w"], "processors": {
        "t_a, t_b, lo, hi": {"function": "min_max", "module": M, "args": ["waveform", "t_a", "t_b", "lo", "hi"], "unit": ["ns", "ns", "ADC", "ADC"]},
        "x": "hi - lo", "y": "-(lo / hi) * 3", "z": "round(t_b, 4)", "w": "t_b - t_a"}}
    chain, _, _ = build_processing_chain(rec, _tb())
    ops = chain.program.ops
    assert sum(o[0] == _lib.OP_SCALAR_DIV for o in ops) == 1
    sub = [o for o in ops if o[0] == _lib.OP_SCALAR_AFFINE and o[5][1].kind == _lib.ARG_CONST and o[5][1].value == -1.0 and o[5][2].kind == _lib.ARG_REG]
    assert len(sub) == 2, "a - b is (-1 * b) + a, one rounding"
    z = [o for o in ops if o[0] == _lib.OP_SCALAR_CONVERT and o[4][0] == 1]
    assert len(z) == 1 and z[0][5][3].value == 0.25, "round(t, 4): onto a grid of 4 periods"
    # the difference of two coordinates is a plain number: written as it is, no conversion to ns
    w_store = [i for i, io in enumerate(chain.program.io) if io[0] == "out:w"][0]
    w_reg = [o[4][0] for o in ops if o[0] == _lib.OP_STORE_SCALAR and o[3] == w_store][0]
    assert any(o[0] == _lib.OP_SCALAR_AFFINE and o[1] == w_reg for o in ops)


def test_what_the_language_does_not_take_fails_by_name():
    for expr, exc in (("waveform * 2", ProcessingChainError), ("waveform[0:100:-2]", ProcessingChainError), ("waveform[0:100:2]", ProcessingChainError), ("waveform[t_b:t_b+10]", ProcessingChainError),
                      ("baseline.grid", ProcessingChainError), ("t_b % 2", (NotImplementedError, ProcessingChainError))):
        rec = {"outputs": ["x"], "processors": {
            "t_a, t_b, lo, hi": {"function": "min_max", "module": M, "args": ["waveform", "t_a", "t_b", "lo", "hi"], "unit": ["ns", "ns", "ADC", "ADC"]},
            "x": {"function": "fixed_time_pickoff", "module": M, "args": ["waveform", expr, "'n'", "x"]}}}
        with pytest.raises(exc):
            build_processing_chain(rec, _tb())
    plain = {"waveform": np.zeros((4, 1024), dtype=np.float32)}  # no WaveformInput: no grid to take a period from
    with pytest.raises(ProcessingChainError):
        build_processing_chain({"outputs": ["x"], "processors": {"x": f"{M}.fixed_time_pickoff(waveform, 10*us/waveform.period, 'n', x)"}}, plain)


def test_integer_columns_select_numpys_integer_loops():
    """the first ufunc signature every variable can be cast to (reference :1565-1572, 1654-1664); constants are rounded into its type (:1765-1768)"""
    from dspeed_amd.processing_chain import Var, _int_loop_const, _int_loop_of

    v = lambda dt: Var("v", "wf", 8, dt)  # noqa: E731
    for dts, want in ((("uint16",), "uint16"), (("int16", "int16"), "int16"), (("uint16", "int16"), "int32"), (("bool", "int16"), "int16"),
                      (("uint16", "int32"), "int32"), (("uint32",), "uint32")):
        assert _int_loop_of([v(d) for d in dts], "x") == np.dtype(want), dts
        assert np.dtype(want).char == next(t[0] for t in np.add.types if all(np.can_cast(d, t[0]) for d in dts) and t[0] != "?")
    # 64-bit loops (round 4: per-event values in an integer program, waveforms range-tracked in the float64 chain); truth values alone take the
    # int8 loop where the ufunc has no '??' one (floor_divide); int64 beside uint64 is NumPy's float64 loop, refused by name
    for dts, want in ((("int32", "uint32"), "int64"), (("int64",), "int64"), (("uint64", "uint16"), "uint64"), (("bool", "bool"), "int8")):
        assert _int_loop_of([v(d) for d in dts], "x") == np.dtype(want), dts
    with pytest.raises(NotImplementedError, match="float64"):
        _int_loop_of([v("int64"), v("uint64")], "x")
    assert _int_loop_const(2.5, np.dtype("uint16"), "x") == 2.0 and _int_loop_const(3.5, np.dtype("uint16"), "x") == 4.0  # np.round: half to even
    # a constant outside the loop's type wraps around, as NumPy's conversion between its integer scalars does (reference :1765-1768)
    assert _int_loop_const(-1, np.dtype("uint16"), "x") == 65535.0 == float(np.uint16(np.int64(-1)))
    assert _int_loop_const(40000, np.dtype("int16"), "x") == float(np.int16(np.int64(40000)))
    tb = {"u": np.zeros((4, 64), np.uint16), "h": np.zeros((4, 64), np.int16), "ev": np.zeros(4, np.uint16), "n32": np.zeros(4, np.int32)}
    procs = {"a": "u * 2.5", "b": "-h", "c": "ev // 3", "d": "astype(h, 'uint16') + u", "e": "n32 // 2", "f": "u / 2"}
    chain, _, out = build_processing_chain({"outputs": list(procs), "processors": procs}, tb)
    assert {k: x.dtype.name for k, x in out.items()} == {"a": "uint16", "b": "int16", "c": "uint16", "d": "uint16", "e": "int32", "f": "float32"}
    ew = {o[4][0] for o in chain.program.ops if o[0] == _lib.OP_ELEMENTWISE}
    assert ew == {_lib.fn_int(_lib.FN_IMUL, np.uint16), _lib.fn_int(_lib.FN_ISUB, np.int16), _lib.fn_int(_lib.FN_ICAST, np.uint16),
                  _lib.fn_int(_lib.FN_IADD, np.uint16), _lib.FN_DIV}
    sf = {o[4][0] for o in chain.program.ops if o[0] == _lib.OP_SCALAR_FUNC}
    assert sf == {_lib.fn_int(_lib.FN_IFLOORDIV, np.uint16)}
    # (a 32-bit loop between per-event columns of a float32 chain: no float32 holds every int32 -- it runs in the integer program ahead of the chain)
    isl = chain._stages[0]
    assert isl["compute"] == np.int64 and [o[4][0] for o in isl["program"].ops if o[0] == _lib.OP_SCALAR_FUNC] == [_lib.fn_int(_lib.FN_IFLOORDIV, np.int32)]
    mul = next(o for o in chain.program.ops if o[0] == _lib.OP_ELEMENTWISE and o[4][0] == _lib.fn_int(_lib.FN_IMUL, np.uint16))
    assert mul[5][1].value == 2.0


def test_scheduler_only_reorders_within_dependencies():
    for rec in (recipes.C1, recipes.C2, recipes.C2_UNITS, recipes.C3, recipes.C5, recipes.ICPC):
        wf_len = 8192 if rec in (recipes.C3, recipes.C5, recipes.ICPC) else 4096
        tb = _tb(wf_len=wf_len, dtype=np.int16 if rec is recipes.C5 else np.float32)
        tb["t_pick"] = np.zeros(4, dtype=np.float32)
        tb["thr"] = np.zeros(4, dtype=np.float32)
        chain, _, _ = build_processing_chain(rec, tb)
        _check_program_order(chain.program)


REFERENCE_RECIPES = ["tests/configs/icpc-dsp-config.json", "tests/configs/icpc-dsp-config-yaml.yaml", "tests/configs/numpy-parsing.json",
                     "docs/source/notebooks/metadata/dsp-config.json"]


@pytest.mark.parametrize("rel", REFERENCE_RECIPES)
def test_the_references_own_ge_recipes_translate_as_they_stand(rel):
    """Drop-in check on the caller's side of the path: the recipe files the reference ships for germanium detectors go through
    build_processing_chain unmodified and become one program within the device limits.  Only where the reference checkout is
    mounted (the build container); nothing of it is copied into this repository."""
    import os

    path = os.path.join("/root/reference", rel)
    if not os.path.exists(path):
        pytest.skip("reference checkout not mounted")
    tb = _tb(t0=np.zeros(4, dtype=np.float32))
    tb.update({"timestamp": np.zeros(4), "channel": np.zeros(4), "energy": np.zeros(4)})
    chain, mask, out = build_processing_chain(path, tb)
    P = chain.program
    assert len(out) >= 7 and len(P.ops) <= _lib.MAX_OPS and len(P.slots) <= _lib.MAX_SLOTS and P.n_sregs <= _lib.MAX_SREGS
    _check_program_order(P)
    if "icpc" in rel:
        assert len(out) == 34 and _peak_live_samples(P) <= 2 * 8192 + 1024


def test_expressions_on_waveforms_translate_into_elementwise_ops():
    """operators, comparisons, where, isnan, astype, samples and slices of the language (reference :832-1078, 1266-1430) -> device ops"""
    tb = _tb(wf_len=1000, dtype=np.float32)
    tb["eventnumber"] = np.arange(4, dtype=np.int32)
    procs = {"wf_blsub": "waveform - baseline", "pos": "where(wf_blsub < 0, 0, wf_blsub)", "first": "eventnumber == 0",
             "pick": "where(first, wf_blsub[10], wf_blsub[-1])", "down": "wf_blsub[::4]", "ok": "isfinite(pos)", "half": "wf_blsub[100:200] / 2"}
    chain, mask, out = build_processing_chain({"outputs": ["pos", "pick", "down", "ok", "half", "first"], "processors": procs}, tb)
    P = chain.program
    _check_program_order(P)
    ew = [o for o in P.ops if o[0] == _lib.OP_ELEMENTWISE]
    assert sorted(o[4][0] for o in ew) == sorted([_lib.FN_SUB, _lib.FN_LT, _lib.FN_WHERE, _lib.FN_ISFINITE, _lib.FN_DIV])
    sub = next(o for o in ew if o[4][0] == _lib.FN_SUB)
    assert sub[4][1] == -1 and sub[5][1].kind == _lib.ARG_INPUT, "waveform - baseline: slot operand and a per-event column"
    fn = [o for o in P.ops if o[0] == _lib.OP_SCALAR_FUNC]
    assert sorted(o[4][0] for o in fn) == sorted([_lib.FN_EQ, _lib.FN_WHERE])
    samples = [o for o in P.ops if o[0] == _lib.OP_PICKOFF and o[4][1] == 1]
    assert sorted(o[5][0].value for o in samples) == [10.0, 999.0]
    copies = [o for o in P.ops if o[0] == _lib.OP_COPY]
    assert sorted((o[4][0], o[4][1] if len(o[4]) > 1 else 0) for o in copies) == [(0, 4), (100, 0)]
    assert out["ok"].dtype == np.bool_ and out["first"].dtype == np.bool_ and out["pos"].dtype == np.float32
    assert out["down"].shape == (4, 250) and out["half"].shape == (4, 100) and sorted(mask) == ["baseline", "eventnumber", "waveform"]
    io = {name: code for name, _k, code, *_ in P.io}
    assert io["out:ok"] == _lib.BOOL and io["out:first"] == _lib.BOOL and io["out:pos"] == _lib.F32
    # grids (reference :1032-1054): a strided slice multiplies the period, a start shifts the offset; a comparison result has the operand's
    b = _Builder(tb, {})
    assert float(b.eval_arg("waveform[50:100:2].period")) == 32.0 and float(b.eval_arg("waveform[50:100:2].offset")) == 800.0
    assert b.eval_arg("(waveform[50:100] * 2).grid") == Grid(16.0, 800.0) and b.eval_arg("len(waveform[50:100:2])") == 25
    with pytest.raises(ProcessingChainError, match="out of bounds"):
        b.eval_arg("waveform[1000]")


def test_list_literals_are_constant_arrays():
    """test_list_parsing of the reference (tests/test_processing_chain.py:145-159): a list is a constant array, arithmetic between constant
    arrays is NumPy's, and as an output every row holds the array"""
    rec = {"outputs": ["a1", "a2", "wf_out", "half"], "processors": {"a1": "[1,2,3,4,5]", "a2": "[[1, 2], [3, 4]]", "wf_out": "a1+[6,7,8,9,10]",
                                                                      "half": "wf_out / 2"}}
    chain, mask, out = build_processing_chain(rec, _tb())
    n = len(_tb()["baseline"])
    assert mask == [] and out["a1"].shape == (n, 5) and out["a2"].shape == (n, 2, 2)
    assert np.all(out["a1"] == np.array([1, 2, 3, 4, 5])) and np.all(out["a2"] == np.array([[1, 2], [3, 4]]))
    assert np.all(out["wf_out"] == np.array([7, 9, 11, 13, 15])) and out["wf_out"].dtype.kind == "i"
    assert np.all(out["half"] == np.array([3.5, 4.5, 5.5, 6.5, 7.5]))
    with pytest.raises(NotImplementedError, match="constant array beside a variable"):
        build_processing_chain({"outputs": ["x"], "processors": {"x": "waveform[0:3] + [1, 2, 3]"}}, _tb())


def test_loadlh5_gives_a_constant(tmp_path):
    """loadlh5(file, path) (reference processing_chain.py:1444-1467): the taps of a filter kept in a file beside the recipe"""
    from dspeed_amd import lgdo_io

    taps = np.array([0.25, 0.5, 0.25, -0.125], dtype=np.float32)
    f = str(tmp_path / "kernels.npz")
    np.savez(f, taps=taps, gain=np.float64(2.5))
    rec = {"outputs": ["w", "k", "g"], "processors": {"k": f"loadlh5('{f}', 'taps')", "g": f"loadlh5('{f}', '/gain')",
                                                      "w": f"{M}.convolve_wf(waveform, k, 'f', w(8195, 'f'))"}}
    chain, _, out = build_processing_chain(rec, _tb())
    n = len(_tb()["baseline"])
    assert np.array_equal(out["k"], np.broadcast_to(taps, (n, 4))) and np.all(out["g"] == 2.5)
    held = [c for consts in (chain._consts, *(st["consts"] for st in chain._stages)) for name, c in consts.items() if name.startswith("taps:")]
    assert len(held) == 1 and np.array_equal(held[0][:4], taps)
    with pytest.raises(ProcessingChainError, match="LH5 file not found"):
        build_processing_chain({"outputs": ["k"], "processors": {"k": f"loadlh5('{f}', 'nothing')"}}, _tb())
    lgdo_io.CONSTANT_LOADERS.append(lambda file, path: np.arange(3.0) if file == "mem:" else None)
    try:
        _, _, out = build_processing_chain({"outputs": ["k"], "processors": {"k": "loadlh5('mem:', 'x')"}}, _tb())
        assert np.all(out["k"] == np.arange(3.0))
    finally:
        lgdo_io.CONSTANT_LOADERS.pop()


def test_negative_steps_become_backward_copies():
    """wf[::-1], wf[100:10:-2] (NumPy's slice of the buffer in the reference, processing_chain.py:1009-1048): a COPY with a negative stride; of
    a chain input only the span the slice covers is loaded; the grid's period takes the sign of the step"""
    tb = _tb()
    L = tb["waveform"].values.shape[1]
    chain, _, out = build_processing_chain({"outputs": ["r", "s"], "processors": {
        "wf_blsub": f"{M}.bl_subtract(waveform, baseline, wf_blsub)", "r": "wf_blsub[::-1]", "s": "waveform[100:10:-2]"}}, tb)
    assert out["r"].shape[1] == L and out["s"].shape[1] == 45
    copies = [o for o in chain.program.ops if o[0] == _lib.OP_COPY]
    assert sorted((o[4][0], o[4][1]) for o in copies) == [(88, -2), (L - 1, -1)]  # (first source sample, step): 100 is sample 88 of the span 12..100
    spans = {io[0]: (io[3], io[4]) for io in chain.program.io if io[0].startswith("in:waveform")}
    assert spans["in:waveform[12:101]"] == (89, 12)
    for expr in ("waveform[10:100:-1]", "waveform[::0]"):
        with pytest.raises(ProcessingChainError):
            build_processing_chain({"outputs": ["x"], "processors": {"x": expr}}, tb)


def test_arithmetic_that_needs_nothing_of_the_program_runs_ahead_of_it(monkeypatch):
    """the scalar head (_split_scalar_head): the pick-off times of the two energy trapezoids -- arithmetic on the t0 estimate that a stage left in
    HBM -- leave the program, are computed with a row per lane as one more stage, and the program reads them as columns; what is left between
    its waveform ops are the four thresholds of the rise-time walks, fractions of the trapezoid's maximum, which the planner folds into the
    walks.  On the interpreter an op costs a row's wavefront ~ 1 700 cycles whatever it computes."""
    monkeypatch.delenv("DSPEED_HIP_NO_SCALAR_TAIL")
    monkeypatch.delenv("DSPEED_HIP_NO_SCALAR_HEAD")
    chain, _, _ = build_processing_chain(recipes.ICPC, _tb())
    P, head = chain.program, chain._stages[-1]
    assert head["what"] == "per-event arithmetic ahead of the program" and len(chain._stages) == 7
    assert [o[0] for o in head["program"].ops] == [_lib.OP_SCALAR_AFFINE, _lib.OP_STORE_SCALAR, _lib.OP_SCALAR_AFFINE, _lib.OP_SCALAR_AFFINE,
                                                   _lib.OP_SCALAR_CONVERT, _lib.OP_STORE_SCALAR]
    assert head["alias"] == {"in:tp_0_est": "in:tp_0_est"} and [k for _o, k, _l in head["outs"]] == ["in:head:r13.0", "in:head:r19.3"]
    assert plan(head["program"])["kernel"] == "dsp_scalar_kernel"
    _check_program_order(head["program"])
    _check_program_order(P)
    arithmetic = [o for o in P.ops if o[0] in (_lib.OP_SCALAR_AFFINE, _lib.OP_SCALAR_CONVERT, _lib.OP_SCALAR_FUNC, _lib.OP_SCALAR_DIV)]
    assert len(arithmetic) == 4 and all(o[0] == _lib.OP_SCALAR_AFFINE and o[5][0].kind == _lib.ARG_REG for o in arithmetic)
    picks = [o for o in P.ops if o[0] in (_lib.OP_TRAP_PICKOFF, _lib.OP_TRAP_REDUCE) and any(a.kind == _lib.ARG_INPUT for a in o[5])]
    assert len(picks) == 2 and {P.io[a.index][0] for o in picks for a in o[5] if a.kind == _lib.ARG_INPUT} == {"in:head:r13.0", "in:head:r19.3"}
    # the device program of the interpreter: thresholds folded into the walks, each team member's stores one op -- 26 ops of the recipe's program are 12 for a team of three
    info = plan(P)
    assert info["kernel"].startswith("dsp_vm_kernel") and info["team"] == 3 and info["n_device_ops"] == 12, info["n_device_ops"]
    # a register that is reused along the program: only what an op reads at its place in the program counts
    monkeypatch.setenv("DSPEED_HIP_NO_SCALAR_HEAD", "1")
    whole, _, _ = build_processing_chain(recipes.ICPC, _tb())
    assert len(whole.program.ops) == len(P.ops) + 4 and len(whole._stages) == 6


def test_the_scalar_tail_of_a_program_becomes_a_program_of_its_own(monkeypatch):
    monkeypatch.delenv("DSPEED_HIP_NO_SCALAR_TAIL")
    """everything behind the last op that touches a waveform -- arithmetic between per-event values, unit conversions, stores -- is cut off
    (processing_chain._split_scalar_tail) for the row-per-lane kernel; the head hands the registers the tail reads over as columns"""
    scalar = (_lib.OP_SCALAR_AFFINE, _lib.OP_SCALAR_DIV, _lib.OP_SCALAR_CONVERT, _lib.OP_SCALAR_FUNC, _lib.OP_STORE_SCALAR)
    chain, _, out = build_processing_chain(recipes.ICPC, _tb())
    head, tail = chain.program, chain._tail["program"]
    assert tail.slots == [] and all(o[0] in scalar for o in tail.ops) and len(tail.ops) >= 40
    hand = chain._tail["handover"]
    # the head ends with one store per handed-over register, the tail starts with one load per register, same names, same registers
    n = len(hand)
    assert [o[0] for o in head.ops[-n:]] == [_lib.OP_STORE_SCALAR] * n and [head.io[o[3]][0] for o in head.ops[-n:]] == hand
    assert [o[0] for o in tail.ops[:n]] == [_lib.OP_SCALAR_FUNC] * n and [tail.io[o[5][0].index][0] for o in tail.ops[:n]] == hand
    assert [o[4][0] for o in head.ops[-n:]] == [o[1] for o in tail.ops[:n]]
    assert head.ops[-n - 1][0] not in scalar, "the cut is right behind the last waveform op"
    # every output column of the recipe is stored exactly once, by one of the two programs
    stored = [prog.io[o[3]][0] for prog in (head, tail) for o in prog.ops if o[0] == _lib.OP_STORE_SCALAR and prog.io[o[3]][0].startswith("out:")]
    assert sorted(stored) == sorted(f"out:{k}" for k in out)
    # a register the tail computes itself is not handed over; a short tail stays where it is
    written = set()
    for o in tail.ops[n:]:
        for a in o[5]:
            assert a.kind != _lib.ARG_REG or a.index in written or f"tail:r{a.index}" in hand
        if o[0] != _lib.OP_STORE_SCALAR:
            written.add(o[1])
    small, _, _ = build_processing_chain(recipes.C2, _tb(wf_len=4096) | {"t_pick": np.zeros(4, dtype=np.float32)})
    assert small._tail is None


def _programs(chain):
    """everything the device is given for a recipe: the main program, the stages ahead of it, the scalar tail behind it, the fits on the rows"""
    def prog(p):
        return {"ops": [(o[0], o[1], o[2], o[3], o[4], tuple((a.kind, a.index, a.value if a.value == a.value else "nan") for a in o[5])) for o in p.ops],
                "io": list(p.io), "slots": list(p.slots), "n_sregs": p.n_sregs}

    out = {"main": prog(chain.program), "stages": [(st["what"], prog(st["program"]), {k: v.tobytes() for k, v in st["consts"].items()}) for st in chain._stages],
           "tail": prog(chain._tail["program"]) if chain._tail else None, "fits": [{k: v for k, v in g.items()} for g in chain._aux],
           "consts": {k: v.tobytes() for k, v in chain._consts.items()}}
    return out


def test_the_recipe_with_the_references_values_is_the_references_file_op_for_op():
    """recipes.ICPC_REF -- ICPC with the reference's parameter values and its 34 outputs, which the GPU tests run against the oracle -- and the
    reference's own file (tests/configs/icpc-dsp-config.json) translate into the same device programs: ops, bindings, slots, registers, kernels'
    taps, stage by stage.  Only where the reference checkout is mounted; nothing of the file is copied."""
    import os

    path = "/root/reference/tests/configs/icpc-dsp-config.json"
    if not os.path.exists(path):
        pytest.skip("reference checkout not mounted")
    tb = _tb(t0=np.zeros(4, dtype=np.float32))
    theirs, mask_t, out_t = build_processing_chain(path, tb)
    ours, mask_o, out_o = build_processing_chain(recipes.ICPC_REF, tb)
    assert sorted(mask_t) == sorted(mask_o) and list(out_t) == list(out_o) and len(out_o) == 34
    a, b = _programs(theirs), _programs(ours)
    assert a["main"] == b["main"] and a["tail"] == b["tail"] and a["fits"] == b["fits"] and a["consts"] == b["consts"]
    assert len(a["stages"]) == len(b["stages"])
    for sa, sb in zip(a["stages"], b["stages"]):
        assert sa == sb, sa[0]


def test_the_compilers_idea_of_a_piecewise_constant_kernel_is_the_kernels():
    """compiler._piecewise_constant sets CONVOLVE's hint with the limits of csrc/dsp_fir_runs.hip (DSP_FIR_RUNS_MAX runs, DSP_FIR_RUNS_MAX_TAPS taps):
    the two files must agree, and the kernels of the Ge recipes fall where they should"""
    import os
    import re

    from dspeed_amd import compiler

    src = open(os.path.join(os.path.dirname(compiler.__file__), "csrc", "dsp_program.h")).read()
    limits = {k: int(v) for k, v in re.findall(r"#define (DSP_FIR_RUNS_MAX(?:_TAPS)?) +(\d+)", src)}
    assert limits == {"DSP_FIR_RUNS_MAX": compiler.FIR_RUNS_MAX, "DSP_FIR_RUNS_MAX_TAPS": compiler.FIR_RUNS_MAX_TAPS}
    import golden_util

    assert compiler._piecewise_constant(golden_util.recipe_kernel("t0"))                 # a ramp of 8 taps, a plateau of 125
    assert not compiler._piecewise_constant(golden_util.recipe_kernel("cusp"))           # 5792 taps, all different
    assert compiler._piecewise_constant(np.full(512, 0.25, np.float32)) and not compiler._piecewise_constant(np.full(513, 0.25, np.float32))
    ramp = np.arange(30, dtype=np.float32)
    assert not compiler._piecewise_constant(ramp) and compiler._piecewise_constant(ramp[:24]) and not compiler._piecewise_constant(np.array([1.0, np.inf], np.float32))


def test_the_scalar_head_follows_registers_by_position():
    """_split_scalar_head on a hand-made program: a register the head wrote and a waveform op overwrites later, a head result that is only
    stored (a copy of its column), an op that reads the overwritten register (stays)"""
    from dspeed_amd import compiler
    from dspeed_amd.chain import Program, Scalar

    p = Program()
    p.slots = [256]
    p.n_sregs = 8
    wf = p.add_io("in:wf", _lib.IO_WF_IN, np.float32, 256, 0, 256)
    a = p.add_io("in:a", _lib.IO_SCALAR_IN, np.float32)
    outs = [p.add_io(f"out:{k}", _lib.IO_SCALAR_OUT, np.float32) for k in range(3)]
    p.add_op(_lib.OP_LOAD, dst=0, io=wf)
    p.add_op(_lib.OP_SCALAR_AFFINE, dst=1, sp=(Scalar.input(a), Scalar.const(2.0), Scalar.const(0.0)))      # head
    p.add_op(_lib.OP_PICKOFF, dst=2, src=0, ip=(ord("n"), 0), sp=(Scalar.reg(1),))                          # reads the head's r1
    p.add_op(_lib.OP_MIN_MAX, dst=1, src=0)                                                                 # r1 .. r4 start another life
    p.add_op(_lib.OP_SCALAR_AFFINE, dst=5, sp=(Scalar.reg(1), Scalar.const(1.0), Scalar.const(1.0)))        # reads MIN_MAX's r1: stays
    p.add_op(_lib.OP_SCALAR_AFFINE, dst=6, sp=(Scalar.input(a), Scalar.const(3.0), Scalar.const(0.0)))      # head, only stored
    p.add_op(_lib.OP_STORE_SCALAR, io=outs[0], ip=(6,))
    p.add_op(_lib.OP_STORE_SCALAR, io=outs[1], ip=(5,))
    p.add_op(_lib.OP_STORE_SCALAR, io=outs[2], ip=(2,))
    ext = {}
    head = compiler._split_scalar_head(p, np.dtype(np.float32), ext)
    assert [o[0] for o in head["program"].ops] == [_lib.OP_SCALAR_AFFINE, _lib.OP_STORE_SCALAR, _lib.OP_SCALAR_AFFINE, _lib.OP_STORE_SCALAR]
    assert [k for _o, k, _l in head["outs"]] == ["in:head:r1.0", "in:head:r6.1"] and set(ext) == {"in:head:r1.0", "in:head:r6.1"}
    ops = p.ops
    assert [o[0] for o in ops] == [_lib.OP_LOAD, _lib.OP_PICKOFF, _lib.OP_MIN_MAX, _lib.OP_SCALAR_AFFINE, _lib.OP_SCALAR_FUNC, _lib.OP_STORE_SCALAR,
                                   _lib.OP_STORE_SCALAR, _lib.OP_STORE_SCALAR]
    pick, keep, copy = ops[1], ops[3], ops[4]
    assert pick[5][0].kind == _lib.ARG_INPUT and p.io[pick[5][0].index][0] == "in:head:r1.0"
    assert keep[5][0].kind == _lib.ARG_REG and keep[5][0].index == 1, "the register MIN_MAX wrote, not the head's"
    assert copy[1] == 6 and copy[4][0] == _lib.FN_COPY and p.io[copy[5][0].index][0] == "in:head:r6.1"
    _check_program_order(p)
    plan(p)  # (the planner takes it)
    plan(head["program"])


def test_the_planner_folds_a_thresholds_factor_into_the_walk():
    """dsp_plan.cpp: SCALAR_AFFINE d <- x * const + 0 whose result only TIME_POINT_THRESH ops read as their threshold becomes a no-op (the walk
    multiplies); anything else that reads d, an offset, or x changing in between keeps the op"""
    from dspeed_amd.chain import Program, Scalar

    def program(offset=0.0, store_thr=False, rewrite_x=False):
        p = Program()
        p.slots = [512]
        p.n_sregs = 8
        wf = p.add_io("wf", _lib.IO_WF_IN, np.float32, 512, 0, 512)
        p.add_op(_lib.OP_LOAD, dst=0, io=wf)
        p.add_op(_lib.OP_MIN_MAX, dst=0, src=0)
        p.add_op(_lib.OP_SCALAR_AFFINE, dst=4, sp=(Scalar.reg(3), Scalar.const(0.5), Scalar.const(offset)))
        if rewrite_x:
            p.add_op(_lib.OP_AMAX, dst=3, src=0)
        p.add_op(_lib.OP_TIME_POINT_THRESH, dst=5, src=0, sp=(Scalar.reg(4), Scalar.reg(1), Scalar.const(0.0)))
        p.add_op(_lib.OP_TIME_POINT_THRESH, dst=6, src=0, sp=(Scalar.reg(4), Scalar.reg(5), Scalar.const(0.0)))
        for r in (5, 6) + ((4,) if store_thr else ()):
            p.add_op(_lib.OP_STORE_SCALAR, io=p.add_io(f"o{r}", _lib.IO_SCALAR_OUT, np.float32), ip=(r,))
        return p

    # LOAD, MIN_MAX, two walks, the stores as one op
    assert plan(program())["n_device_ops"] == 5
    assert plan(program(offset=-0.0))["n_device_ops"] == 5
    assert plan(program(offset=1.0))["n_device_ops"] == 6          # an offset: the op stays
    assert plan(program(store_thr=True))["n_device_ops"] == 6      # the threshold itself is an output
    assert plan(program(rewrite_x=True))["n_device_ops"] == 7      # the value the walks would multiply changes before they run


def test_the_rise_time_walks_run_behind_the_program(monkeypatch):
    """_split_walks: the five threshold walks of the Ge recipe's program (0.99 / 0.9 / 0.5 / 0.1 of the trapezoid's maximum, each from where the one
    before ended, and the maximum itself) leave it for a launch of their own on the reductions kernel, which reads the pole-zero rows straight
    off HBM with thousands of rows in flight; the maximum is handed over as a column, the walks' stores move along"""
    for k in ("DSPEED_HIP_NO_SCALAR_TAIL", "DSPEED_HIP_NO_SCALAR_HEAD", "DSPEED_HIP_NO_WALKS_BEHIND"):
        monkeypatch.delenv(k)
    chain, _, _ = build_processing_chain(recipes.ICPC, _tb())
    P, W = chain.program, chain._walks["program"]
    assert chain._walks["handover"] == ["walk:r3"]
    assert [o[0] for o in P.ops] == [_lib.OP_LOAD, _lib.OP_TRAP_REDUCE, _lib.OP_TRAP_PICKOFF, _lib.OP_TRAP_REDUCE] + [_lib.OP_STORE_SCALAR] * 5
    assert [P.io[o[3]][0] for o in P.ops[4:]] == ["tail:r3", "tail:r14", "tail:r18", "tail:r22", "walk:r3"]
    kinds = [o[0] for o in W.ops]
    assert kinds == [_lib.OP_LOAD] + [_lib.OP_SCALAR_AFFINE, _lib.OP_TIME_POINT_THRESH] * 4 + [_lib.OP_TIME_POINT_THRESH] + [_lib.OP_STORE_SCALAR] * 5
    assert W.io[W.ops[0][3]][0] == "in:wf_pz" and W.ops[0][4] == P.ops[0][4], "the same rows, with the load's promise about their NaNs"
    assert all(W.io[o[5][0].index][0] == "walk:r3" for o in W.ops if o[0] == _lib.OP_SCALAR_AFFINE)
    assert sorted(W.io[o[3]][0] for o in W.ops if o[0] == _lib.OP_STORE_SCALAR) == sorted(f"tail:r{r}" for r in (5, 7, 9, 11, 12))
    _check_program_order(P)
    _check_program_order(W)
    assert plan(W)["kernel"] == "dsp_reduce_kernel"
    info = plan(P)
    assert info["kernel"].startswith("dsp_vm_kernel") and info["team"] == 3 and info["n_device_ops"] == 8  # (LOAD, three trapezoid ops, the members' stores)
