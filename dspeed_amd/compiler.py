"""From the steps of a recipe to device programs (``dsp_chain_create``): the fits that run on the rows ahead of the chain, the stages the planner gives
to specialised kernels (``_extract_stages``), the integer program (``_int_island``), the main program with its slots, fusions and scalar tail
(``_compile``, ``_split_scalar_tail``) -- and ``_add_step``, which resolves one recipe entry into a step (the role of ProcessorManager.__init__,
reference src/dspeed/processing_chain.py:1527-1775, and of the processor loop of build_processing_chain, :2655-2823)."""
from __future__ import annotations

import ast
import os
from types import SimpleNamespace

import numpy as np

from . import _lib
from .chain import Program, Scalar
from .device import dtype_code
from .errors import DSPFatal, ProcessingChainError
from .language import (Grid, Quantity, SExpr, Var, _Builder, _GENERATORS, _MODULES, _NUMPY_BINARY, _SIGS, _column, _grid_of, _is_int_dtype, _is_scalar,
                       _is_wf, _resolve, _roles, _time_unit_ns, _wf_len)

# processors whose output waveform has the input's dimension name in the gufunc signature ("(n),...->(n)") and therefore its
# coordinate grid (reference :1601-1619, 1700); the others' outputs have no grid unless the recipe declares one
_SAME_DIM = ("bl_subtract", "numpy_subtract", "numpy_add", "min_max_norm", "pole_zero", "double_pole_zero", "trap_filter", "trap_norm", "asym_trap_filter", "moving_window_multi")


def _add_step(b: _Builder, key, node, new_vars, proc_strings):
    module, function = node["module"], node["function"]
    b.cur_key = key
    if module is None:  # inline expression: alias / constant / the result of operators and functions of the language (reference :2676-2696)
        val = b.eval_arg(node["args"][0])
        if isinstance(val, tuple) and not _is_wf(val):
            raise ProcessingChainError(f"'{key}': {val!r} is not a value")
        if isinstance(val, (Var, SExpr, tuple)):
            if isinstance(val, (Var, SExpr)) and "#" in val.name and not getattr(val, "is_input", False):
                val.name = new_vars[0]  # (an expression's result takes the name the recipe gives it)
            if "unit" in node and isinstance(val, (Var, SExpr)) and val.unit is None and isinstance(node["unit"], str):
                val.unit = node["unit"]
            b.vars[new_vars[0]] = val
        else:
            b.vars[new_vars[0]] = Var(new_vars[0], "const", const=val)
        return
    if module not in _MODULES:
        raise NotImplementedError(f"module '{module}' is not available on the device path (processor {module}.{function})")
    if module in ("numpy", "np") and function == "copyto":
        # numpy.copyto(dst, src) as a processor: the copy of a (variable-length) array into a declared output (reference
        # tests/test_processing_chain.py:656-674)
        args = [b.eval_arg(a, new_vars) for a in node["args"]]
        if len(args) != 2 or not isinstance(args[0], Var) or args[0].length is None or not _is_wf(args[1]):
            raise ProcessingChainError(f"numpy.copyto takes a declared output array and an array for parameter {key}")
        dst, src = args
        if _wf_len(src) < dst.length:
            raise ProcessingChainError(f"numpy.copyto for parameter {key}: the output holds {dst.length} samples, the source only {_wf_len(src)}")
        dst.kind = "wf"
        dst.dtype = dst.dtype if dst.dtype is not None else np.dtype(np.float32)
        b._step("slice", [src if _wf_len(src) == dst.length else ("slice", *( (src[1], src[2], src[2] + dst.length) if isinstance(src, tuple) else (src, 0, dst.length))), 0, 1, dst], "wiiW")
        return
    if module in ("numpy", "np") and function not in ("amax",) + tuple(_NUMPY_BINARY):
        raise NotImplementedError(f"numpy.{function} is not available on the device path")
    if "unit" in node:  # "unit": one string, or one per new variable (reference :2705-2711)
        for i, name in enumerate(new_vars):
            unit = node["unit"][i] if isinstance(node["unit"], (list, tuple)) else node["unit"]
            v = b.vars.get(name)
            if v is None:
                b.vars[name] = Var(name, None, unit=unit)
            elif isinstance(v, Var) and v.unit is None:
                v.unit = unit
    args = [b.eval_arg(a, new_vars) for a in node["args"]]
    if module in ("numpy", "np") and function in _NUMPY_BINARY:
        # a NumPy binary ufunc as a processor (numpy.subtract(waveform, bl_mean, wf_blsub), numpy.divide(A_max, trapEmax, AoE)):
        # between per-event values it is the same scalar op the operators make; waveform -/+ per-event value is the subtraction of
        # bl_subtract without its NaN rule (a NaN sample stays a NaN sample)
        if len(args) != 3 or not isinstance(args[2], Var):
            raise ProcessingChainError(f"numpy.{function} takes two operands and an output variable for parameter {key}")
        x, y, out = args
        is_wf = lambda v: (isinstance(v, Var) and v.kind == "wf") or (isinstance(v, tuple) and v and v[0] == "slice")  # noqa: E731
        if is_wf(x) and not is_wf(y) and function in ("subtract", "add"):
            function = "numpy_subtract" if function == "subtract" else "numpy_add"
        elif is_wf(x) or is_wf(y):
            # the ufunc on waveforms, as the operator of the language makes it (one NumPy loop per sample); the declared output names the result
            val = b._wf_binop(_NUMPY_BINARY[function](), x, y, f"numpy.{function}({', '.join(map(str, node['args']))})")
            if out.length is not None and out.length != val.length:
                raise ProcessingChainError(f"failed to broadcast array dimensions for {function}: '{out.name}' holds {out.length} samples, the operands {val.length}")
            val.name = out.name
            val.unit = out.unit if out.unit is not None else val.unit
            val.grid = out.grid if out.grid is not None else val.grid
            b.vars[new_vars[0]] = val
            return
        else:
            b.vars[new_vars[0]] = b._scalar_binop(_NUMPY_BINARY[function](), x, y, str(node["args"]), declared=out)
            return
    if function in _GENERATORS:
        _fold_generator(b, function, args, new_vars)
        return
    if function not in _SIGS:
        raise NotImplementedError(f"processor '{function}' is not implemented on the device path")
    roles = _SIGS[function]
    if len(args) != len(roles):
        raise ProcessingChainError(f"{function} takes {len(roles)} arguments ({len(args)} given) for parameter {key}")
    # give the variables this processor creates their type now, so later recipe entries can slice / measure them
    src_len = None
    for a, r in zip(args, roles):
        if r == "w":
            if isinstance(a, tuple) and a[0] == "slice":
                src_len = a[3] - a[2]
            elif isinstance(a, Var):
                src_len = a.length
    for a, r in zip(args, roles):
        if r == "W" and isinstance(a, Var):
            if a.kind is None:
                a.kind = "wf"
            if a.length is None and function in _SAME_DIM:
                a.length = src_len
            a.dtype = np.dtype(np.float32)
            a.is_coord = False
    args = [_as_taps(b, a, function) if r == "t" else a for a, r in zip(args, roles)]
    args = [_group_constant(b, a, function, key) if r == "i" and isinstance(a, (Var, SExpr)) else a for a, r in zip(args, roles)]
    _, args = _resolve(b, roles, args, same_dim_out=function in _SAME_DIM)
    b.steps.append((function, args, key))
    proc_strings.append(f"{function}({', '.join(str(a.name if isinstance(a, (Var, SExpr)) else a) for a in args)})")


class _PerEventInteger(Exception):
    """an integer parameter of a processor is a column of the input table: the chain is built per value of it (GroupedProcessingChain)"""

    def __init__(self, column):
        super().__init__(column)
        self.column = column


def _group_constant(b: _Builder, a, function, key):
    """An INTEGER parameter of a processor (the rise and flat times of a trapezoid, a wavelet level, the number of moving windows) given as
    a per-event variable.  The reference broadcasts the variable's buffer into the gufunc's "()" slot, if its type can be cast to the
    signature's (:1565-1572, 1702-1745).  The device program holds such parameters as constants -- they size loops and LDS --, so the rows
    are grouped by the column's value and each group runs a chain built for it: here the column is replaced by the value of the group this
    chain is for, or reported to ``build_processing_chain``, which then returns a GroupedProcessingChain."""
    if isinstance(a, Var) and a.kind == "const":
        return a.const
    if not (isinstance(a, Var) and a.kind == "scalar" and a.is_input and a.source is not None and getattr(a, "ext_key", None) is None):
        raise NotImplementedError(f"{function} ({key}): the integer parameter '{a.name}' is computed per event inside the recipe; the device "
                                  "programs take integer parameters as constants or as columns of the input table (rows grouped by value)")
    if not np.can_cast(a.dtype, np.int32):  # ("fii->f" and the like: the column must cast to the signature's 'i', reference :1565-1572)
        raise ProcessingChainError(f"could not find a type signature matching the types of the variables given for {function} ({a.name} is {a.dtype})")
    if a.source not in b.group_values:
        raise _PerEventInteger(a.source)
    return int(b.group_values[a.source])


def _as_taps(b: _Builder, a, function):
    """A constant array given where a processor takes its kernel -- a list literal, loadlh5(...), or a recipe entry holding one -- becomes
    the same kind of variable a kernel generator leaves.  The reference passes the array itself, and its type takes part in the choice of
    the loop (:1565-1572): a float64 or integer array selects the processor's float64 loop, which the float32 chain does not run."""
    arr = a.const if isinstance(a, Var) and a.kind == "const" and isinstance(a.const, np.ndarray) else a
    if not isinstance(arr, np.ndarray):
        return a
    if arr.ndim != 1 or arr.size < 1:
        raise ProcessingChainError(f"{function}: the kernel must be a one-dimensional array, not one of shape {arr.shape}")
    if not np.can_cast(arr.dtype, np.float32):
        raise NotImplementedError(f"{function}: a {arr.dtype.name} kernel selects the float64 loop of the processor in the reference; give it "
                                  "as float32 values")
    b._anon += 1
    name = a.name if isinstance(a, Var) else f"kernel#{b._anon}"
    return Var(name, "taps", int(arr.size), np.float32, const=np.ascontiguousarray(arr, dtype=np.float32))


def _fold_generator(b: _Builder, function, args, new_vars):
    """Kernel generators (cusp_filter, zac_filter, t0_filter, moving_slope) with constant arguments run once, here, on the host
    (reference :2797-2813)."""
    from . import processors as P

    *scal, out = args
    if not isinstance(out, Var) or out.length is None:
        raise ProcessingChainError(f"{function}: the kernel argument must be declared as name(length, 'f')")
    period = b.default_period
    vals = []
    for s in scal:
        if isinstance(s, Quantity):
            if period is None:
                raise ProcessingChainError(f"{function}: time quantity without a sampling period")
            s = float(s) / period
        if isinstance(s, (Var, tuple)):
            raise NotImplementedError(f"{function} with per-event arguments is not supported")
        vals.append(float(s))
    k = np.zeros(out.length, dtype=np.float32)
    getattr(P, function)(*vals, k)
    out.kind, out.const, out.dtype = "taps", k, np.dtype(np.float32)


# ----------------------------------------------------------------------------------------------------------------
# program generation
# ----------------------------------------------------------------------------------------------------------------
def _loop_dtype(b: _Builder):
    """float32 loop unless an input selects the float64 one (first castable signature wins, reference :1565-1572, 1654-1664):
    float64 / int32 / uint32 waveforms or float64 scalar columns cannot be cast to float32."""
    for v in b.vars.values():
        if isinstance(v, Var) and v.is_input and v.dtype is not None:
            if v.source is not None and v.source.endswith(".t0"):
                continue  # the time of sample 0 is a coordinate offset, not a processor argument: a float64 t0 column (what LH5 files hold)
                # does not make the processors run their float64 loops (the value enters coordinate conversions in the loop's type)
            if v.kind == "wf" and v.dtype in (np.dtype(np.float64), np.dtype(np.int32), np.dtype(np.uint32)):
                return np.dtype(np.float64)
            if v.kind == "scalar" and v.dtype == np.dtype(np.float64):
                return np.dtype(np.float64)
    return np.dtype(np.float32)


def _schedule(steps):
    """Order the processors so that few waveforms are alive at a time -- every waveform variable of a chain lives in LDS, and the
    LDS a waveform needs decides how many run per compute unit.  The reference's order (depth-first from the outputs,
    processing_chain.py:2601-2651) is one valid order of a dependency graph; the processors are pure, so any other valid order
    computes the same values.  List scheduling with two rules: a processor that only reduces waveforms to numbers runs as soon as its
    operands exist (it can only end lifetimes); among the ones that create a waveform, the one reading the oldest waveform goes
    first (finish with a waveform before starting on newer ones), an element-wise or recursive filter that may then take its place
    last; a processor whose result could not be consumed yet (a consumer waits for another operand) yields to the others."""
    def leaves(a, acc):
        if isinstance(a, SExpr):
            for x in a.args:
                leaves(x, acc)
        elif isinstance(a, Var):
            acc.append(a)
        elif isinstance(a, tuple) and a and a[0] == "slice":
            acc.append(a[1])
        return acc

    producer = {}
    ins, creates = [], []
    for j, (fn, args, _k) in enumerate(steps):
        roles = _roles(fn)
        mine, reads = [], []
        for a, r in zip(args, roles):
            (mine if r in "WS" else reads).extend(leaves(a, []))
        ins.append(reads)
        creates.append(any(r == "W" for r in roles))
        for v in mine:
            producer.setdefault(id(v), j)
    deps = [{producer[id(v)] for v in reads if id(v) in producer and producer[id(v)] != j} for j, reads in enumerate(ins)]
    consumers = [[] for _ in steps]
    for c, d in enumerate(deps):
        for j in d:
            consumers[j].append(c)
    born = {}  # waveform -> position in the new order of the processor that made it (inputs: -1)
    order, done = [], set()
    while len(order) < len(steps):
        ready = [j for j in range(len(steps)) if j not in done and deps[j] <= done]
        if not ready:  # (cannot happen for steps that came out of the dependency resolution; keep the given order)
            return steps
        reducers = [j for j in ready if not creates[j]]
        if reducers:
            j = reducers[0]
        else:
            def age(j):
                wfs = [born.get(id(v), -1) for v in ins[j] if v.kind == "wf"]
                return min(wfs) if wfs else len(steps)
            def waits(j):  # a consumer of what j makes still lacks an operand that does not itself come from j: j's waveform
                family, todo = {j}, [j]  # would sit in LDS until that arrives
                while todo:
                    for c in consumers[todo.pop()]:
                        if c not in family:
                            family.add(c)
                            todo.append(c)
                return any(deps[c] - done - family for c in consumers[j])
            # (same oldest waveform: the one that could overwrite it in place waits until the others have read it)
            j = min(ready, key=lambda j: (waits(j), age(j), steps[j][0].startswith("ew:") or steps[j][0] in ("bl_subtract", "numpy_subtract", "numpy_add", "min_max_norm", "pole_zero", "double_pole_zero"), j))
        for a, r in zip(steps[j][1], _roles(steps[j][0])):
            if r == "W" and isinstance(a, Var):
                born[id(a)] = len(order)
        order.append(j)
        done.add(j)
    return [steps[j] for j in order]


def _leaves(a, acc):
    if isinstance(a, SExpr):
        for x in a.args:
            _leaves(x, acc)
    elif isinstance(a, Var):
        acc.append(a)
    elif isinstance(a, tuple) and a and a[0] == "slice":
        acc.append(a[1])
    return acc


def _live_steps(b: _Builder, steps, out_pars):
    """the steps the outputs depend on, in their order"""
    needed = {id(v) for o in out_pars for v in _leaves(b.vars.get(o), [])}
    live = []
    for fn, args, key in reversed(steps):
        roles = _roles(fn)
        mine = [v for a, r in zip(args, roles) if r in "WS" for v in _leaves(a, [])]
        if any(id(v) in needed for v in mine):
            live.append((fn, args, key))
            for a, r in zip(args, roles):
                if r not in "WS":
                    needed.update(id(v) for v in _leaves(a, []))
    return live[::-1]


#: taps from which a convolve_wf / fft_convolve_wf goes to the matrix-core FIR kernels ahead of the program (dsp_fir_mfma.hip needs 64)
STAGE_MIN_TAPS = 64


#: what the run-length FIR kernel takes (DSP_FIR_RUNS_MAX / DSP_FIR_RUNS_MAX_TAPS of csrc/dsp_program.h)
FIR_RUNS_MAX, FIR_RUNS_MAX_TAPS = 24, 512


def _piecewise_constant(taps) -> bool:
    """Is the kernel a few runs of equal taps (``t0_filter``: a ramp of 8 and a plateau of 125; moving averages)?  Then convolve_wf is a
    handful of differences of prefix sums per output instead of a multiply-add per tap (csrc/dsp_fir_runs.hip)."""
    k = np.asarray(taps, dtype=np.float32)
    if k.ndim != 1 or not 1 <= k.size <= FIR_RUNS_MAX_TAPS or not np.isfinite(k).all():
        return False
    edges = np.concatenate([[np.float32(0)], k, [np.float32(0)]])
    return int(np.count_nonzero(edges[1:] != edges[:-1])) <= FIR_RUNS_MAX + 1


def _extract_stages(b: _Builder, steps, out_pars, n_rows, ft):
    """Long FIRs leave the program: each ``convolve_wf`` with a constant kernel of STAGE_MIN_TAPS or more taps becomes a launch of the
    matrix-core FIR kernels ahead of the program (one waveform per wavefront is the wrong shape for 133 x 8192 or 5792 x 301
    multiply-adds per waveform; 64 waveforms x 320 outputs per workgroup on the MFMA units is 4 - 40 times faster, and the filter's
    input and output slots leave the program's LDS).  The FIR kernels read rows from HBM, so a filter's input is a chain input, the
    input minus a per-event value (bl_subtract: done while staging), or -- anything else, the pole-zero corrected waveform of the Ge
    recipes -- a waveform that a small program of its own writes to HBM first (32 kB per waveform: noise at a recipe's rate).  What a
    stage wrote is an input of the later stages and of the program; processors that only fed a stage drop out of the program.
    Returns (steps left to the program, stages in launch order)."""
    import copy

    if all(st[0] in ("convolve_wf", "fft_convolve_wf", "amax", "bl_subtract", "alias") for st in steps):
        return steps, []  # the program is nothing but filters (BASELINE configs[2]): dsp_chain_create gives it the FIR kernel as a whole
    out_names = set(out_pars)
    for o in out_pars:
        for v in _leaves(b.vars.get(o), []):
            out_names.add(v.name)
    stages = []

    def base_of(a):
        if isinstance(a, Var):
            return a
        if isinstance(a, tuple) and a and a[0] == "slice":
            return a[1]
        return None

    def producer_of(v):
        for st in steps:
            for a, r in zip(st[1], _roles(st[0])):
                if r in "WS" and a is v:
                    return st
        return None

    def plain_scalar(x):  # a constant, a per-event input column or a fit / stage result: in HBM before the stage runs
        if isinstance(x, Var):
            return x.kind == "scalar" and x.sreg is None and ((x.is_input and x.source is not None) or getattr(x, "ext_key", None) is not None)
        return isinstance(x, (int, float, np.integer, np.floating)) and not isinstance(x, (bool, Quantity))

    def row_input(v):  # rows of the input table or of an earlier stage
        return isinstance(v, Var) and v.kind == "wf" and ((v.is_input and v.source is not None) or getattr(v, "ext_key", None) is not None)

    def ancestors(v):
        """steps that compute v from inputs and earlier results, in order"""
        want, todo = [], [v]
        seen = set()
        while todo:
            x = todo.pop()
            if id(x) in seen or row_input(x):
                continue
            seen.add(id(x))
            st = producer_of(x)
            if st is None:
                continue
            if not any(st is w for w in want):
                want.append(st)
            for a, r in zip(st[1], _roles(st[0])):
                if r not in "WS":
                    todo.extend(_leaves(a, []))
        return [st for st in steps if any(st is w for w in want)]

    def build(stage_steps, outs, what):
        """compile stage_steps (on copies of the variables) into a program that writes the variables ``outs``"""
        vars2, steps2 = copy.deepcopy((b.vars, stage_steps))
        b2 = copy.copy(b)
        b2.vars, b2.steps, b2._conversions, b2.stage_ft = vars2, list(steps2), {}, ft
        for v in vars2.values():
            if isinstance(v, Var) and getattr(v, "aux_io", None) is not None:
                v.aux_io = None  # (an index into the main program's bindings; the stage binds the fit's column by its name)
        pc, _tb = _compile(b2, [o.name for o in outs], n_rows, [], stage_mode=True)
        rec = {"what": what, "program": pc._program, "consts": pc._consts, "in_vars": pc._in_vars, "alias": pc._ext_alias,
               "outs": [(f"out:{o.name}", f"in:{o.name}", o.length if o.kind == "wf" else None) for o in outs], "chain": None, "bufs": {}}
        stages.append(rec)
        for o in outs:  # from here on the variable is a row / column in HBM
            if o.kind == "wf":  # pole_zero returns an all-NaN waveform for an input with a NaN and DSPFatal for a NaN of its own making
                made_by = producer_of(o)
                o.nan_uniform = made_by is not None and made_by[0] == "pole_zero"
            o.ext_key, o.is_input, o.slot, o.sreg = f"in:{o.name}", True, None, None
            if o.kind == "wf":
                o.ext_len, o.offset, o.dtype = o.length, 0, np.dtype(np.float32)

    def rows_steps(v):
        """steps that read only the rows v and per-event values already in HBM and make per-event values only; in order, closed under
        their own results"""
        made, picked = set(), []
        for st in steps:
            roles = _roles(st[0])
            ins = [(a, r) for a, r in zip(st[1], roles) if r not in "WS"]
            outs = [a for a, r in zip(st[1], roles) if r in "WS"]
            if not outs or any(r == "W" for r in roles) or st[0] in ("alias",) or not any(base_of(a) is v and isinstance(a, Var) for a, r in ins):
                continue
            ok = True
            for a, r in ins:
                if base_of(a) is v and isinstance(a, Var):
                    continue
                if isinstance(a, (Var, SExpr, tuple)) and not (isinstance(a, tuple) and a and a[0] == "char"):
                    ok = ok and isinstance(a, Var) and (plain_scalar(a) or id(a) in made)
            if ok and all(isinstance(o, Var) and o.name not in out_names for o in outs[:0]) and all(isinstance(o, Var) for o in outs):
                picked.append(st)
                made.update(id(o) for o in outs)
        return picked

    def fusable(g, group):
        """what the run-length FIR kernel reads off the waveform it has just filtered (the reductions of dsp_reduce.hip): min_max, numpy.amax,
        a sample at a constant integral time, time_point_thresh from a constant sample or from min_max's t_min / t_max"""
        number = lambda x: isinstance(x, (int, float, np.integer, np.floating)) and not isinstance(x, (bool, Quantity))  # noqa: E731
        if g[0] in ("min_max", "amax"):
            return True
        if g[0] == "fixed_time_pickoff":
            return number(g[1][1]) and float(g[1][1]) == int(float(g[1][1]))
        if g[0] == "time_point_thresh":
            _w, thr, start, walk, _o = g[1]
            extremes = [o for x in group if x[0] == "min_max" for o in x[1][1:3]]
            return ((number(thr) or plain_scalar(thr)) and number(walk) and float(walk) in (0.0, 1.0)
                    and ((number(start) and float(start) == int(float(start))) or any(start is o for o in extremes)))
        return False

    for st in list(steps):
        fn, args, key = st
        if fn not in ("convolve_wf", "fft_convolve_wf") or not any(st is x for x in steps):
            continue
        taps, out = args[1], args[3]
        if not (isinstance(taps, Var) and taps.kind == "taps" and taps.const is not None and isinstance(out, Var) and out.length):
            continue
        m = int(taps.length)
        base, n_in = base_of(args[0]), _wf_len(args[0])
        if base is None or n_in is None or m < STAGE_MIN_TAPS or m > n_in or not np.isfinite(taps.const).all():
            continue
        mode = args[2][1][0] if isinstance(args[2], tuple) and args[2][0] == "char" else (chr(args[2]) if isinstance(args[2], (int, np.integer)) else None)
        want_len = {"v": n_in - m + 1, "s": n_in, "f": n_in + m - 1}.get(mode)
        if want_len is None or want_len != out.length:
            continue  # (the program's own op reports it)
        # --- the filter's input as rows in HBM
        pre = []
        if not row_input(base):
            pst = producer_of(base)
            direct = (pst is not None and pst[0] == "bl_subtract" and row_input(base_of(pst[1][0])) and plain_scalar(pst[1][1])
                      and base.name not in out_names)
            if direct:
                pre = [pst]
            else:
                anc = ancestors(base)
                if not anc or any(a[0] in ("convolve_wf", "fft_convolve_wf") for a in anc):
                    continue
                # min_max of the raw rows goes along: the pole-zero rows kernel streams them anyway (dsp_pz.hip), a launch of the
                # reductions kernel would read them once more
                extra, made = [], []
                raw = base_of(anc[0][1][0]) if anc and anc[0][0] in ("bl_subtract", "pole_zero") else None
                if (ft == np.dtype(np.float32) and isinstance(raw, Var) and row_input(raw) and [a[0] for a in anc] in (["bl_subtract", "pole_zero"], ["pole_zero"])
                        and os.environ.get("DSPEED_HIP_NO_ROW_REDUCTIONS") != "1"):
                    mm = [g for g in rows_steps(raw) if g[0] == "min_max" and g[1][0] is raw]
                    if len(mm) == 1:
                        extra, made = mm, [o for o in mm[0][1][1:5]]
                build(extra + anc, made + [base], f"{base.name} -> HBM" + (f" + min_max of {raw.name}" if extra else ""))
                for o in made:
                    o.kind = "scalar"
                steps = [x for x in steps if not any(x is g for g in extra)]
        # --- the filter itself; numpy.amax goes along when it is the only reader
        users = [x for x in steps if x is not st and any(base_of(a) is out for a, r in zip(x[1], _roles(x[0])) if r not in "WS")]
        if (len(users) == 1 and users[0][0] == "amax" and users[0][1][0] is out and out.name not in out_names and isinstance(users[0][1][2], Var)
                and mode == "v" and out.length <= 320):  # (what the amax form of the kernel takes; else the filtered waveform is kept)
            build(pre + [st, users[0]], [users[0][1][2]], f"{fn} {key} + amax")
            users[0][1][2].kind = "scalar"
            gone = [st, users[0]]
        else:
            # a piecewise-constant kernel (the t0 filter) on float32 rows: prefix sums instead of products, and what the recipe reads off
            # the filtered waveform -- min_max, the threshold walk of the t0 estimate -- in the same pass; a filtered waveform that nothing
            # else reads then never reaches HBM (csrc/dsp_fir_runs.hip)
            group = []
            if (ft == np.dtype(np.float32) and not pre and isinstance(args[0], Var) and row_input(args[0]) and np.dtype(args[0].dtype) == np.dtype(np.float32)
                    and n_in % 8 == 0 and _piecewise_constant(taps.const) and os.environ.get("DSPEED_HIP_NO_FIR_RUNS") != "1"):
                for g in rows_steps(out):
                    if g[1][0] is out and fusable(g, group):
                        group.append(g)
                by_fn = [g[0] for g in group]
                if by_fn.count("min_max") > 1 or by_fn.count("amax") > 1 or by_fn.count("fixed_time_pickoff") > 4 or by_fn.count("time_point_thresh") > 2:
                    group = []
            if group:
                made = [o for g in group for o, r in zip(g[1], _roles(g[0])) if r in "WS"]
                keep = out.name in out_names or any(not any(x is g for g in group) for x in users)
                build([st] + group, made + ([out] if keep else []), f"{fn} {key} on prefix sums + per-event values of {out.name}")
                for o in made:
                    o.kind = "scalar"
                gone = [st] + group
            else:
                build(pre + [st], [out], f"{fn} {key}")
                gone = [st]
        steps = [x for x in steps if not any(x is g for g in gone)]
    if not stages:
        return steps, stages

    # --- short trapezoids that only feed min_max / time_point_thresh, on rows: the lane-per-waveform kernel (dsp_rows.hip) runs the
    # reference's recurrence as it is, 64 waveforms per instruction, where the program replays it twice per waveform (the t0 chain of the
    # Ge recipes, asym_trap_filter -> time_point_thresh: a fifth of the program).  The kernel reads rows, so the trapezoid's input must
    # be rows (an input or what a stage above wrote) and every per-event operand a column in HBM: what such an operand depends on --
    # min_max of the t0-filtered waveform -- moves ahead of the program as well, as a small program of its own on the same rows.
    trap_fns = ("trap_filter", "trap_norm", "asym_trap_filter")
    for st in list(steps):
        fn, args, key = st
        if fn not in trap_fns or not any(st is x for x in steps):
            continue
        src, dst = args[0], args[-1]
        ints = args[1:-1]
        if not (isinstance(src, Var) and row_input(src) and isinstance(dst, Var) and dst.name not in out_names and src.length and src.length % 8 == 0
                and src.length >= 16 and all(isinstance(x, (int, float, np.integer, np.floating)) and not isinstance(x, (bool, Quantity)) and float(x) == int(x)
                                             for x in ints)):
            continue
        iv = [int(x) for x in ints]
        lags = (iv[0], iv[0] + iv[1], iv[0] + iv[1] + iv[2]) if fn == "asym_trap_filter" else (iv[0], iv[0] + iv[1], 2 * iv[0] + iv[1])
        if min(lags) < 8 or (((max(lags) + 8 + 7) // 8) * 8 + 8) * 256 > 80 * 1024:
            continue
        users = [x for x in steps if x is not st and any(base_of(a) is dst for a, r in zip(x[1], _roles(x[0])) if r not in "WS")]
        kinds = sorted(x[0] for x in users)
        if kinds not in (["min_max"], ["time_point_thresh"], ["min_max", "time_point_thresh"]) or not all(x[1][0] is dst for x in users):
            continue
        mm_outs = [o for x in users if x[0] == "min_max" for o in x[1][1:5]]
        # per-event operands of the walk: in HBM already, the reduction's own t_min / t_max, or movable ahead of the program
        need = [a for x in users if x[0] == "time_point_thresh" for a in x[1][1:4] if isinstance(a, (Var, SExpr))]
        movers, fine = [], True
        for a in need:
            if isinstance(a, Var) and (plain_scalar(a) or any(a is o for o in mm_outs)):
                continue
            pst = producer_of(a) if isinstance(a, Var) else None
            rows_v = next((base_of(x) for x, r in zip(pst[1], _roles(pst[0])) if r not in "WS" and isinstance(x, Var) and row_input(x)), None) if pst else None
            group = rows_steps(rows_v) if rows_v is not None else []
            if pst is None or not any(pst is g for g in group):
                fine = False
                break
            movers.append((rows_v, group))
        if not fine:
            continue
        for rows_v, group in movers:
            group = [g for g in group if any(g is x for x in steps)]
            if not group:
                continue
            outs = [o for g in group for o, r in zip(g[1], _roles(g[0])) if r in "WS"]
            build(group, outs, f"per-event values of {rows_v.name}")
            for o in outs:
                o.kind = "scalar"
            steps = [x for x in steps if not any(x is g for g in group)]
        outs = [o for x in users for o, r in zip(x[1], _roles(x[0])) if r in "WS"]
        build([st] + users, outs, f"{fn} {key} on rows")
        for o in outs:
            o.kind = "scalar"
        steps = [x for x in steps if x is not st and not any(x is u for u in users)]

    # --- the current branch (windower -> avg_current -> upsampler -> moving_window_multi -> min_max, the A/E part of the Ge recipes) on
    # rows: three moving averages that alternate direction are float32 recurrences over 4784 samples each -- 30 % of the program, which
    # replays their rounding twice per pass.  dsp_current.hip gives every waveform a lane and runs them as written (bit-exact), keeping
    # checkpoints instead of the intermediate waveforms.  Needs the window's source as rows in HBM and its start as a column there.
    def only_user(v, fn_name):
        users = [x for x in steps if any(base_of(a) is v for a, r in zip(x[1], _roles(x[0])) if r not in "WS")]
        return users[0] if len(users) == 1 and users[0][0] == fn_name and users[0][1][0] is v and v.name not in out_names else None

    for st in list(steps):
        if st[0] != "windower" or not any(st is x for x in steps):
            continue
        src, start, w_le = st[1]
        if not (isinstance(src, Var) and row_input(src) and np.dtype(src.dtype) == np.dtype(np.float32) and isinstance(w_le, Var)):
            continue
        if not plain_scalar(start):
            # the window's start (tp_0_est) is computed by the program from rows in HBM (the t0-filtered waveform): what computes it moves ahead
            # as a small program of its own on those rows, like the operands of the t0 chain's walk above
            pst = producer_of(start) if isinstance(start, Var) else None
            rows_v = next((base_of(x) for x, r in zip(pst[1], _roles(pst[0])) if r not in "WS" and isinstance(x, Var) and row_input(x)), None) if pst else None
            group = [g for g in (rows_steps(rows_v) if rows_v is not None else []) if any(g is x for x in steps)]
            if pst is None or not any(pst is g for g in group):
                continue
            moved = [o for g in group for o, r in zip(g[1], _roles(g[0])) if r in "WS"]
            build(group, moved, f"per-event values of {rows_v.name}")
            for o in moved:
                o.kind = "scalar"
            steps = [x for x in steps if not any(x is g for g in group)]
        chain_steps, v = [st], w_le
        for fn_name in ("avg_current", "upsampler", "moving_window_multi", "min_max"):
            nxt = only_user(v, fn_name)
            if nxt is None:
                break
            chain_steps.append(nxt)
            v = nxt[1][-1]
        if len(chain_steps) != 5:
            continue
        outs = [o for o in chain_steps[-1][1][1:5]]
        if not all(isinstance(o, Var) for o in outs):
            continue
        build(chain_steps, outs, f"current branch of {src.name} on rows")
        for o in outs:
            o.kind = "scalar"
        steps = [x for x in steps if not any(x is c for c in chain_steps)]

    # what the stages' results replaced is not computed any more: producers of staged variables, and whatever only fed them
    staged = {id(v) for v in b.vars.values() if isinstance(v, Var) and getattr(v, "ext_key", None) is not None and getattr(v, "aux_io", None) is None}
    steps = [x for x in steps if not any(r in "WS" and id(a) in staged for a, r in zip(x[1], _roles(x[0])))]
    steps = _live_steps(b, steps, out_pars)

    # --- per-event values read straight off rows in HBM: min_max, numpy.amax and a sample at a constant integral time of an input or of a
    # stage's waveform.  In the program they cost a LOAD of the whole row into LDS and a pass over it, at the occupancy the longest
    # waveform leaves (one wavefront per SIMD for 8192 samples); dsp_reduce.hip streams the row through registers once.
    def reducible(st):
        if st[0] in ("min_max", "amax"):
            return True
        if st[0] == "fixed_time_pickoff":
            t = st[1][1]
            return isinstance(t, (int, float, np.integer, np.floating)) and not isinstance(t, (bool, Quantity)) and float(t) == int(float(t))
        if st[0] == "time_point_thresh":  # a walk from a constant sample (from an extreme of the same rows: what the t0 chain above moves)
            _w, thr, start, walk, _o = st[1]
            number = lambda x: isinstance(x, (int, float, np.integer, np.floating)) and not isinstance(x, (bool, Quantity))  # noqa: E731
            return (number(thr) or plain_scalar(thr)) and number(start) and float(start) == int(float(start)) and number(walk) and float(walk) in (0.0, 1.0)
        return False

    if ft == np.dtype(np.float32) and os.environ.get("DSPEED_HIP_NO_ROW_REDUCTIONS") != "1":
        for v in [x for x in list(b.vars.values()) if row_input(x)]:
            if np.dtype(v.dtype) not in (np.dtype(np.float32), np.dtype(np.int16), np.dtype(np.uint16)):
                continue
            group = [g for g in rows_steps(v) if reducible(g) and g[1][0] is v]
            by_fn = [g[0] for g in group]
            if not group or by_fn.count("min_max") > 1 or by_fn.count("amax") > 1 or by_fn.count("fixed_time_pickoff") > 4 or by_fn.count("time_point_thresh") > 2:
                continue
            rest = [x for x in steps if not any(x is g for g in group)]
            if any(base_of(a) is v for x in rest for a, r in zip(x[1], _roles(x[0])) if r not in "WS"):
                continue  # (the program loads these rows for something else as well: there the reduction is one more pass over LDS, no row traffic)
            if not any(r in "wW" for x in rest for r in _roles(x[0])):
                continue  # (the program would be left without a waveform: nothing gained by a launch of its own)
            outs = [o for g in group for o, r in zip(g[1], _roles(g[0])) if r in "WS"]
            build(group, outs, f"per-event values of {v.name} off its rows")
            for o in outs:
                o.kind = "scalar"
            steps = rest

    return steps, stages


def _column_dtype(dt):
    """type of the column an integer value travels in (the 8-bit integer types have no column type of their own: 16 bits hold them)"""
    dt = np.dtype(dt)
    return {np.dtype(np.int8): np.dtype(np.int16), np.dtype(np.uint8): np.dtype(np.uint16)}.get(dt, dt)


def _int_island(b: _Builder, steps, out_pars, ft):
    """Per-event INTEGER arithmetic that no float register holds -- NumPy's 64-bit loops ('ll->l', 'QQ->Q': int64 / uint64 columns, int32
    beside uint32; reference :1565-1572), comparisons, ``where`` and casts of their results, and in a float32 chain the 32-bit loops too -- leaves
    the programs: it becomes an *integer program* (``dsp_chain_create(..., DSP_I64)``: 64-bit integer registers, NumPy's wrap-around bit
    for bit, dsp_scalar.hip) that runs ahead of everything else on the input table's integer columns.  What the recipe's outputs or the
    other programs read of it arrives as a column of the value's own type (``SExpr.op == 'ext'``).  Operands must be columns of the input
    table, constants or such arithmetic itself: a 64-bit loop on a value a processor computes is refused by name (a 32-bit one then stays
    where it was: the float operation, exact below 2^24).  Returns the stage's description (None: nothing to do) and {output: dtype} of the
    recipe outputs it writes itself."""
    nodes, seen = [], set()

    def visit(x):
        if isinstance(x, SExpr) and id(x) not in seen:
            seen.add(id(x))
            for y in x.args:
                visit(y)
            nodes.append(x)  # (operands first)

    for _fn, args, _key in steps:
        for a in args:
            visit(a)
    for o in out_pars:
        visit(b.vars.get(o))

    def int_dt(x):
        dt = getattr(x, "dtype", None)
        if isinstance(x, SExpr):
            return np.dtype(dt) if x.op == "func" and dt is not None and np.dtype(dt).kind in "iub" else None
        if isinstance(x, Var) and x.kind == "scalar":
            return np.dtype(dt) if dt is not None and np.dtype(dt).kind in "iub" else None
        return None

    def is_leaf(x):  # a column of the input table: in HBM before any program runs
        return isinstance(x, Var) and x.kind == "scalar" and x.is_input and x.source is not None and x.sreg is None and getattr(x, "ext_key", None) is None

    wide = lambda dt: dt is not None and dt.itemsize == 8 and dt.kind in "iu"  # noqa: E731
    eligible = {}

    def ok(x):  # computable ahead of the programs, in integers
        if isinstance(x, (int, float, np.integer, np.floating)) and not isinstance(x, Quantity):
            return True
        if is_leaf(x):
            return int_dt(x) is not None
        if isinstance(x, SExpr):
            if id(x) not in eligible:
                eligible[id(x)] = x.op == "func" and int_dt(x) is not None and all(ok(y) for y in x.args[1:])
            return eligible[id(x)]
        return False

    need = []
    for n in nodes:
        if n.op != "func":
            continue
        opn = [x for x in n.args[1:] if isinstance(x, (Var, SExpr))]
        is_wide = wide(int_dt(n)) or any(wide(int_dt(x)) for x in opn)
        narrow32 = ((int(n.args[0]) >> 8) & 0xff) == 32 and ft != np.dtype(np.float64) and int_dt(n) is not None
        if is_wide:
            if int_dt(n) is None or not ok(n):
                raise NotImplementedError(f"'{n.name}': 64-bit integers reach the device as columns of the input table and arithmetic between them; "
                                          "here they meet a value a processor computes, or leave as a float (astype of a 64-bit integer)")
            need.append(n)
        elif narrow32 and ok(n):
            need.append(n)
    if not need:
        return None, {}
    island = {}

    def take(x):
        if isinstance(x, SExpr) and id(x) not in island:
            for y in x.args[1:]:
                take(y)
            island[id(x)] = x

    for n in need:
        take(n)
    members = [n for n in nodes if id(n) in island]  # (operands first)

    # who reads a member from outside: a processor, an expression that stays behind, an output of the recipe
    outside = set()
    for _fn, args, _key in steps:
        outside.update(id(a) for a in args if isinstance(a, SExpr) and id(a) in island)
    for n in nodes:
        if id(n) not in island:
            outside.update(id(y) for y in n.args if isinstance(y, SExpr) and id(y) in island)
    direct = {}
    for o in out_pars:
        v = b.vars.get(o)
        if isinstance(v, SExpr) and id(v) in island and not (v.is_coord is True and _time_unit_ns(v.unit) is not None):
            direct.setdefault(id(v), []).append(o)
        elif isinstance(v, SExpr) and id(v) in island:
            outside.add(id(v))

    t = Program()
    in_vars, leaf_io = {}, {}

    def opnd(x):
        if isinstance(x, SExpr):
            return Scalar.reg(x.sreg)
        if isinstance(x, Var):
            if id(x) not in leaf_io:
                name = f"in:{x.name}"
                leaf_io[id(x)] = t.add_io(name, _lib.IO_SCALAR_IN, np.dtype(x.dtype))
                in_vars[name] = x
            return Scalar.input(leaf_io[id(x)])
        return Scalar.const(float(x))

    outs, out_dtypes, direct_out = [], {}, {}
    for k, n in enumerate(members):
        n.sreg = t.add_sregs(1)
        sp = [opnd(x) for x in n.args[1:]]
        t.add_op(_lib.OP_SCALAR_FUNC, dst=n.sreg, ip=(int(n.args[0]),), sp=tuple(sp + [Scalar.const(0.0)] * (3 - len(sp))))
    for k, n in enumerate(members):
        nat = int_dt(n)
        is_u64 = int(nat == np.dtype(np.uint64))
        for o in direct.get(id(n), ()):
            t.add_op(_lib.OP_STORE_SCALAR, io=t.add_io(f"out:{o}", _lib.IO_SCALAR_OUT, _column_dtype(nat)), ip=(n.sreg, is_u64))
            direct_out[o] = (nat, _column_dtype(nat))
        if id(n) in outside:
            key = f"in:isl{k}"
            t.add_op(_lib.OP_STORE_SCALAR, io=t.add_io(f"out:isl{k}", _lib.IO_SCALAR_OUT, _column_dtype(nat)), ip=(n.sreg, is_u64))
            outs.append((f"out:isl{k}", key, None))
            out_dtypes[key] = _column_dtype(nat)
            n.ext_key = key
    for n in members:  # from here on a member is a column in HBM to everybody else
        n.ext_dtype = _column_dtype(int_dt(n))
        n.op, n.args, n.sreg = "ext", (), None
    if len(t.ops) > _lib.MAX_OPS or len(t.io) > _lib.MAX_IO or t.n_sregs > _lib.MAX_SREGS:
        raise NotImplementedError("the recipe's integer arithmetic is too large for one device program (ops/bindings/registers limit)")
    stage = {"what": "integer arithmetic between per-event columns (64-bit registers)", "program": t, "consts": {}, "in_vars": in_vars, "alias": {},
             "outs": outs, "out_dtypes": out_dtypes, "compute": np.dtype(np.int64), "chain": None, "bufs": {}}
    return stage, direct_out


def _compile(b: _Builder, out_pars, n_rows, proc_strings, stage_mode=False):
    """``stage_mode``: the program of a stage that runs ahead of the main program (_extract_stages): fits and other stages are not taken
    out of it again; their results arrive as bindings (``Var.ext_key``)."""
    p = Program()
    ft = b.stage_ft if stage_mode else _loop_dtype(b)
    in_bind, out_bind, consts = {}, {}, {}
    ext_alias = {}  # binding name -> name of the buffer a fit / stage ahead of the program filled (a slice of it has a name of its own)
    vector_lens = {}
    steps = b.steps
    island, island_out = (None, {}) if stage_mode else _int_island(b, steps, out_pars, ft)
    # --- linear_slope_fit on the rows of the batch (dsp_linear_slope_fit_rows: one waveform per lane) instead of inside the program,
    # where its sequential float32 recurrences cost a third of a LEGEND recipe: a fit whose waveform is an input, the input minus a
    # per-event input / constant (bl_subtract or numpy.subtract), or the pole_zero of that (constant tau), read whole or through a
    # constant slice.  The kernel runs ahead of the chain on the same stream; the chain reads its results as per-event inputs.
    aux = []  # one launch per (input waveform, subtraction, pole-zero) pipeline
    if not stage_mode and os.environ.get("DSPEED_HIP_FIT_IN_CHAIN", "0") != "1":
        producer = {}
        for fn, args, _k in steps:
            for a, r in zip(args, _roles(fn)):
                if r == "W" and isinstance(a, Var):
                    producer[a.name] = (fn, args)

        def plain_scalar(x):  # a constant or a per-event input column (known before the chain runs)
            if isinstance(x, Var):
                return x.kind == "scalar" and x.is_input and x.sreg is None
            return isinstance(x, (int, float, np.integer, np.floating)) and not isinstance(x, (bool, Quantity))

        def pipeline_of(v):
            """(input wf Var, first sample, length, sub operand, sub mode, tau) of waveform v, or None"""
            tau = None
            if not v.is_input and v.name in producer and producer[v.name][0] == "pole_zero":
                _fn, a = producer[v.name]
                if not isinstance(a[0], Var) or isinstance(a[1], (Var, SExpr, Quantity, tuple)):
                    return None
                tau, v = float(a[1]), a[0]
            sub, mode = None, 0
            src = v
            if not v.is_input:
                if v.name not in producer or producer[v.name][0] not in ("bl_subtract", "numpy_subtract"):
                    return None
                fn2, a = producer[v.name]
                if not plain_scalar(a[1]):
                    return None
                sub, mode, src = a[1], (1 if fn2 == "bl_subtract" else 2), a[0]
            lo, n = 0, None
            if isinstance(src, tuple) and src[0] == "slice":
                src, lo, n = src[1], src[2], src[3] - src[2]
            if not (isinstance(src, Var) and src.is_input and src.kind == "wf" and src.offset == 0):
                return None
            return src, lo, (src.length if n is None else n), sub, mode, tau

        kept = []
        for fn, args, key in steps:
            done = False
            if fn == "linear_slope_fit" and all(isinstance(a, Var) for a in args[1:5]):
                a0, first, count = args[0], 0, None
                if isinstance(a0, tuple) and a0[0] == "slice":
                    a0, first, count = a0[1], a0[2], a0[3] - a0[2]
                pl = pipeline_of(a0) if isinstance(a0, Var) else None
                if pl is not None:
                    src, lo, n, sub, mode, tau = pl
                    count = n - first if count is None else count
                    if 0 <= first and first + count <= n and count >= 1:
                        gkey = (src.name, lo, n, id(sub) if isinstance(sub, Var) else ("c", sub), mode)
                        grp = next((g for g in aux if g["key"] == gkey and len(g["fits"]) < _lib.FIT_MAX
                                    and (g["tau"] == tau or tau is None or g["tau"] is None)), None)
                        if grp is None:
                            grp = {"key": gkey, "src": src, "lo": lo, "len": n, "sub": sub, "mode": mode, "tau": tau, "fits": [], "outs": []}
                            aux.append(grp)
                        if tau is not None:
                            grp["tau"] = tau
                        grp["fits"].append((1 if tau is not None else 0, first, count))
                        grp["outs"].append(list(args[1:5]))
                        done = True
            if not done:
                kept.append((fn, args, key))
        steps = kept
        if aux:  # what only fed those fits is not computed any more
            def leaves(a, acc):
                if isinstance(a, SExpr):
                    for x in a.args:
                        leaves(x, acc)
                elif isinstance(a, Var):
                    acc.append(a)
                elif isinstance(a, tuple) and a and a[0] == "slice":
                    acc.append(a[1])
                return acc

            needed = {id(v) for o in out_pars for v in leaves(b.vars.get(o), [])}
            live = []
            for fn, args, key in reversed(steps):
                roles = _roles(fn)
                mine = [v for a, r in zip(args, roles) if r in "WS" for v in leaves(a, [])]
                if any(id(v) in needed for v in mine):
                    live.append((fn, args, key))
                    for a, r in zip(args, roles):
                        if r not in "WS":
                            needed.update(id(v) for v in leaves(a, []))
            steps = live[::-1]
        for gi, g in enumerate(aux):  # the chain reads the results as per-event input columns
            for k, outs in enumerate(g["outs"]):
                for q, o in enumerate(outs):
                    o.kind = "scalar"
                    o.aux_io = p.add_io(f"aux:{gi}:{4 * k + q}", _lib.IO_SCALAR_IN, ft)
                    o.ext_key = f"aux:{gi}:{4 * k + q}"

    # --- long FIRs on the matrix cores, ahead of the program: their results are bindings of the program
    stages = []
    if not stage_mode and ft == np.dtype(np.float32) and os.environ.get("DSPEED_HIP_NO_STAGES", "0") != "1":
        steps, stages = _extract_stages(b, steps, out_pars, n_rows, ft)

    if island is not None:
        stages = [island] + stages
    steps = b.steps = _schedule(steps)
    out_names = set(out_pars)  # names of the variables that are outputs (a variable may have another name than the output: alias, named slice)
    for o in out_pars:
        ov = b.vars.get(o)
        ov = ov[1] if _is_wf(ov) and isinstance(ov, tuple) else ov
        if isinstance(ov, Var):
            out_names.add(ov.name)

    # --- uses: which step reads which variable last (slot reuse, in-place decisions, fusions)
    def wf_of(a):
        if isinstance(a, Var):
            return a
        if isinstance(a, tuple) and a[0] == "slice":
            return a[1]
        return None

    # --- slice push-down: an element-wise result (bl_subtract) that is read ONLY through one constant slice [lo:hi] -- the
    # long-FIR recipes do that, icpc-dsp-config.json:160-239 -- is computed on that slice alone: a 6092-sample slot instead of
    # an 8192-sample one plus a copy.  Same values: the op is per sample.
    whole_nan_rule = {}  # sliced-input variable -> (samples before, samples after) the slice that a LOAD screens for NaN

    def slices_of(v):
        found, plain = set(), False
        for _fn, a2, _k in steps:
            for x in a2:
                if isinstance(x, tuple) and x[0] == "slice" and x[1] is v:
                    found.add((x[2], x[3]))
                elif x is v:
                    plain = True
        return found, plain

    for si, (fn, args, key) in enumerate(steps):
        if fn != "bl_subtract" or not isinstance(args[0], Var) or not isinstance(args[-1], Var) or args[-1].name in out_names:
            continue
        src_v, dst_v = args[0], args[-1]
        found, plain = slices_of(dst_v)
        uses_of_dst = sum(1 for _fn, a2, _k in steps for x in a2 if x is dst_v)  # the producing step itself counts once
        if len(found) != 1 or uses_of_dst != 1 or not src_v.is_input or src_v.kind != "wf":
            continue
        (lo, hi), = found
        if not (0 <= lo < hi <= (src_v.length or 0)):
            continue
        # ... except for bl_subtract's NaN rule, which looks at the WHOLE waveform (bl_subtract.py:41-44): the load of the slice also screens
        # the samples outside it (LOAD ip[0..1]); another processor reading the same slice of the input as a plain view must not see that
        if any(isinstance(x, tuple) and x[0] == "slice" and x[1] is src_v and (x[2], x[3]) == (lo, hi) for _f, a2, _k in steps for x in a2):
            continue
        whole_nan_rule[f"{src_v.name}[{lo}:{hi}]"] = (lo, src_v.length - hi)
        new_args = list(args)
        new_args[0] = ("slice", src_v, lo, hi)
        steps[si] = (fn, new_args, key)
        dst_v.length = hi - lo
        for sj, (fn2, a2, k2) in enumerate(steps):
            if sj != si:
                steps[sj] = (fn2, [dst_v if (isinstance(x, tuple) and x[0] == "slice" and x[1] is dst_v) else x for x in a2], k2)

    last_use = {}
    for si, (fn, args, _) in enumerate(steps):
        roles = _roles(fn)
        for a, r in zip(args, roles):
            v = wf_of(a)
            if v is not None and r in "wts":
                last_use[v.name] = si
    for o in out_pars:  # (a variable may be known by another name than the output's: an alias, a named slice)
        last_use[o] = len(steps) + 1
        v = wf_of(b.vars.get(o)) if isinstance(b.vars.get(o), (Var, tuple)) else None
        if v is not None:
            last_use[v.name] = len(steps) + 1

    free_slots, slot_len = [], []

    def new_slot(length):
        # one slot per waveform variable: dsp_chain_create packs slots with disjoint lifetimes into the same LDS, whatever their
        # lengths.  Only a recipe with more variables than slot ids goes back to an id whose variable is dead.
        if len(slot_len) >= _lib.MAX_SLOTS:
            for s in free_slots:
                if slot_len[s] == length:
                    free_slots.remove(s)
                    return s
        slot_len.append(int(length))
        return len(slot_len) - 1

    def release(v, si):
        if v.slot is not None and last_use.get(v.name, -1) <= si and v.kind == "wf":
            if v.slot not in free_slots:
                free_slots.append(v.slot)

    def period_of(args):
        for a in args:
            v = wf_of(a)
            if v is not None and v.period is not None:
                return v.period
        return b.default_period

    def ensure_loaded(a, si):
        """Waveform operand -> slot.  Chain inputs are loaded on first use (a constant slice of an input is free)."""
        if isinstance(a, tuple) and a[0] == "slice":
            _, base, lo, hi = a
            if base.is_input:
                key = f"{base.name}[{lo}:{hi}]"
                v = b.vars.get(key)
                if v is None:
                    v = Var(key, "wf", hi - lo, base.dtype, source=base.source, offset=lo, grid=_grid_of(a), is_coord=False)
                    v.is_input = True
                    if getattr(base, "ext_key", None) is not None:  # (a waveform a stage wrote: same buffer, first sample lo)
                        v.ext_key, v.ext_len = base.ext_key, getattr(base, "ext_len", base.length)
                    b.vars[key] = v
                    last_use[key] = last_use.get(base.name, si)
                return ensure_loaded(v, si)
            key = f"{base.name}[{lo}:{hi}]"
            v = b.vars.get(key)
            if v is not None and v.slot is not None:
                return v  # the same slice was materialised for an earlier processor and is still alive
            src = ensure_loaded(base, si)
            v = Var(key, "wf", hi - lo, np.float32, grid=_grid_of(a), is_coord=False)
            v.slot = new_slot(v.length)
            p.add_op(_lib.OP_COPY, dst=v.slot, src=src.slot, ip=(lo,))
            b.vars[key] = v
            last_use[key] = max((sj for sj, (_, a2, _k) in enumerate(steps)
                                 for x in a2 if isinstance(x, tuple) and x[0] == "slice" and x[1] is base and x[2] == lo and x[3] == hi), default=si)
            return v
        v = a
        if v.kind != "wf":
            raise ProcessingChainError(f"'{v.name}' is not a waveform")
        if v.slot is None:
            if not v.is_input:
                raise ProcessingChainError(f"waveform '{v.name}' is used before it is computed")
            if getattr(v, "ext_key", None) is not None:  # written by a stage ahead of the program: float32 rows of the variable's length
                io = p.add_io(f"in:{v.name}", _lib.IO_WF_IN, np.float32, v.length, v.offset, getattr(v, "ext_len", v.length))
                ext_alias[f"in:{v.name}"] = v.ext_key
            else:
                col = _column(b.tb_in, v.source)
                full_len = col.shape[1]
                io = p.add_io(f"in:{v.name}", _lib.IO_WF_IN, col.dtype, v.length, v.offset, full_len)
                in_bind[f"in:{v.name}"] = v
            v.slot = new_slot(v.length)
            screens = whole_nan_rule.get(v.name, ())
            if getattr(v, "nan_uniform", False):  # rows a stage wrote with pole_zero's rule: all NaN or free of NaN (DSP_OP_LOAD ip[2])
                screens = (*(screens or (0, 0)), 1)
            p.add_op(_lib.OP_LOAD, dst=v.slot, io=io, ip=screens)
        return v

    def scalar_operand(a, args, integer=False, what=""):
        """Scalar argument -> Scalar (const / input column / register)."""
        if isinstance(a, SExpr) and a.op == "ext":  # a column the integer program ahead of this one wrote (_int_island)
            if getattr(a, "ext_key", None) is None:
                raise ProcessingChainError(f"{what}: '{a.name}' is written by the integer program as an output only")
            if a.io is None:
                a.io = p.add_io(a.ext_key, _lib.IO_SCALAR_IN, a.ext_dtype)
                ext_alias[a.ext_key] = a.ext_key
            return Scalar.input(a.io)
        if isinstance(a, SExpr):
            if a.sreg is None:  # first reader: emit the op (its operands were computed by earlier processors)
                def opnd(x):
                    return scalar_operand(x, args, what=what) if isinstance(x, (Var, SExpr)) else Scalar.const(float(x))

                r = p.add_sregs(1)
                if a.op == "affine":
                    p.add_op(_lib.OP_SCALAR_AFFINE, dst=r, sp=tuple(opnd(x) for x in a.args))
                elif a.op == "div":
                    p.add_op(_lib.OP_SCALAR_DIV, dst=r, sp=tuple(opnd(x) for x in a.args))
                elif a.op == "func":
                    code, *xs = a.args
                    if code == _lib.FN_COPY and getattr(a, "want_dtype", None) not in (None, ft):
                        raise NotImplementedError(f"{what}: astype to {a.want_dtype} in a chain whose loop type is {ft}")
                    if (code >> 8) & 0xff == 32 and ft != np.dtype(np.float64):
                        # a 32-bit integer loop between per-event values of a float32 chain (len(v) // 2, eventnumber + 1): the registers are
                        # float32, so the operation is the float one -- the same integer as long as operands and result stay below 2**24
                        float_fn = {_lib.FN_IADD: _lib.FN_ADD, _lib.FN_ISUB: _lib.FN_SUB, _lib.FN_IMUL: _lib.FN_MUL, _lib.FN_IFLOORDIV: _lib.FN_FLOORDIV}
                        if code & 0xff not in float_fn:
                            raise NotImplementedError(f"{what}: astype to a 32-bit integer in a chain whose loop type is {ft}")
                        code = float_fn[code & 0xff]
                    sp = [opnd(x) for x in xs] + [Scalar.const(0.0)] * (3 - len(xs))
                    p.add_op(_lib.OP_SCALAR_FUNC, dst=r, ip=(code,), sp=tuple(sp))
                elif a.op == "convert":
                    x, off_in, off_out, ratio = a.args
                    p.add_op(_lib.OP_SCALAR_CONVERT, dst=r, ip=(a.mode,), sp=(opnd(x), opnd(off_in), opnd(off_out), Scalar.const(ratio)))
                else:
                    raise ProcessingChainError(f"{what}: cannot evaluate '{a.name}'")
                a.sreg = r
            return Scalar.reg(a.sreg)
        if isinstance(a, Var):
            if a.kind == "const":
                a = a.const
            elif a.kind == "scalar":
                if a.sreg is not None:
                    return Scalar.reg(a.sreg)
                if getattr(a, "aux_io", None) is not None:  # a fit done ahead of the chain
                    return Scalar.input(a.aux_io)
                if getattr(a, "ext_key", None) is not None:  # a fit or a stage ahead of this program
                    if a.io is None:
                        a.io = p.add_io(f"in:{a.name}", _lib.IO_SCALAR_IN, ft)
                        ext_alias[f"in:{a.name}"] = a.ext_key
                    return Scalar.input(a.io)
                if a.is_input:
                    if a.io is None:
                        col = _column(b.tb_in, a.source)
                        a.io = p.add_io(f"in:{a.name}", _lib.IO_SCALAR_IN, col.dtype)
                        in_bind[f"in:{a.name}"] = a
                    return Scalar.input(a.io)
                raise ProcessingChainError(f"scalar '{a.name}' is used before it is computed")
            else:
                raise ProcessingChainError(f"{what}: '{a.name}' is not a scalar")
        if isinstance(a, Quantity):  # (no grid on this processor: the reference refuses; the input's sampling period is used)
            per = period_of(args)
            if per is None:
                raise ProcessingChainError(f"{what}: time quantity without a sampling period (wrap the input in WaveformInput)")
            a = float(a) / per
        if isinstance(a, (tuple, Grid)):
            raise ProcessingChainError(f"{what}: expected a number or a per-event variable, got {a!r}")
        if integer:  # reference :1767-1768: integer parameters are rounded after the unit conversion
            return int(a) if isinstance(a, (int, np.integer)) else int(np.rint(float(a)))
        return Scalar.const(float(a))

    def char_of(a):
        if isinstance(a, tuple) and a[0] == "char":
            return ord(a[1][0])
        if isinstance(a, (int, np.integer)):
            return int(a)
        raise ProcessingChainError(f"expected a character argument, got {a!r}")

    def out_wf(a, length, src_var=None):
        if not isinstance(a, Var):
            raise ProcessingChainError("output argument must be a variable name")
        if a.kind is None:
            a.kind, a.length = "wf", length
        if a.kind != "wf":
            raise ProcessingChainError(f"'{a.name}' is not a waveform output")
        if a.length is None:
            a.length = length
        a.dtype = np.dtype(np.float32)
        return a

    def out_scalar(a):
        if not isinstance(a, Var):
            raise ProcessingChainError("output argument must be a variable name")
        if a.kind is None:
            a.kind = "scalar"
        if a.sreg is None:
            a.sreg = p.add_sregs(1)
        return a

    trap_ops = {"trap_filter": _lib.OP_TRAP_FILTER, "trap_norm": _lib.OP_TRAP_NORM, "asym_trap_filter": _lib.OP_ASYM_TRAP}
    skip = set()
    pending_reduce = {}  # trapezoid output name -> what its fused min_max / time_point_thresh op needs
    for si, (fn, args, key) in enumerate(steps):
        if si in skip:
            continue
        what = f"{fn} ({key})"
        if fn == "alias":
            continue
        if fn in ("bl_subtract", "numpy_subtract", "numpy_add", "min_max_norm", "pole_zero", "double_pole_zero"):
            src = ensure_loaded(args[0], si)
            dst = out_wf(args[-1], src.length, src)
            inplace = last_use.get(src.name, -1) <= si
            dst.slot = src.slot if inplace else new_slot(src.length)
            if fn == "bl_subtract":
                p.add_op(_lib.OP_BL_SUBTRACT, dst=dst.slot, src=src.slot, sp=(scalar_operand(args[1], args, what=what),))
            elif fn in ("numpy_subtract", "numpy_add"):
                y = args[1]
                if fn == "numpy_add":  # w + y = w - (-y), exactly
                    y = SExpr("affine", (y, -1.0, -0.0), "(-...)", None, False, None) if isinstance(y, (Var, SExpr)) else -float(y)
                p.add_op(_lib.OP_BL_SUBTRACT, dst=dst.slot, src=src.slot, ip=(1,), sp=(scalar_operand(y, args, what=what),))
            elif fn == "min_max_norm":
                p.add_op(_lib.OP_MIN_MAX_NORM, dst=dst.slot, src=src.slot, sp=(scalar_operand(args[1], args, what=what),
                                                                                scalar_operand(args[2], args, what=what)))
            elif fn == "pole_zero":
                tau = scalar_operand(args[1], args, what=what)
                p.add_op(_lib.OP_POLE_ZERO, dst=dst.slot, src=src.slot, sp=(tau,))
            else:
                sp = tuple(scalar_operand(a, args, what=what) for a in args[1:4])
                p.add_op(_lib.OP_DOUBLE_POLE_ZERO, dst=dst.slot, src=src.slot, sp=sp)
            if not inplace:
                release(src, si)
        elif fn.startswith("ew:"):
            code, *opn, dst = args
            if code == _lib.FN_COPY and getattr(dst, "want_dtype", None) not in (None, ft):
                raise NotImplementedError(f"{what}: astype to {dst.want_dtype} in a chain whose loop type is {ft}")
            if (code >> 8) & 0xff == 32 and ft != np.dtype(np.float64):
                raise NotImplementedError(f"{what}: a 32-bit integer loop on waveforms in a chain whose loop type is {ft} (its values do not hold every "
                                          "32-bit integer); make one operand a float (astype)")
            slots, sps, srcs = [], [], []
            for x, r in zip(opn, fn[3:]):
                if r == "w":
                    v = ensure_loaded(x, si)
                    slots.append(v.slot)
                    sps.append(Scalar.const(0.0))
                    srcs.append(v)
                else:
                    slots.append(-1)
                    sps.append(scalar_operand(x, args, what=what) if r == "s" else Scalar.const(0.0))
            dead = next((v for v in srcs if last_use.get(v.name, -1) <= si), None)  # the result may take the place of an operand nobody reads again
            dst.slot = dead.slot if dead is not None else new_slot(dst.length)
            p.add_op(_lib.OP_ELEMENTWISE, dst=dst.slot, src=slots[0], ip=(code, slots[1], slots[2]), sp=tuple(sps))
            for v in srcs:
                if v is not dead and v.slot != dst.slot:
                    release(v, si)
        elif fn == "sample":
            src = ensure_loaded(args[0], si)
            o = out_scalar(args[2])
            p.add_op(_lib.OP_PICKOFF, dst=o.sreg, src=src.slot, ip=(ord("n"), 1), sp=(Scalar.const(float(args[1])),))
            release(src, si)
        elif fn == "get":
            src = ensure_loaded(args[0], si)
            o = out_scalar(args[2])
            p.add_op(_lib.OP_PICKOFF, dst=o.sreg, src=src.slot, ip=(ord("n"), 2), sp=(scalar_operand(args[1], args, what=what), Scalar.const(float("nan"))))
            release(src, si)
        elif fn == "slice":
            src = ensure_loaded(args[0], si)
            dst = args[3]
            dst.slot = new_slot(dst.length)
            p.add_op(_lib.OP_COPY, dst=dst.slot, src=src.slot, ip=(int(args[1]), int(args[2])))
            release(src, si)
        elif fn in trap_ops:
            src = ensure_loaded(args[0], si)
            ints = [scalar_operand(a, args, integer=True, what=what) for a in args[1:-1]]
            ints += [0] * (3 - len(ints))
            dst = out_wf(args[-1], src.length, src)
            # fusion: the trapezoid's only consumer is the next fixed_time_pickoff and it is not an output
            nxt = steps[si + 1] if si + 1 < len(steps) else None
            if (nxt and nxt[0] == "fixed_time_pickoff" and wf_of(nxt[1][0]) is dst and last_use.get(dst.name) == si + 1
                    and dst.name not in out_names and char_of(nxt[1][2]) != ord("s")):
                t_in = scalar_operand(nxt[1][1], nxt[1], what=what)
                o = out_scalar(nxt[1][3])
                p.add_op(_lib.OP_TRAP_PICKOFF, dst=o.sreg, src=src.slot, io=char_of(nxt[1][2]), ip=(*ints, trap_ops[fn]), sp=(t_in,))
                skip.add(si + 1)
                release(src, si + 1)
                continue
            # fusion: the trapezoid only feeds one min_max and / or one time_point_thresh (the t0 chain of the LEGEND recipes:
            # asym_trap_filter -> min_max -> time_point_thresh) and is not an output -> it is never stored.  The fused op is emitted
            # where the last of the two stands, so their scalar operands (a threshold computed in between) are ready
            users = [sj for sj, (f2, a2, _k) in enumerate(steps) if sj > si and sj not in skip and any(wf_of(x) is dst for x in a2)]
            kinds = [steps[sj][0] for sj in users]
            plain = all(steps[sj][1][0] is dst for sj in users)  # (not through a slice)
            pick_ok = all(char_of(steps[sj][1][2]) != ord("s") for sj in users if steps[sj][0] == "fixed_time_pickoff")
            if (users and plain and pick_ok and dst.name not in out_names
                    and sorted(kinds) in (["min_max"], ["time_point_thresh"], ["min_max", "time_point_thresh"], ["amax"], ["amax", "fixed_time_pickoff"])
                    and not any(isinstance(x, tuple) and x[0] == "slice" and x[1] is dst for _f, a2, _k in steps for x in a2)):
                pending_reduce[dst.name] = {"src": src, "ints": ints, "kind": trap_ops[fn], "emit_at": max(users), "mm_first": -1}
                last_use[src.name] = max(last_use.get(src.name, si), max(users))
                continue
            dst.slot = new_slot(src.length)
            p.add_op(trap_ops[fn], dst=dst.slot, src=src.slot, ip=ints)
            release(src, si)
        elif fn in ("min_max", "time_point_thresh", "amax", "fixed_time_pickoff") and isinstance(args[0], Var) and args[0].name in pending_reduce:
            pr = pending_reduce[args[0].name]
            if fn == "fixed_time_pickoff":  # trapEftp beside trapEmax: the samples around the pick-off time are captured in the same pass
                pr["pick"] = (scalar_operand(args[1], args, what=what), char_of(args[2]), out_scalar(args[3]))
            elif fn == "amax":  # numpy.amax of a trapezoid (trapEmax): the a_max of the same reduction (NaN in, NaN out in both)
                pr["mm_first"] = p.add_sregs(4)
                pr["amax_only"] = pr["kind"] != _lib.OP_ASYM_TRAP
                if not isinstance(args[2], Var):
                    raise ProcessingChainError("numpy.amax output must be a variable name")
                args[2].kind, args[2].sreg = "scalar", pr["mm_first"] + 3
            elif fn == "min_max":
                pr["mm_first"] = p.add_sregs(4)
                for k, a in enumerate(args[1:5]):
                    if not isinstance(a, Var):
                        raise ProcessingChainError("min_max outputs must be variable names")
                    a.kind, a.sreg = "scalar", pr["mm_first"] + k
            else:
                pr["tpt"] = (tuple(scalar_operand(a, args, what=what) for a in args[1:4]), out_scalar(args[4]))
            if si == pr["emit_at"]:
                sp, o = pr.get("tpt", ((), None))
                code = pr["kind"] | ((1 << 30) if pr.get("amax_only") else 0)
                if "pick" in pr:
                    t_in, mode, po = pr["pick"]
                    code |= (mode << 8) | ((po.sreg + 1) << 16)
                    sp = tuple(sp) + (Scalar.const(0.0),) * (3 - len(sp)) + (t_in,)
                p.add_op(_lib.OP_TRAP_REDUCE, dst=pr["mm_first"], src=pr["src"].slot, io=(o.sreg if o is not None else -1),
                         ip=(*pr["ints"], code), sp=sp)
                release(pr["src"], si)
                del pending_reduce[args[0].name]
        elif fn == "fixed_time_pickoff":
            src = ensure_loaded(args[0], si)
            o = out_scalar(args[3])
            p.add_op(_lib.OP_PICKOFF, dst=o.sreg, src=src.slot, ip=(char_of(args[2]),), sp=(scalar_operand(args[1], args, what=what),))
            release(src, si)
        elif fn == "time_point_thresh":
            src = ensure_loaded(args[0], si)
            sp = tuple(scalar_operand(a, args, what=what) for a in args[1:4])
            o = out_scalar(args[4])
            p.add_op(_lib.OP_TIME_POINT_THRESH, dst=o.sreg, src=src.slot, sp=sp)
            release(src, si)
        elif fn == "interpolated_time_point_thresh":
            src = ensure_loaded(args[0], si)
            walk = scalar_operand(args[3], args, integer=True, what=what)
            sp = (scalar_operand(args[1], args, what=what), scalar_operand(args[2], args, what=what), Scalar.const(float(walk)))
            o = out_scalar(args[5])
            p.add_op(_lib.OP_INTERP_TIME_POINT_THRESH, dst=o.sreg, src=src.slot, ip=(char_of(args[4]),), sp=sp)
            release(src, si)
        elif fn == "min_max":
            src = ensure_loaded(args[0], si)
            first = p.add_sregs(4)
            for k, a in enumerate(args[1:5]):
                if not isinstance(a, Var):
                    raise ProcessingChainError("min_max outputs must be variable names")
                a.kind, a.sreg = "scalar", first + k
            p.add_op(_lib.OP_MIN_MAX, dst=first, src=src.slot)
            release(src, si)
        elif fn in ("windower", "avg_current"):
            src = ensure_loaded(args[0], si)
            dst = out_wf(args[2], None, src)
            if dst.length is None:
                raise ProcessingChainError(f"{fn}: declare the output as name(length, 'f')")
            dst.slot = new_slot(dst.length)
            p.add_op(_lib.OP_WINDOWER if fn == "windower" else _lib.OP_AVG_CURRENT, dst=dst.slot, src=src.slot,
                     sp=(scalar_operand(args[1], args, what=what),))
            release(src, si)
        elif fn == "upsampler":
            src = ensure_loaded(args[0], si)
            dst = out_wf(args[2], None, src)
            if dst.length is None:
                raise ProcessingChainError("upsampler: declare the output as name(length, 'f')")
            dst.slot = new_slot(dst.length)
            p.add_op(_lib.OP_UPSAMPLER, dst=dst.slot, src=src.slot, sp=(scalar_operand(args[1], args, what=what),))
            release(src, si)
        elif fn == "moving_window_multi":
            src = ensure_loaded(args[0], si)
            num = scalar_operand(args[2], args, integer=True, what=what)
            typ = scalar_operand(args[3], args, integer=True, what=what)
            dst = out_wf(args[4], src.length, src)
            win = args[1]
            chunk = -(-(-(-src.length // 64)) // 16) * 16  # samples of a waveform per lane (dsp_chain_create: a multiple of 16)
            if (last_use.get(src.name, -1) <= si and num >= 1 and isinstance(win, (int, float, np.integer, np.floating)) and not isinstance(win, Quantity)
                    and float(win) == int(win) and 1 <= int(win) <= chunk):
                # in place: a source nobody reads again is overwritten pass by pass; only the ends of the lanes' chunks (64 x window
                # samples) are kept aside.  One waveform instead of two in LDS for the averaged current of the Ge recipes (22 + 13 kB
                # instead of 43): with that the whole recipe fits four times into a CU instead of three
                dst.slot = src.slot
                side = new_slot(64 * int(win))
                p.add_op(_lib.OP_MOVING_WINDOW_MULTI, dst=dst.slot, src=src.slot, ip=(typ, num, side, 1), sp=(scalar_operand(win, args, what=what),))
                if side not in free_slots:
                    free_slots.append(side)
                continue
            dst.slot = new_slot(src.length)
            # ping-pong target of the passes before the last: with an odd number of windows the first pass goes source -> target, so a
            # source nobody reads again serves
            own = num > 1 and not (num % 2 == 1 and last_use.get(src.name, -1) <= si)
            tmp = new_slot(src.length) if own else (src.slot if num > 1 else dst.slot)
            p.add_op(_lib.OP_MOVING_WINDOW_MULTI, dst=dst.slot, src=src.slot, ip=(typ, num, tmp), sp=(scalar_operand(args[1], args, what=what),))
            if own and tmp not in free_slots:
                free_slots.append(tmp)
            release(src, si)
        elif fn == "trap_pickoff":
            src = ensure_loaded(args[0], si)
            ints = [scalar_operand(a, args, integer=True, what=what) for a in args[1:3]]
            o = out_scalar(args[4])
            p.add_op(_lib.OP_TRAP_WINDOW_PICKOFF, dst=o.sreg, src=src.slot, ip=tuple(ints), sp=(scalar_operand(args[3], args, what=what),))
            release(src, si)
        elif fn == "mean_below_threshold":
            src = ensure_loaded(args[0], si)
            o = out_scalar(args[2])
            p.add_op(_lib.OP_MEAN_BELOW, dst=o.sreg, src=src.slot, sp=(scalar_operand(args[1], args, what=what),))
            release(src, si)
        elif fn == "linear_slope_fit":
            a0, view = args[0], (0, 0)
            if isinstance(a0, tuple) and a0[0] == "slice" and not a0[1].is_input:  # a window of an intermediate: read in place
                a0, view = a0[1], (a0[2], a0[3] - a0[2])
            src = ensure_loaded(a0, si)
            first = p.add_sregs(4)
            for k, a in enumerate(args[1:5]):
                if not isinstance(a, Var):
                    raise ProcessingChainError("linear_slope_fit outputs must be variable names")
                a.kind, a.sreg = "scalar", first + k
            p.add_op(_lib.OP_LINEAR_SLOPE_FIT, dst=first, src=src.slot, ip=view)
            release(src, si)
        elif fn == "amax":
            src = ensure_loaded(args[0], si)
            o = out_scalar(args[2])
            p.add_op(_lib.OP_AMAX, dst=o.sreg, src=src.slot)
            release(src, si)
        elif fn == "discrete_wavelet_transform":
            src = ensure_loaded(args[0], si)
            level = scalar_operand(args[1], args, integer=True, what=what)
            wt, part = char_of(args[2]), char_of(args[3])
            if wt not in (ord("h"), ord("d")):
                raise NotImplementedError("only the Haar wavelet ('h' / 'd') is implemented on the device")
            dst = out_wf(args[4], None, src)
            if dst.length is None:
                raise ProcessingChainError("discrete_wavelet_transform: declare the output as name(length, 'f')")
            dead = last_use.get(src.name, -1) <= si and not src.is_input or (src.is_input and last_use.get(src.name, -1) <= si)
            scratch = src.slot if dead else new_slot(src.length)
            dst.slot = new_slot(dst.length)
            p.add_op(_lib.OP_DWT_HAAR, dst=dst.slot, src=src.slot, ip=(level, part, scratch))
            if dead:
                release(src, si)
            elif scratch not in free_slots:
                free_slots.append(scratch)
        elif fn in ("convolve_wf", "fft_convolve_wf"):
            src = ensure_loaded(args[0], si)
            taps = args[1]
            if not (isinstance(taps, Var) and taps.kind == "taps"):
                raise NotImplementedError(f"{fn}: the kernel must be a constant computed in the recipe (cusp_filter / zac_filter)")
            if taps.io is None:  # (zeros after the taps up to a multiple of the FIR op's tap block: its fast path then covers every tap)
                padded = -(-taps.length // 16) * 16
                taps.io = p.add_io(f"taps:{taps.name}", _lib.IO_TAPS, ft, padded, 0, 0)
                consts[f"taps:{taps.name}"] = np.concatenate([taps.const.astype(ft), np.zeros(padded - taps.length, dtype=ft)])
            dst = out_wf(args[3], None, src)
            if dst.length is None:
                raise ProcessingChainError(f"{fn}: declare the output as name(length, 'f')")
            has_nan = int(np.isnan(taps.const).any()) | (2 if np.isinf(taps.const).any() else 0)  # (bit 1: an infinite tap)
            # fusion: the filtered waveform's only consumer is one numpy.amax and it is not an output -> it is never stored
            users = [sj for sj, (f2, a2, _k) in enumerate(steps) if sj != si and any(wf_of(x) is dst for x in a2)]
            if (len(users) == 1 and steps[users[0]][0] == "amax" and steps[users[0]][1][0] is dst and dst.name not in out_names
                    and users[0] > si and users[0] not in skip):
                o = out_scalar(steps[users[0]][1][2])
                p.add_op(_lib.OP_CONVOLVE_AMAX, dst=o.sreg, src=src.slot, io=taps.io, ip=(char_of(args[2]), has_nan, int(dst.length), int(taps.length)))
                skip.add(users[0])
                last_use[src.name] = max(last_use.get(src.name, si), si)
                release(src, si)
                continue
            dst.slot = new_slot(dst.length)
            p.add_op(_lib.OP_CONVOLVE, dst=dst.slot, src=src.slot, io=taps.io,
                     ip=(char_of(args[2]), has_nan, int(_piecewise_constant(taps.const)), int(taps.length)))
            release(src, si)
        else:
            raise NotImplementedError(f"processor '{fn}' is not implemented on the device path")

    tb_out = {}
    for o in out_pars:
        v = b.vars.get(o)
        if o in island_out:  # written by the integer program, in its own type
            nat, col_dt = island_out[o]
            out_bind[f"out:{o}"] = (SimpleNamespace(name=o, dtype=col_dt), None)
            tb_out[o] = np.empty(n_rows, dtype=nat)
            continue
        if _is_wf(v) and isinstance(v, tuple):  # a named slice: of an input it is read straight from the rows, else copied out of its waveform
            v = ensure_loaded(v, len(steps))
        if v is None or v.kind in (None,):
            raise ProcessingChainError(f"output '{o}' was never computed")
        if v.kind == "const":
            c = np.asarray(v.const)  # a number, or a constant array ("a1": "[1, 2, 3]"): every row holds it
            tb_out[o] = np.broadcast_to(c, (n_rows, *c.shape)).copy()
            continue
        if v.kind == "taps":
            tb_out[o] = np.broadcast_to(v.const, (n_rows, v.length)).copy()
            continue
        if v.kind == "wf":
            if v.slot is None and v.is_input:  # (an input under another name, or astype of nothing: load it to store it)
                v = ensure_loaded(v, len(steps))
            if v.slot is None:
                raise ProcessingChainError(f"output waveform '{o}' was never computed")
            odt = np.dtype(np.bool_) if v.dtype == np.dtype(np.bool_) else ft
            io = p.add_io(f"out:{o}", _lib.IO_WF_OUT, odt, v.length)
            p.add_op(_lib.OP_STORE, src=v.slot, io=io)
            out_bind[f"out:{o}"] = (SimpleNamespace(name=o, dtype=odt), v.length)
            if getattr(v, "vector_len", None) is not None:
                vl = v.vector_len
                if not (isinstance(vl, Var) and vl.is_input):
                    raise NotImplementedError(f"output '{o}': vector_len must be the length of an input array (len(<input>))")
                vector_lens[o] = vl.source
            tb_out[o] = np.empty((n_rows, v.length), dtype=v.dtype if _is_int_dtype(v) and v.dtype.kind != "b" else odt)
        else:
            # a time coordinate is written in its unit, not in samples: (index + grid offset) * period (reference :1990-2014, get_buffer(unit))
            unit_ns = _time_unit_ns(v.unit)
            if v.is_coord is True and v.grid is not None and unit_ns is not None and not stage_mode:  # (a stage hands on sample indices)
                v = b.converted(v, Grid(unit_ns))
            if isinstance(v, Var) and v.sreg is None and (getattr(v, "aux_io", None) is not None or getattr(v, "ext_key", None) is not None):
                src_op = scalar_operand(v, [], what=f"output {o}")  # (stores read registers)
                v.sreg = p.add_sregs(1)
                p.add_op(_lib.OP_SCALAR_FUNC, dst=v.sreg, ip=(_lib.FN_COPY,), sp=(src_op, Scalar.const(0.0), Scalar.const(0.0)))
            if isinstance(v, Var) and v.sreg is None:
                if v.is_input:
                    tb_out[o] = _column(b.tb_in, v.source)
                    continue
                raise ProcessingChainError(f"output '{o}' was never computed")
            reg = scalar_operand(v, [], what=f"output {o}")
            odt = np.dtype(np.bool_) if getattr(v, "dtype", None) == np.dtype(np.bool_) else ft
            io = p.add_io(f"out:{o}", _lib.IO_SCALAR_OUT, odt)
            p.add_op(_lib.OP_STORE_SCALAR, io=io, ip=(reg.index,))
            out_bind[f"out:{o}"] = (SimpleNamespace(name=o, dtype=odt), None)
            tb_out[o] = np.empty(n_rows, dtype=v.dtype if isinstance(v, SExpr) and _is_int_dtype(v) and v.dtype.kind != "b" else odt)
    aux_desc = []
    for gi, g in enumerate(aux):
        src = g["src"]
        col = _column(b.tb_in, src.source)
        wf_bind = next((nm for nm, v in in_bind.items() if isinstance(v, Var) and v.kind == "wf" and v.source == src.source and v.offset == 0
                        and v.length == src.length), None)
        if wf_bind is None:  # (nothing in the program reads the whole row: bind it for the fit alone)
            wf_bind = f"in:{src.name}:fit{gi}"
            p.add_io(wf_bind, _lib.IO_WF_IN, col.dtype, src.length, 0, col.shape[1])
            in_bind[wf_bind] = src
        sub_bind, sub_const, sub_code = None, 0.0, _lib.F32
        if isinstance(g["sub"], Var):
            sub_bind = p.io[scalar_operand(g["sub"], [], what="linear_slope_fit").index][0]
            sub_code = dtype_code(_column(b.tb_in, g["sub"].source).dtype)
        elif g["sub"] is not None:
            sub_const = float(g["sub"])
        aux_desc.append({"wf": wf_bind, "dtype": dtype_code(col.dtype), "itemsize": np.dtype(col.dtype).itemsize, "lo": g["lo"], "len": g["len"],
                         "stride": col.shape[1], "sub": sub_bind, "sub_dtype": sub_code, "sub_const": sub_const, "mode": g["mode"],
                         "tau": g["tau"], "fits": list(g["fits"]), "names": [f"aux:{gi}:{j}" for j in range(4 * len(g["fits"]))]})
    p.slots = slot_len
    if not p.ops:  # (every output is written by the integer program or handed through: the program is a placeholder)
        p.add_op(_lib.OP_SCALAR_AFFINE, dst=p.add_sregs(1), sp=(Scalar.const(0.0), Scalar.const(0.0), Scalar.const(0.0)))
    if len(p.ops) > _lib.MAX_OPS or len(p.slots) > _lib.MAX_SLOTS or len(p.io) > _lib.MAX_IO or p.n_sregs > _lib.MAX_SREGS:
        raise NotImplementedError("recipe is too large for one device chain (ops/slots/bindings limit)")
    for st in stages:  # columns of the input table that only a stage reads are linked like the program's own
        for nm, v in st["in_vars"].items():
            in_bind.setdefault(nm, v)
    tail = None
    if not stage_mode and os.environ.get("DSPEED_HIP_NO_SCALAR_TAIL", "0") != "1":
        tail = _split_scalar_tail(p, ft)
    if not stage_mode and os.environ.get("DSPEED_HIP_NO_SCALAR_HEAD", "0") != "1":  # (behind the tail's cut: what is only stored is the tail's)
        head = _split_scalar_head(p, ft, ext_alias)
        if head is not None:
            stages.append(head)
    walks = None
    if not stage_mode and os.environ.get("DSPEED_HIP_NO_WALKS_BEHIND", "0") != "1":
        walks = _split_walks(p, ft)
    from .processing_chain import ProcessingChain  # (the runtime imports this module)

    chain = ProcessingChain(p, in_bind, out_bind, consts, n_rows, proc_strings, ft, aux_desc, stages=stages, ext_alias=ext_alias, tail=tail, walks=walks)
    chain.vector_lens = vector_lens  # variable-length outputs -> the input column that holds their per-event lengths
    return chain, tb_out


_SCALAR_OPS = (_lib.OP_SCALAR_AFFINE, _lib.OP_SCALAR_DIV, _lib.OP_SCALAR_CONVERT, _lib.OP_SCALAR_FUNC, _lib.OP_STORE_SCALAR)
#: a tail is cut off when it has at least this many ops (a launch and a column per handed-over register have to pay for themselves)
SCALAR_TAIL_MIN_OPS = 8


#: a head is split off when it takes at least this many ops out of the program
SCALAR_HEAD_MIN_OPS = 2


def _split_scalar_head(p: Program, ft, ext_alias: dict):
    """The counterpart of the scalar tail at the other end: arithmetic between per-event values that needs nothing the program computes --
    pick-off times from the t0 estimate a stage left in HBM, thresholds from fit results -- leaves the program and runs ahead of it with a row
    per lane (dsp_scalar.hip), as one more stage; the program reads the results as columns.  On the interpreter an op costs a row's wavefront
    about 1 700 cycles whatever it computes (a lone wavefront issues its ~ 150 dispatch instructions one per ~ 12 cycles: measured 45 us per
    op and 65 536 rows), and a program of 8192-sample rows holds one row per SIMD.  ``p`` is changed in place; returns the stage's description or
    None.  Registers may be reused along the program: what an op reads is tracked by position."""
    if not p.slots:
        return None  # (no waveform: the program is a row-per-lane program already)
    from_head = {}          # register -> (column name, index into `made`) while the register's current value comes from a head op
    head_ops, made = [], []  # made: [column name, register, read by the program?]
    new_ops = []

    def reads_of(opcode, ip, sp):
        r = [a.index for a in sp if a.kind == _lib.ARG_REG]
        if opcode == _lib.OP_STORE_SCALAR:
            r.append(ip[0])
        return r

    def writes_of(opcode, dst, io, ip):
        if opcode == _lib.OP_STORE_SCALAR or opcode in (_lib.OP_LOAD, _lib.OP_STORE):
            return []
        if opcode in (_lib.OP_MIN_MAX,):
            return [dst + k for k in range(4)]
        if opcode == _lib.OP_TRAP_REDUCE:  # (its extremes, the maximum alone, a pick-off of the same trapezoid: include/dspeed_hip.h)
            w = [dst + k for k in range(4)] if dst >= 0 else []
            if io >= 0:
                w.append(io)
            if ((ip[3] >> 16) & 0x3fff) - 1 >= 0:
                w.append(((ip[3] >> 16) & 0x3fff) - 1)
            return w
        return [dst]

    for k, (opcode, dst, src, io, ip, sp) in enumerate(p.ops):
        is_head = (opcode in _SCALAR_OPS and opcode != _lib.OP_STORE_SCALAR
                   and all(a.kind in (_lib.ARG_CONST, _lib.ARG_INPUT) or (a.kind == _lib.ARG_REG and a.index in from_head) for a in sp))
        if is_head:
            head_ops.append((opcode, dst, src, io, ip, sp))
            made.append([f"head:r{dst}.{len(made)}", dst, False])
            from_head[dst] = len(made) - 1
            continue
        # an op that stays: what it reads of the head's results arrives as a column
        sp2 = []
        for a in sp:
            if a.kind == _lib.ARG_REG and a.index in from_head:
                made[from_head[a.index]][2] = True
                sp2.append(("head", from_head[a.index]))
            else:
                sp2.append(a)
        if opcode == _lib.OP_STORE_SCALAR and ip[0] in from_head:  # a stored value needs its register: a copy of the column
            made[from_head[ip[0]]][2] = True
            new_ops.append(("copy", ip[0], from_head[ip[0]]))
            del from_head[ip[0]]
        new_ops.append((opcode, dst, src, io, ip, tuple(sp2)))
        for r in writes_of(opcode, dst, io, ip):
            from_head.pop(r, None)
    if len(head_ops) < SCALAR_HEAD_MIN_OPS or not any(m[2] for m in made):
        return None
    h = Program()
    h.n_sregs = p.n_sregs
    io_map = {}

    def head_io(idx):
        if idx not in io_map:
            name, kind, code, length, offset, stride = p.io[idx]
            io_map[idx] = h.add_io(name, kind, code, length, offset, stride)
        return io_map[idx]

    for (opcode, dst, src, io, ip, sp), (name, r, used) in zip(head_ops, made):
        h.add_op(opcode, dst=dst, src=src, io=io, ip=ip, sp=tuple(Scalar.input(head_io(a.index)) if a.kind == _lib.ARG_INPUT else a for a in sp))
        if used:  # (stored right behind the op that made it: the head may write the register again)
            h.add_op(_lib.OP_STORE_SCALAR, io=h.add_io("out:" + name, _lib.IO_SCALAR_OUT, ft), ip=(r,))
    col_io = {}

    def column(j):
        if j not in col_io:
            col_io[j] = p.add_io("in:" + made[j][0], _lib.IO_SCALAR_IN, ft)
            ext_alias["in:" + made[j][0]] = "in:" + made[j][0]
        return col_io[j]

    del p.ops[:]
    for entry in new_ops:
        if entry[0] == "copy":
            _c, r, j = entry
            p.add_op(_lib.OP_SCALAR_FUNC, dst=r, ip=(_lib.FN_COPY,), sp=(Scalar.input(column(j)), Scalar.const(0.0), Scalar.const(0.0)))
            continue
        opcode, dst, src, io, ip, sp = entry
        p.add_op(opcode, dst=dst, src=src, io=io, ip=ip, sp=tuple(Scalar.input(column(a[1])) if isinstance(a, tuple) else a for a in sp))
    if len(p.io) > _lib.MAX_IO or len(h.io) > _lib.MAX_IO:
        raise NotImplementedError("recipe is too large for one device chain (ops/slots/bindings limit)")
    names = [io[0] for io in h.io if io[1] == _lib.IO_SCALAR_IN]
    return {"what": "per-event arithmetic ahead of the program", "program": h, "consts": {}, "in_vars": {}, "alias": {n: ext_alias.get(n, n) for n in names},
            "outs": [("out:" + name, "in:" + name, None) for name, _r, used in made if used], "chain": None, "bufs": {}}


#: the rise-time walks leave a program whose waveform has at least this many samples (LDS then holds a handful of rows per CU)
WALKS_BEHIND_MIN_SAMPLES = 4096
#: walks the reductions kernel takes in one program (DSP_REDUCE_WALKS of csrc/dsp_program.h)
WALKS_BEHIND_MAX = 6


def _split_walks(p: Program, ft):
    """The threshold walks of a program that only reads its one long waveform -- the rise times of the Ge recipes: time_point_thresh at 0.99 /
    0.9 / 0.5 / 0.1 of the trapezoid's maximum, each from where the one before ended -- run behind the program as a launch of their own
    (dsp_reduce.hip: a wavefront per row straight off HBM, thousands of rows in flight) instead of in it, where LDS holds four 8192-sample
    rows per CU and the row's wavefronts wait for the chain *trapezoid -> maximum -> walks* (2.1 of 14.2 ms of the Ge recipe's pass on the
    synthetic rows, whose tp_100 walk runs to the end of the waveform; 0.6 - 0.9 on pulses with a rise time).  The per-event values the walks
    start from or scale (the maximum) are handed over as columns (``walk:r<k>``); the stores of the walks' results move along.  ``p`` is
    changed in place; returns {"program", "handover"} or None."""
    from .chain import plan

    ops = p.ops
    if len(p.slots) != 1 or not ops or ops[0][0] != _lib.OP_LOAD or p.slots[0] < WALKS_BEHIND_MIN_SAMPLES:
        return None
    slot = ops[0][1]
    TPT, AFF, STS = _lib.OP_TIME_POINT_THRESH, _lib.OP_SCALAR_AFFINE, _lib.OP_STORE_SCALAR

    def writes_of(op):
        opcode, dst, _src, io, ip, _sp = op
        if opcode in (STS, _lib.OP_LOAD, _lib.OP_STORE):
            return []
        if opcode == _lib.OP_MIN_MAX:
            return [dst + k for k in range(4)]
        if opcode == _lib.OP_TRAP_REDUCE:
            w = [dst + k for k in range(4)] if dst >= 0 else []
            if io >= 0:
                w.append(io)
            if ((ip[3] >> 16) & 0x3fff) - 1 >= 0:
                w.append(((ip[3] >> 16) & 0x3fff) - 1)
            return w
        return [dst]

    n_writes = {}
    for op in ops:
        for r in writes_of(op):
            n_writes[r] = n_writes.get(r, 0) + 1
    writer = {r: k for k, op in enumerate(ops) for r in writes_of(op)}

    def readers(r):
        return [(k, j) for k, op in enumerate(ops) for j, a in enumerate(op[5]) if a.kind == _lib.ARG_REG and a.index == r] + \
               [(k, -1) for k, op in enumerate(ops) if op[0] == STS and op[4][0] == r]

    number = lambda a: a.kind == _lib.ARG_CONST  # noqa: E731
    walks = [k for k, op in enumerate(ops) if op[0] == TPT and op[2] == slot and number(op[5][2]) and op[5][2].value in (0.0, 1.0)]
    scales = set()
    changed = True
    while changed:  # a walk stays if its result is read by anything but a store or a moving walk's start; a fraction if anything else reads it
        changed = False
        for k in list(walks):
            d = ops[k][1]
            fine = n_writes.get(d, 0) == 1 and all(j == -1 or (kk in walks and j == 1) for kk, j in readers(d))
            for j in (0, 1):  # threshold, start
                a = ops[k][5][j]
                if a.kind != _lib.ARG_REG:
                    continue
                src = writer.get(a.index)
                if src is None or n_writes.get(a.index, 0) != 1 or src > k:
                    fine = False
                elif ops[src][0] == TPT and src in walks:
                    fine = fine and j == 1
                elif (j == 0 and ops[src][0] == AFF and number(ops[src][5][1]) and number(ops[src][5][2]) and ops[src][5][2].value == 0.0
                      and ops[src][5][0].kind in (_lib.ARG_REG, _lib.ARG_INPUT) and all(kk in walks and jj == 0 for kk, jj in readers(a.index))):
                    x = ops[src][5][0]
                    if x.kind == _lib.ARG_REG and (n_writes.get(x.index, 0) != 1 or writer[x.index] > src or ops[writer[x.index]][0] in (TPT, AFF)):
                        fine = False
                elif ops[src][0] in (TPT, AFF):
                    fine = False  # (a walk that stays, or arithmetic of another form: the program's)
            if not fine:
                walks.remove(k)
                changed = True
    if not 2 <= len(walks) <= WALKS_BEHIND_MAX:
        return None
    for k in walks:
        a = ops[k][5][0]
        if a.kind == _lib.ARG_REG and ops[writer[a.index]][0] == AFF:
            scales.add(writer[a.index])
    moved = set(walks) | scales
    walk_regs = {ops[k][1] for k in walks}
    stores = [k for k, op in enumerate(ops) if op[0] == STS and op[4][0] in walk_regs]
    # what the walks read of the program: registers made by ops that stay -> columns
    handed = []
    for k in sorted(moved):
        for a in ops[k][5][:2] if ops[k][0] == TPT else ops[k][5][:1]:
            if a.kind == _lib.ARG_REG and writer[a.index] not in moved and a.index not in handed:
                handed.append(a.index)
    w = Program()
    w.n_sregs = p.n_sregs
    w.slots = list(p.slots)
    io_map = {}

    def w_io(idx):
        if idx not in io_map:
            name, kind, code, length, offset, stride = p.io[idx]
            io_map[idx] = w.add_io(name, kind, code, length, offset, stride)
        return io_map[idx]

    col = {r: w.add_io(f"walk:r{r}", _lib.IO_SCALAR_IN, ft) for r in handed}

    def operand(a):
        if a.kind == _lib.ARG_INPUT:
            return Scalar.input(w_io(a.index))
        if a.kind == _lib.ARG_REG and a.index in col:
            return Scalar.input(col[a.index])
        return a

    ld = ops[0]
    w.add_op(_lib.OP_LOAD, dst=ld[1], src=ld[2], io=w_io(ld[3]), ip=ld[4], sp=ld[5])
    for k in sorted(moved):
        opcode, dst, src, io, ip, sp = ops[k]
        w.add_op(opcode, dst=dst, src=src, io=io, ip=ip, sp=tuple(operand(a) for a in sp))
    for k in stores:
        opcode, dst, src, io, ip, sp = ops[k]
        w.add_op(opcode, dst=dst, src=src, io=w_io(io), ip=ip, sp=sp)
    try:
        if "dsp_reduce_kernel" not in plan(w, ft)["kernel"]:
            return None
    except Exception:  # noqa: BLE001  (a form the planner refuses: the walks stay where they were)
        return None
    gone = moved | set(stores)
    kept = [op for k, op in enumerate(ops) if k not in gone]
    if not any(op[0] not in (_lib.OP_LOAD, STS) for op in kept):
        return None  # (nothing but the walks read the waveform: the program is theirs)
    del ops[:]
    ops.extend(kept)
    handover = []
    for r in handed:
        handover.append(f"walk:r{r}")
        p.add_op(STS, io=p.add_io(f"walk:r{r}", _lib.IO_SCALAR_OUT, ft), ip=(r,))
    if len(p.io) > _lib.MAX_IO or len(w.io) > _lib.MAX_IO:
        raise NotImplementedError("recipe is too large for one device chain (ops/slots/bindings limit)")
    return {"program": w, "handover": handover}


def _split_scalar_tail(p: Program, ft):
    """Cut the all-scalar tail off a program: the ops after the last one that touches a waveform -- arithmetic between per-event values,
    unit conversions, stores; two thirds of a whole recipe's ops -- become a program of their own that ``dsp_chain_create`` gives to the
    row-per-lane kernel (dsp_scalar.hip: 64 rows per interpreter dispatch instead of one).  The head stores every register the tail
    reads and does not make itself into a column (``tail:r<k>``), the tail starts by loading them.  ``p`` is changed in place; returns the
    tail's description ({"program", "handover": [binding names]}) or None when the program has no tail worth a launch."""
    ops = p.ops
    k = len(ops)
    while k > 0 and ops[k - 1][0] in _SCALAR_OPS:
        k -= 1
    if k == 0 or len(ops) - k < SCALAR_TAIL_MIN_OPS:
        return None
    tail_ops = ops[k:]
    written, live_in = set(), []
    for opcode, dst, _src, _io, ip, sp in tail_ops:
        reads = [a.index for a in sp if a.kind == _lib.ARG_REG]
        if opcode == _lib.OP_STORE_SCALAR:
            reads.append(ip[0])
        for r in reads:
            if r not in written and r not in live_in:
                live_in.append(r)
        if opcode != _lib.OP_STORE_SCALAR:
            written.add(dst)
    t = Program()
    t.n_sregs = p.n_sregs
    io_map = {}  # binding of the head -> the tail's copy of it

    def tail_io(idx):
        if idx not in io_map:
            name, kind, code, length, offset, stride = p.io[idx]
            io_map[idx] = t.add_io(name, kind, code, length, offset, stride)
        return io_map[idx]

    handover = []
    del ops[k:]
    for r in live_in:
        name = f"tail:r{r}"
        handover.append(name)
        p.add_op(_lib.OP_STORE_SCALAR, io=p.add_io(name, _lib.IO_SCALAR_OUT, ft), ip=(r,))
        t.add_op(_lib.OP_SCALAR_FUNC, dst=r, ip=(_lib.FN_COPY,),
                 sp=(Scalar.input(t.add_io(name, _lib.IO_SCALAR_IN, ft)), Scalar.const(0.0), Scalar.const(0.0)))
    for opcode, dst, src, io, ip, sp in tail_ops:
        sp2 = tuple(Scalar.input(tail_io(a.index)) if a.kind == _lib.ARG_INPUT else a for a in sp)
        t.add_op(opcode, dst=dst, src=src, io=tail_io(io) if opcode == _lib.OP_STORE_SCALAR else io, ip=ip, sp=sp2)
    if len(t.io) > _lib.MAX_IO or len(p.io) > _lib.MAX_IO or len(p.ops) > _lib.MAX_OPS:
        raise NotImplementedError("recipe is too large for one device chain (ops/slots/bindings limit)")
    return {"program": t, "handover": handover}


