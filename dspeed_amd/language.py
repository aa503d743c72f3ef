"""The reference's argument language and unit system, evaluated into chain variables: ``Quantity`` (times), ``Grid`` (the reference's
CoordinateGrid), ``Var`` / ``SExpr`` (the subset of ProcChainVar the device path needs), the loop-selection rules of ``ProcessorManager`` (NumPy's
``can_cast`` on the variables' types) and ``_Builder``, which turns a recipe argument -- ``round(tp_0_est + 10*us, wf.grid)``, ``where(a < b, a, b)``,
``wf[10:100:2]`` -- into variables and the steps that compute them (reference src/dspeed/processing_chain.py:718-1130, 1193-1430, 1556-1732,
1806-1908).  ``dspeed_amd.compiler`` turns the steps into device programs, ``dspeed_amd.processing_chain`` runs them."""
from __future__ import annotations

import ast
import logging
import math

import numpy as np

from . import _lib
from .errors import DSPFatal, ProcessingChainError
from .recipe import LANGUAGE_CALLS as _CALLS

log = logging.getLogger("dspeed")  # (the reference's logger name: processing_chain.py:33)

_UNITS_NS = {"ns": 1.0, "us": 1e3, "ms": 1e6, "s": 1e9}


class Quantity(float):
    """A time in nanoseconds (the only dimension hot-path recipes use); ``unit`` is the unit it was written in."""

    def __new__(cls, value, unit="ns"):
        q = float.__new__(cls, value)
        q.unit = unit
        return q

    def __repr__(self):
        return f"{float(self):g}*ns"


class WaveformInput:
    """Input column with sampling information, the role of ``lgdo.WaveformTable`` (values, dt, t0) in the reference
    (processing_chain.py:2263-2360).  ``dt`` and ``t0`` in nanoseconds; ``t0`` is one number or one value per row (an ndarray or
    DeviceArray), the time of sample 0 -- the offset of the waveform's coordinate grid."""

    def __init__(self, values, dt: float = 16.0, t0=0.0):
        self.values = values
        self.dt = float(dt)
        self.t0 = float(t0) if isinstance(t0, (int, float, np.integer, np.floating)) else t0

    def __len__(self):
        return len(self.values)


class Grid:
    """The reference's CoordinateGrid (processing_chain.py:67-144): sampling period and the time of sample 0, both in ns; the offset
    is a constant plus, for inputs with one t0 per row, a per-event variable holding ns."""

    __slots__ = ("period", "offset", "offset_var")

    def __init__(self, period, offset=0.0, offset_var=None):
        self.period, self.offset, self.offset_var = float(period), float(offset), offset_var

    def __eq__(self, other):  # (a variable offset compares by identity, reference :107-113)
        return (isinstance(other, Grid) and self.period == other.period and self.offset == other.offset
                and self.offset_var is other.offset_var)

    __hash__ = None

    def key(self):
        return (self.period, self.offset, id(self.offset_var))

    def shifted(self, first_sample: int, step: int = 1) -> "Grid":
        """grid of wf[first_sample::step] (reference :1032-1054)"""
        return Grid(self.period * step, self.offset + first_sample * self.period, self.offset_var)

    def __repr__(self):
        off = f"{self.offset:g}" + (f"+{self.offset_var.name}" if self.offset_var is not None else "")
        return f"({self.period:g}*ns,{off})"


def _time_unit_ns(unit):
    """ns per `unit` if it is a time unit, else None (what ureg.is_compatible_with(grid.period, unit) decides, reference :1709-1713)"""
    if isinstance(unit, Quantity):
        return float(unit)
    if isinstance(unit, str):
        return _UNITS_NS.get(unit)
    return None


class Var:
    """A chain variable (the subset of ProcChainVar, processing_chain.py:147-377, that the device path needs).  ``unit``,
    ``is_coord`` (None = the reference's ``auto``) and ``grid`` carry the coordinate information: a per-event variable with
    ``is_coord`` holds a sample index of ``grid`` and is converted when a processor working on another grid, or an output column in
    time units, reads it."""

    def __init__(self, name, kind, length=None, dtype=np.float32, period=None, const=None, source=None, offset=0, grid=None,
                 unit=None, is_coord=None):
        self.name = name
        self.kind = kind          # 'wf' | 'scalar' | 'const' | 'char' | 'taps'
        self.length = length      # samples (wf/taps)
        self.dtype = np.dtype(dtype) if dtype is not None else None
        self.grid = grid if grid is not None else (Grid(period) if period is not None else None)
        self.unit = unit
        self.is_coord = is_coord
        self.const = const        # python value for constants, ndarray for taps
        self.source = source      # input column name for chain inputs
        self.offset = offset      # first sample for sliced inputs
        self.is_input = source is not None
        self.slot = None
        self.sreg = None
        self.io = None
        self.vector_len = None    # per-event number of valid samples of a variable-length array (reference ProcChainVar.vector_len, :164-208)

    @property
    def period(self):  # ns per sample
        return self.grid.period if self.grid is not None else None

    def __repr__(self):
        return f"<Var {self.name} {self.kind} len={self.length}>"


class SExpr:
    """A per-event value computed inside a recipe argument -- ``tp_0 + 8*us``, ``0.9*trapTmax``, ``QDrift/trapTmax``,
    ``round(tp, wf.grid)`` -- or a coordinate conversion the chain inserts.  The reference adds a NumPy ufunc or a unit-conversion
    processor and a new ProcChainVar for each (processing_chain.py:832-917, 1193-1266, 1806-1908); here it becomes one scalar op
    when something first reads it."""

    kind = "scalar"
    is_input = False

    def __init__(self, op, args, name, unit=None, is_coord=None, grid=None, mode=0):
        self.op, self.args, self.name = op, tuple(args), name   # 'affine' (x, mul, add) | 'div' (a, b) | 'convert' (x, off_in, off_out, ratio) | 'func' (FN_*, a, b, c)
        self.unit, self.is_coord, self.grid, self.mode = unit, is_coord, grid, mode
        self.sreg = None
        self.io = None     # (op 'ext': the binding through which a program reads the column an integer program wrote)
        self.dtype = None  # np.bool_ for truth values ('func' results of comparisons, isnan, isfinite)

    def __repr__(self):
        return f"<SExpr {self.name}>"


def _is_scalar(a) -> bool:
    return isinstance(a, SExpr) or (isinstance(a, Var) and a.kind == "scalar")


# signatures of the supported processors: argument roles, in recipe order
#   w = waveform in, W = waveform out, s = float scalar in (const or per-event), i = int const, c = char const,
#   S = scalar out, t = taps in
_SIGS = {
    "bl_subtract": "wsW", "pole_zero": "wsW", "double_pole_zero": "wsssW", "trap_filter": "wiiW", "trap_norm": "wiiW",
    "asym_trap_filter": "wiiiW", "fixed_time_pickoff": "wscS", "time_point_thresh": "wsssS", "interpolated_time_point_thresh": "wssicS",
    "min_max": "wSSSS",
    "discrete_wavelet_transform": "wiccW", "convolve_wf": "wtcW", "fft_convolve_wf": "wtcW", "amax": "wiS",
    "mean_below_threshold": "wsS", "windower": "wsW", "avg_current": "wsW", "trap_pickoff": "wiisS",
    "upsampler": "wsW", "moving_window_multi": "wsiiW", "numpy_subtract": "wsW", "numpy_add": "wsW", "min_max_norm": "wssW", "linear_slope_fit": "wSSSS",
}
_SIGS.update({"sample": "wiS", "slice": "wiiW", "get": "wsS"})  # wf[i], wf[lo:hi:step], wf[variable] (reference :948-1071)


def _roles(fn) -> str:
    """argument roles of a step; the recipe language's element-wise steps carry theirs in the name: 'ew:' + one of w / s / c (unused) per
    operand"""
    if fn.startswith("ew:"):
        return "c" + fn[3:] + "W"
    return _SIGS.get(fn, "")


def _is_wf(a) -> bool:
    return (isinstance(a, Var) and a.kind == "wf") or (isinstance(a, tuple) and len(a) == 4 and a[0] == "slice")


def _wf_len(a):
    return a[3] - a[2] if isinstance(a, tuple) else a.length


def _is_int_dtype(a) -> bool:
    """does the variable select an integer ufunc loop in the reference (np.can_cast on its dtype, :1565-1572)"""
    v = a[1] if isinstance(a, tuple) else a
    dt = getattr(v, "dtype", None)
    return dt is not None and np.dtype(dt).kind in "iub"


_INT_LOOPS = "bBhHiIlLqQ"  # the integer signatures of numpy.add / subtract / multiply / floor_divide / negative, in the order of ufunc.types


def _all_bool(variables) -> bool:
    return bool(variables) and all(np.dtype((v[1] if isinstance(v, tuple) else v).dtype) == np.dtype(np.bool_) for v in variables)


def _int_loop_of(variables, src, loops=_INT_LOOPS):
    """The integer ufunc loop the reference picks for these variables: the first signature every variable can be cast to
    (np.can_cast per parameter, reference :1565-1572; the first one left, :1654-1664).  Constants do not take part: they are converted to
    the loop's type afterwards (:1765-1768).  ``loops``: the integer signatures of the ufunc in the order of its ``types`` (numpy.add's
    and its relatives' by default; ``where`` has its own, processors/where.py:11-20).  Truth values alone select NumPy's '??' loops where the
    ufunc has one (add, multiply: the callers' business) and the int8 loop otherwise (floor_divide)."""
    dts = [np.dtype((v[1] if isinstance(v, tuple) else v).dtype) for v in variables]
    c = next((c for c in loops if all(np.can_cast(d, c) for d in dts)), None)
    if c is None:  # int64 beside uint64: no integer signature takes both, NumPy goes on to the float64 one
        raise NotImplementedError(f"'{src}' mixes {' and '.join(sorted({d.name for d in dts}))}: NumPy's loop for them is the float64 one, which cannot "
                                  "hold them; cast one side (astype)")
    return np.dtype(c)


def _int_loop_const(c, dt, src, period=None):
    """a constant beside integer variables: the reference converts it to the loop's type, dtype.type(np.round(c)) (:1765-1768) -- a value
    outside the type wraps around (NumPy's conversion between its own integer scalars); a time counts periods of the processor's grid first
    (:1747-1764)"""
    if isinstance(c, Quantity):
        if period is None:
            raise ProcessingChainError(f"could not find valid conversion for {c!r} in '{src}'; CoordinateGrid is None")
        c = float(c) / period
    if dt == np.dtype(np.bool_):  # (not an integer type: dtype.type(c), the truth of the number)
        return float(bool(c))
    r = int(np.round(float(c)))
    if abs(r) > 2 ** 53:
        raise NotImplementedError(f"'{src}': the constant {c} beside integer variables is beyond 2^53")
    return float(int(np.array(r, dtype=np.int64).astype(dt)))


_BOOL_MINUS = ("numpy boolean subtract, the `-` operator, is not supported, use the bitwise_xor, the `^` operator, or the logical_xor "
               "function instead.")  # (what numpy.subtract / numpy.negative raise for truth values when the reference's processor first runs)
_WHERE_LOOPS = "BHILbhiq"  # processors/where.py:11-20: u1 u2 u4 u8 i1 i2 i4 i8 (then f4, f8)


_GENERATORS = ("cusp_filter", "zac_filter", "t0_filter", "moving_slope")
_MODULES = ("dspeed.processors", "dspeed_amd.processors", "numpy", "np")
_NUMPY_BINARY = {"add": ast.Add, "subtract": ast.Sub, "multiply": ast.Mult, "divide": ast.Div, "true_divide": ast.Div}
_ROUND_MODES = {"round": 1, "floor": 2, "ceil": 3, "trunc": 4}



def _column(tb, name):
    if name not in tb and name.endswith(".t0"):  # the per-row t0 of a WaveformInput
        return tb[name[:-3]].t0
    col = tb[name]
    return col.values if isinstance(col, WaveformInput) else col



class _Builder:
    def __init__(self, tb_in, db_dict):
        self.tb_in = tb_in if tb_in is not None else {}
        self.db = db_dict or {}
        self.vars: dict[str, Var] = {}
        self.steps = []  # (function name, [operands], recipe key)
        self.default_period = None
        self.cur_key = None   # recipe entry being added (the expression steps it creates carry its name)
        self._anon = 0        # counter behind the names of expression results
        self._conversions = {}  # (id(value), target grid key, rounding) -> SExpr: one conversion per variable and grid (reference :303-313)
        self.group_values = {}  # input column -> the integer it holds in the rows this chain is built for (GroupedProcessingChain)
        for name, col in self.tb_in.items():
            if isinstance(col, WaveformInput) and self.default_period is None:
                self.default_period = col.dt

    # ---- variables
    def input_var(self, name) -> Var:
        if name in self.vars:
            return self.vars[name]
        if name not in self.tb_in:
            raise ProcessingChainError(f"'{name}' not found in input table or recipe")
        col = self.tb_in[name]
        vals = col.values if isinstance(col, WaveformInput) else col
        shape, dtype = vals.shape, vals.dtype
        if len(shape) == 2:
            grid = None
            if isinstance(col, WaveformInput):  # (values, dt, t0) -> grid(dt, t0), reference :2277-2299
                if isinstance(col.t0, float):
                    grid = Grid(col.dt, col.t0)
                else:  # one t0 per row: a per-event variable in ns, itself a coordinate on the (1 ns, 0) grid
                    t0 = Var(f"{name}.t0", "scalar", None, col.t0.dtype, source=f"{name}.t0", grid=Grid(1.0), unit="ns", is_coord=True)
                    self.vars[t0.name] = t0
                    grid = Grid(col.dt, 0.0, t0)
            v = Var(name, "wf", shape[1], dtype, source=name, grid=grid, is_coord=False)
            if f"len({name})" in self.tb_in:  # a VectorOfVectors: rows padded to a common length + their true lengths (lgdo_io.RaggedColumn)
                self.vars[name] = v
                v.vector_len = self.input_var(f"len({name})")
        elif len(shape) == 1:
            v = Var(name, "scalar", None, dtype, source=name)
        else:
            raise ProcessingChainError(f"input '{name}' has unsupported shape {shape}")
        self.vars[name] = v
        return v

    # ---- coordinate conversions
    def offset_ns(self, grid: Grid):
        """the per-event offset of a grid with one t0 per row, in ns: t0, or t0 + start for a slice (reference :1039-1053)"""
        ns = grid.offset_var
        if grid.offset != 0.0:
            key = (id(ns), "shift", grid.offset)
            if key not in self._conversions:
                self._conversions[key] = SExpr("affine", (ns, 1.0, grid.offset), f"({ns.name}+{grid.offset:g}*ns)", "ns", True, Grid(1.0))
            ns = self._conversions[key]
        return ns

    def offset_in_periods(self, grid: Grid, period: float):
        """grid's offset in units of `period`: a number, or a per-event value (CoordinateGrid.get_offset, reference :126-136)"""
        if grid.offset_var is None:
            return grid.offset / period
        return self.converted(self.offset_ns(grid), Grid(period))

    def converted(self, v, to: Grid, rounding: int = 0):
        """v (a coordinate on v.grid) expressed on `to`: (v + offset_in) * period_ratio - offset_out, UnitConversionManager
        (reference :1806-1908) with unit_conversion.py:16-79."""
        if not rounding and v.grid == to:
            return v
        key = (id(v), to.key(), rounding)
        if key not in self._conversions:
            src = v.grid
            ratio = src.period / to.period
            off_in = self.offset_in_periods(src, src.period)
            off_out = self.offset_in_periods(to, to.period)
            name = f"{'convert' if not rounding else [k for k, m in _ROUND_MODES.items() if m == rounding][0]}({v.name}, {to})"
            self._conversions[key] = SExpr("convert", (v, off_in, off_out, ratio), name, v.unit, True, to, rounding)
        return self._conversions[key]

    # ---- expression evaluation
    def eval_arg(self, arg, want_new=None):
        """Turn a recipe argument into a Var / SExpr / number / Quantity / char.  ``want_new``: names this processor creates."""
        if not isinstance(arg, str):
            return arg
        tree = ast.parse(arg.strip(), mode="eval").body
        return self._eval(tree, arg, want_new or ())

    def _eval(self, n, src, new):
        if isinstance(n, ast.List):  # [1, 2, 3]: a constant array (reference :806-810), the same for every row
            return np.array(ast.literal_eval(src[n.col_offset:n.end_col_offset]))
        if isinstance(n, ast.Constant):
            if isinstance(n.value, str):
                return ("char", n.value)
            return n.value
        if isinstance(n, ast.Name):
            if n.id in _UNITS_NS:
                return Quantity(_UNITS_NS[n.id], n.id)
            if n.id in self.vars:
                v = self.vars[n.id]
                return v.const if isinstance(v, Var) and v.kind == "const" else v
            if n.id in new:
                v = Var(n.id, None)
                self.vars[n.id] = v
                return v
            return self.input_var(n.id)
        if isinstance(n, ast.UnaryOp) and isinstance(n.op, (ast.USub, ast.UAdd)):
            v = self._eval(n.operand, src, new)
            if _is_scalar(v):
                if isinstance(n.op, ast.UAdd):
                    return v
                if _is_int_dtype(v):  # numpy.negative's integer loops: 0 - v in the variable's type
                    if _all_bool([v]):
                        raise ProcessingChainError(f"'{src}': {_BOOL_MINUS.replace('subtract', 'negative')}")
                    dt = _int_loop_of([v], src)
                    return self._scalar_func(_lib.fn_int(_lib.FN_ISUB, dt), [0.0, v], f"(-{v.name})", v.unit, v.is_coord, v.grid, dt)
                return SExpr("affine", (v, -1.0, -0.0), f"(-{v.name})", v.unit, v.is_coord, v.grid)
            if _is_wf(v):
                if isinstance(n.op, ast.UAdd):
                    return v
                if _is_int_dtype(v):
                    if _all_bool([v]):
                        raise ProcessingChainError(f"'{src}': {_BOOL_MINUS.replace('subtract', 'negative')}")
                    dt = self._wide_wf_loop(_int_loop_of([v], src), _lib.FN_ISUB, [0.0, v], src)
                    return self._elementwise(_lib.fn_int(_lib.FN_ISUB, dt), [0.0, v], f"(-{self._nm(v)})", src, self._unit_of(v), dt)
                return self._elementwise(_lib.FN_NEG, [v], f"(-{self._nm(v)})", src, self._unit_of(v))
            if isinstance(v, (Var, tuple)):
                raise ProcessingChainError(f"cannot negate {v!r} in '{src}'")
            return -v if isinstance(n.op, ast.USub) else v
        if isinstance(n, ast.Compare):  # reference :919-946: the NumPy comparison as a processor, a bool variable
            if len(n.comparators) != 1:
                raise ProcessingChainError("Compound comparisons are not supported.")
            a, b2 = self._eval(n.left, src, new), self._eval(n.comparators[0], src, new)
            fn, sym = {ast.Lt: (_lib.FN_LT, "<"), ast.LtE: (_lib.FN_LE, "<="), ast.Gt: (_lib.FN_GT, ">"), ast.GtE: (_lib.FN_GE, ">="),
                       ast.Eq: (_lib.FN_EQ, "=="), ast.NotEq: (_lib.FN_NE, "!=")}.get(type(n.ops[0]), (None, None))
            if fn is None:
                raise ProcessingChainError(f"unsupported comparison in '{src}'")
            if not any(_is_wf(x) or _is_scalar(x) for x in (a, b2)):
                if any(isinstance(x, (Var, tuple, Grid)) for x in (a, b2)):
                    raise ProcessingChainError(f"cannot compare {a!r} and {b2!r} in '{src}'")
                return bool({"<": a < b2, "<=": a <= b2, ">": a > b2, ">=": a >= b2, "==": a == b2, "!=": a != b2}[sym])
            name = f"({self._nm(a)}{sym}{self._nm(b2)})"
            if _is_wf(a) or _is_wf(b2):
                return self._elementwise(fn, [a, b2], name, src, None, np.bool_)
            variables = [x for x in (a, b2) if _is_scalar(x)]
            if all(_is_int_dtype(x) for x in variables) and any(np.dtype(x.dtype).itemsize == 8 for x in variables):
                # 64-bit integers are compared as integers (NumPy's 'qq->?' / 'QQ->?' loops; a constant is converted to the loop's type,
                # :1765-1768): the comparison joins the integer program that holds them (_int_island)
                dt = _int_loop_of(variables, src)
                a, b2 = (x if _is_scalar(x) else _int_loop_const(x, dt, src) for x in (a, b2))
                return self._scalar_func(_lib.fn_int(fn, dt), [a, b2], name, None, False, None, np.bool_)
            return self._scalar_func(fn, [a, b2], name, None, False, None, np.bool_)
        if isinstance(n, ast.IfExp):  # a if condition else b  (reference :1073-1078)
            return self._where(self._eval(n.test, src, new), self._eval(n.body, src, new), self._eval(n.orelse, src, new), src)
        if isinstance(n, ast.BinOp):
            a, b = self._eval(n.left, src, new), self._eval(n.right, src, new)
            return self._binop(n.op, a, b, src)
        if isinstance(n, ast.Attribute):
            if isinstance(n.value, ast.Name) and n.value.id in ("np", "numpy") and n.attr in ("pi", "e", "inf", "nan", "euler_gamma"):
                return getattr(np, n.attr)
            base = self._eval(n.value, src, new)
            if isinstance(base, tuple) and base[0] == "slice":
                grid, what = _grid_of(base), f"{base[1].name}[{base[2]}:{base[3]}]"
            elif isinstance(base, (Var, SExpr)):
                grid, what = base.grid, base.name
            else:
                raise ProcessingChainError(f"unsupported attribute in '{src}'")
            if n.attr == "unit":  # name.unit in a declaration: unit=vov.unit (reference tests/test_processing_chain.py:660)
                return getattr(base[1] if isinstance(base, tuple) else base, "unit", None)
            if n.attr not in ("period", "offset", "grid"):
                raise ProcessingChainError(f"unsupported attribute '.{n.attr}' in '{src}'")
            if grid is None:
                raise ProcessingChainError(f"'{what}' has no coordinate grid (wrap the input in WaveformInput, or declare grid=/period=)")
            if n.attr == "grid":
                return grid
            if n.attr == "period":
                return Quantity(grid.period)
            if grid.offset_var is None:
                return Quantity(grid.offset)
            return self.offset_ns(grid)
        if isinstance(n, ast.Subscript):
            base = self._eval(n.value, src, new)
            first = 0
            if _is_wf(base) and isinstance(base, tuple):  # a slice of a (named) slice: the same view of the waveform underneath
                first, length, base = base[2], base[3] - base[2], base[1]
            elif isinstance(base, Var) and base.kind == "wf":
                length = base.length
            else:
                raise ProcessingChainError(f"Cannot apply subscript to {self._nm(base)} in '{src}'")
            if isinstance(n.slice, ast.Tuple):
                raise ProcessingChainError("Tuple still isn't implemented...")
            if not isinstance(n.slice, ast.Slice):  # wf[i]: one sample, a per-event value (reference :976-1005)
                idx = self._eval(n.slice, src, new)
                vlen = base.vector_len if isinstance(base, Var) else None
                if not _is_scalar(idx) and vlen is not None and not isinstance(idx, (Quantity, tuple, Grid)) and float(idx) < 0:
                    idx = self._scalar_binop(ast.Sub(), vlen, -int(round(float(idx))), src)  # -k counts from the row's own end: "len-k" (:972-973)
                if _is_scalar(idx):
                    # a per-event index: the reference adds get_default(w, i, NaN) (processors/get.py:50-92) -- the sample, or NaN when the
                    # index lies outside the array or the sample itself is NaN; a negative index counts from the end
                    self._anon += 1
                    out = Var(f"{base.name}[{idx.name}]#{self._anon}", None, unit=base.unit, is_coord=False)
                    whole = base if first == 0 and length == base.length else ("slice", base, first, first + length)
                    self._step("get", [whole, idx, out], "wsS")
                    return out
                i = self._const_int(n.slice, src, new, 0, base)
                i = i + length if i < 0 else i
                if not 0 <= i < length:
                    raise ProcessingChainError(f"index {i} is out of bounds for '{base.name}' with {length} samples in '{src}'")
                self._anon += 1
                out = Var(f"{base.name}[{first + i}]#{self._anon}", None, unit=base.unit, is_coord=False)
                view = ("slice", base, first + i, first + i + 1) if base.is_input else base  # (of an input only that sample is read)
                self._step("sample", [view, 0 if base.is_input else first + i, out], "wiS")
                return out
            step = self._const_int(n.slice.step, src, new, 1, None)
            if step == 0:
                raise ProcessingChainError(f"slice step cannot be zero in '{src}'")
            if step < 0:  # wf[::-1], wf[100:10:-2]: NumPy's slice of the buffer (reference :1009-1048), a copy with a negative stride here
                lower = None if n.slice.lower is None else self._const_int(n.slice.lower, src, new, 0, base)
                upper = None if n.slice.upper is None else self._const_int(n.slice.upper, src, new, 0, base)
                picks = range(*slice(lower, upper, step).indices(length))
                if len(picks) < 1:
                    raise ProcessingChainError(f"empty slice in '{src}'")
                self._anon += 1
                g = _grid_of(base if isinstance(base, Var) else ("slice", base, first, first + length))
                if g is not None:  # the period times the step; the offset moves only for an explicit positive start (reference :1031-1048)
                    g = Grid(g.period * step, g.offset + (lower * g.period if lower is not None and lower > 0 else 0.0), g.offset_var)
                out = Var(f"{base.name}[{'' if lower is None else first + lower}:{'' if upper is None else first + upper}:{step}]#{self._anon}",
                          "wf", len(picks), np.float32, grid=g, unit=base.unit, is_coord=False)
                if base.is_input:  # only the span the slice covers is read from the input
                    self._step("slice", [("slice", base, first + picks[-1], first + picks[0] + 1), picks[0] - picks[-1], step, out], "wiiW")
                else:
                    self._step("slice", [base, first + picks[0], step, out], "wiiW")
                return out
            lo = self._const_int(n.slice.lower, src, new, 0, base)
            hi = self._const_int(n.slice.upper, src, new, length, base)
            lo = max(lo + length, 0) if lo < 0 else min(lo, length)
            hi = max(hi + length, 0) if hi < 0 else min(hi, length)
            view = ("slice", base, first + lo, first + max(hi, lo))
            if step == 1:
                return view
            count = len(range(lo, hi, step))
            if count < 1:
                raise ProcessingChainError(f"empty slice in '{src}'")
            self._anon += 1
            g = _grid_of(view)
            out = Var(f"{base.name}[{first + lo}:{first + hi}:{step}]#{self._anon}", "wf", count, np.float32,
                      grid=Grid(g.period * step, g.offset, g.offset_var) if g is not None else None, unit=base.unit, is_coord=False)
            if base.is_input:  # only the span the slice covers is read from the input
                self._step("slice", [("slice", base, first + lo, first + lo + (count - 1) * step + 1), 0, step, out], "wiiW")
            else:
                self._step("slice", [base, first + lo, step, out], "wiiW")
            return out
        if isinstance(n, ast.Call) and isinstance(n.func, ast.Name):
            f = n.func.id
            if f == "loadlh5":  # loadlh5(file, path): an object of an LH5 file as a constant (reference :1444-1467)
                if len(n.args) != 2 or not all(isinstance(x, ast.Constant) and isinstance(x.value, str) for x in n.args):
                    raise ProcessingChainError(f"loadlh5() takes a file and a path in it, both strings, in '{src}'")
                from .lgdo_io import load_constant

                return load_constant(n.args[0].value, n.args[1].value)
            if f in _CALLS:
                a = [self._eval(x, src, new) for x in n.args]
                if f == "len":
                    v = a[0]
                    if isinstance(v, Var) and v.vector_len is not None:  # a variable-length array: its per-event length (reference :1182-1183)
                        return v.vector_len
                    if isinstance(v, tuple) and v[0] == "slice":
                        return v[3] - v[2]
                    if not isinstance(v, Var) or v.length is None:
                        raise ProcessingChainError(f"len() of something without a length in '{src}'")
                    return v.length
                if f in _ROUND_MODES:
                    return self._round(f, a, src)
                if f == "where":  # where(condition, a, b, dtype=...)  (reference :1345-1430)
                    if len(a) != 3:
                        raise ProcessingChainError(f"where() takes a condition and two values in '{src}'")
                    return self._where(a[0], a[1], a[2], src)
                if f in ("isnan", "isfinite"):
                    x = a[0]
                    fn = _lib.FN_ISNAN if f == "isnan" else _lib.FN_ISFINITE
                    if _is_wf(x):
                        return self._elementwise(fn, [x], f"{f}({self._nm(x)})", src, self._unit_of(x), np.bool_)
                    if _is_scalar(x):
                        return self._scalar_func(fn, [x], f"{f}({x.name})", x.unit, x.is_coord, x.grid, np.bool_)
                    return bool(getattr(np, f)(float(x)))
                if f == "astype":  # a copy in another type (reference :1268-1300); the device loops are float32 / float64
                    x, d = a[0], np.dtype(a[1][1] if isinstance(a[1], tuple) else a[1])
                    if not (_is_wf(x) or _is_scalar(x)):
                        raise ProcessingChainError(f"cannot call astype() on {x!r}")
                    if d.kind in "iu" and (d.itemsize <= 4 or (_is_scalar(x) and _is_int_dtype(x))):
                        # numpy.copyto(casting="unsafe"): truncation, then the wrap to the type (a per-event integer to a 64-bit type: in the
                        # integer program, _int_island)
                        fn, nm = _lib.fn_int(_lib.FN_ICAST, d), f"{self._nm(x)}.astype(`{d.char}`)"
                        if _is_wf(x):
                            return self._elementwise(fn, [x], nm, src, self._unit_of(x), d)
                        return self._scalar_func(fn, [x], nm, x.unit, x.is_coord, x.grid, d)
                    if d == np.dtype(np.bool_):  # ... to a truth value: x != 0
                        nm = f"{self._nm(x)}.astype(`?`)"
                        if _is_wf(x):
                            return self._elementwise(_lib.FN_NE, [x, 0.0], nm, src, self._unit_of(x), np.bool_)
                        return self._scalar_func(_lib.FN_NE, [x, 0.0], nm, x.unit, x.is_coord, x.grid, np.bool_)
                    if d.kind != "f" or d.itemsize < 4:
                        raise NotImplementedError(f"astype to {d} is not available on the device path (float32 / float64 loops; 64-bit integers from "
                                                  f"per-event integers only): '{src}'")
                    if _is_wf(x):
                        out = self._elementwise(_lib.FN_COPY, [x], f"{self._nm(x)}.astype(`{d.char}`)", src, self._unit_of(x))
                    elif _is_scalar(x):
                        out = self._scalar_func(_lib.FN_COPY, [x], f"{x.name}.astype(`{d.char}`)", x.unit, x.is_coord, x.grid, None)
                    else:
                        raise ProcessingChainError(f"cannot call astype() on {x!r}")
                    out.want_dtype = d
                    return out
                return {"float": float, "int": int}[f](a[0])
            # declaration:  name(length, 'f', grid=..., unit=..., period=..., offset=...)  (reference :1101-1122, 334-374)
            if f in new or f not in self.vars or isinstance(self.vars.get(f), Var):
                v = self.vars.get(f)
                if v is None:
                    v = Var(f, None)
                    self.vars[f] = v
                if n.args and v.length is None:
                    shape = self._eval(n.args[0], src, new)
                    if isinstance(shape, Quantity):
                        raise ProcessingChainError(f"shape in '{src}' has time units; divide by a period")
                    v.kind, v.length = "wf", int(round(float(shape)))
                    v.dtype = np.dtype(np.float32)
                    if len(n.args) > 1:
                        d = self._eval(n.args[1], src, new)
                        v.dtype = np.dtype(d[1] if isinstance(d, tuple) else d)
                elif not n.args and not n.keywords:
                    raise ProcessingChainError(f"declaration '{src}' needs a shape")
                kw = {k.arg: self._eval(k.value, src, new) for k in n.keywords}
                for k in kw:
                    if k not in ("unit", "period", "offset", "grid", "dtype", "is_coord", "shape", "vector_len"):
                        raise ProcessingChainError(f"unknown keyword '{k}' in declaration '{src}'")
                if "shape" in kw:
                    shape = int(round(float(kw["shape"])))
                    if v.is_input and v.kind == "wf":
                        # the maximum length of a variable-length input (reference :2213-2232): the rows arrive padded (lgdo_io.RaggedColumn);
                        # the variable takes the first `shape` samples of them, and no row may hold more
                        lens = self.tb_in.get(f"len({v.name})")
                        if shape > v.length:
                            raise NotImplementedError(f"'{src}': the input arrives padded to {v.length} samples; pad it to {shape} (RaggedColumn.from_vov(max_len=...))")
                        if lens is not None and len(lens) and int(np.max(np.asarray(lens))) > shape:
                            raise DSPFatal("VectorOfVectors entry has length larger than array variable length")
                        v.length = shape
                    elif v.length is None:
                        v.kind, v.length = "wf", shape
                        v.dtype = v.dtype if v.dtype is not None else np.dtype(np.float32)
                if "vector_len" in kw:
                    vl = kw["vector_len"]
                    if not _is_scalar(vl):
                        raise ProcessingChainError(f"vector_len in '{src}' must be a per-event variable")
                    v.vector_len = vl
                if "dtype" in kw:
                    d = kw["dtype"]
                    v.dtype = np.dtype(d[1] if isinstance(d, tuple) else d)
                if "unit" in kw and v.unit is None:
                    u = kw["unit"]
                    v.unit = u[1] if isinstance(u, tuple) else u
                if "is_coord" in kw and v.is_coord is None:
                    v.is_coord = bool(kw["is_coord"])
                if v.grid is None:
                    if isinstance(kw.get("grid"), Grid):
                        v.grid = kw["grid"]
                    elif "period" in kw:
                        per, off = kw["period"], kw.get("offset", 0.0)
                        if not isinstance(per, Quantity):
                            raise ProcessingChainError(f"period= in '{src}' must be a time")
                        if _is_scalar(off):
                            if off.is_coord is not True or off.grid is None:
                                raise NotImplementedError(f"offset= in '{src}': a per-event offset must be a time coordinate")
                            ns = off if off.grid == Grid(1.0) else self.converted(off, Grid(1.0))
                            v.grid = Grid(float(per), 0.0, ns)
                        else:  # a number counts periods (reference :101-102), a time is a time
                            v.grid = Grid(float(per), float(off) if isinstance(off, Quantity) else float(off) * float(per))
                return v
        raise ProcessingChainError(f"could not parse argument '{src}'")

    def _const_int(self, node, src, new, default, base=None):
        if node is None:
            return default
        v = self._eval(node, src, new)
        if isinstance(v, Quantity):  # a time as slice bound: in samples of the sliced waveform (reference :962-963)
            if base is None or base.period is None:
                raise ProcessingChainError(f"slice bound with time units in '{src}' on a waveform without a sampling period")
            v = float(v) / base.period
        if isinstance(v, (Var, SExpr, tuple, Grid)):
            # the reference refuses a variable as a slice bound with exactly this (:1016-1022); a window that starts at a per-event time is
            # the `windower` processor's job there (icpc-dsp-config.json: wf_le)
            raise ProcessingChainError(f"Slice values must be constants: '{src}'")
        return int(round(float(v)))

    def _round(self, f, a, src):
        """round / floor / ceil / trunc (value, to_nearest = 1) -- reference :1193-1266 with round_to_nearest.py"""
        fun = {"round": lambda x: float(np.rint(x)), "floor": math.floor, "ceil": math.ceil, "trunc": math.trunc}[f]
        val, to = a[0], (a[1] if len(a) > 1 else 1)
        if not isinstance(val, (Var, SExpr, tuple)):
            if isinstance(to, Grid):
                raise ProcessingChainError(f"cannot round a constant to a grid in '{src}'; use its period")
            r = float(to) * fun(float(val) / float(to))
            if isinstance(val, Quantity) != isinstance(to, Quantity):
                raise ProcessingChainError(f"'{src}': value and to_nearest must both be times or both be numbers")
            if isinstance(val, Quantity):
                return Quantity(r)
            return int(r) if float(r).is_integer() and not isinstance(to, float) else r
        if _is_wf(val):
            # a waveform: the reference's round_to_nearest / floor_to_nearest / ... ufunc sample by sample (processors/round_to_nearest.py):
            # to_nearest * f(val / to_nearest), each operation in the loop's type; a NaN sample stays NaN
            if isinstance(to, (Grid, Quantity)):
                raise ProcessingChainError(f"could not find valid conversion for {to!r} in '{src}': a waveform's samples are not times")
            fn = {"round": _lib.FN_RINT, "floor": _lib.FN_FLOOR, "ceil": _lib.FN_CEIL, "trunc": _lib.FN_TRUNC}[f]
            unit, nm = self._unit_of(val), self._nm(val)
            q = val if float(to) == 1.0 else self._elementwise(_lib.FN_DIV, [val, float(to)], f"({nm}/{to})", src, unit)
            r = self._elementwise(fn, [q], f"{f}({nm}, {to})", src, unit)
            return r if float(to) == 1.0 else self._elementwise(_lib.FN_MUL, [r, float(to)], f"{f}({nm}, {to})", src, unit)
        if not _is_scalar(val):
            raise ProcessingChainError(f"cannot round {val!r} in '{src}'")
        mode = _ROUND_MODES[f]
        if val.is_coord is True:
            if val.grid is None:
                raise ProcessingChainError(f"'{val.name}' in '{src}' has no coordinate grid yet")
            if isinstance(to, Grid):
                grid = to
            elif isinstance(to, Quantity):
                grid = Grid(float(to), val.grid.offset, val.grid.offset_var)
            else:
                grid = Grid(val.grid.period * float(to), val.grid.offset, val.grid.offset_var)
            return self.converted(val, grid, mode)
        if isinstance(to, (Grid, Quantity)):  # (the reference hands the time to the rounding ufunc, whose manager finds no grid to count it in, :1752-1756)
            raise ProcessingChainError(f"could not find valid conversion for {to!r} in '{src}'; '{val.name}' is not a time coordinate")
        q = val if float(to) == 1.0 else SExpr("div", (val, float(to)), f"({val.name}/{to})", val.unit, False, None)
        r = SExpr("convert", (q, 0.0, 0.0, 1.0), f"{f}({val.name}, {to})", val.unit, False, None, mode)
        return r if float(to) == 1.0 else SExpr("affine", (r, float(to), -0.0), f"{f}({val.name}, {to})", val.unit, False, None)

    def _binop(self, op, a, b, src=""):
        sa, sb = _is_scalar(a), _is_scalar(b)
        if (isinstance(a, np.ndarray) or isinstance(b, np.ndarray)) and (_is_wf(a) or _is_wf(b) or sa or sb):
            raise NotImplementedError(f"a constant array beside a variable in '{src}': declare it as the kernel of a processor, or spell the "
                                      "operation per sample")
        if _is_wf(a) or _is_wf(b):
            return self._wf_binop(op, a, b, src)
        if sa or sb:
            return self._scalar_binop(op, a, b, src)
        if isinstance(a, (Var, tuple, Grid)) or isinstance(b, (Var, tuple, Grid)):
            raise ProcessingChainError(f"operands {a!r} and {b!r} of '{src}' are not numbers or variables")
        if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):  # constant arrays: the NumPy operation itself, once, on the host
            if isinstance(a, Quantity) or isinstance(b, Quantity):
                raise ProcessingChainError(f"a constant array and a time in '{src}'")
            fn = {ast.Add: np.add, ast.Sub: np.subtract, ast.Mult: np.multiply, ast.Div: np.divide, ast.FloorDiv: np.floor_divide}.get(type(op))
            if fn is None:
                raise ProcessingChainError("unsupported operator in argument expression")
            return fn(a, b)
        qa, qb = isinstance(a, Quantity), isinstance(b, Quantity)
        fa, fb = float(a), float(b)
        if isinstance(op, ast.Add):
            r, q = fa + fb, qa or qb
            if qa != qb:
                raise ProcessingChainError("adding a time to a plain number")
        elif isinstance(op, ast.Sub):
            r, q = fa - fb, qa or qb
            if qa != qb:
                raise ProcessingChainError("subtracting a time and a plain number")
        elif isinstance(op, ast.Mult):
            r, q = fa * fb, qa != qb
            if qa and qb:
                raise ProcessingChainError("time * time is not a time")
        elif isinstance(op, ast.Div):
            r, q = fa / fb, qa and not qb
            if qb and not qa:
                raise ProcessingChainError("number / time is not supported")
        elif isinstance(op, ast.FloorDiv):
            r, q = fa // fb, qa and not qb
        else:
            raise ProcessingChainError("unsupported operator in argument expression")
        if q:
            return Quantity(r, getattr(a if qa else b, "unit", "ns"))
        if all(isinstance(x, int) and not isinstance(x, bool) for x in (a, b)) and not isinstance(op, ast.Div):
            return int(r)
        return r

    # ---- the NumPy ufuncs the language adds as processors (reference :832-947, 1266-1430)
    @staticmethod
    def _nm(a):
        if isinstance(a, tuple) and a and a[0] == "slice":
            return f"{a[1].name}[{a[2]}:{a[3]}]"
        return a.name if isinstance(a, (Var, SExpr)) else str(a)

    @staticmethod
    def _unit_of(a):
        return (a[1] if isinstance(a, tuple) else a).unit

    def _step(self, fn, args, roles):
        _, args = _resolve(self, roles, args, same_dim_out=True)
        self.steps.append((fn, args, self.cur_key))

    def _elementwise(self, fn, opnds, name, src, unit=None, dtype=np.float32):
        """f(A, B, C) sample by sample with at least one waveform among the operands: a new waveform variable and the step that fills it"""
        ops3 = list(opnds) + [None] * (3 - len(opnds))
        n = None
        for a in ops3:
            if _is_wf(a):
                if _wf_len(a) is None:
                    raise ProcessingChainError(f"'{src}': waveform '{self._nm(a)}' has no length yet")
                if n is not None and _wf_len(a) != n:
                    raise ProcessingChainError(f"failed to broadcast array dimensions in '{src}': waveforms of {n} and {_wf_len(a)} samples")
                n = _wf_len(a)
            elif isinstance(a, (Grid, tuple)) or (isinstance(a, Var) and a.kind not in ("scalar",)):
                raise ProcessingChainError(f"'{src}': {a!r} is not a number, a per-event variable or a waveform")
        grid = next((g for g in (_grid_of(a) for a in ops3 if _is_wf(a)) if g is not None), None)
        self._anon += 1
        out = Var(f"{name}#{self._anon}", "wf", n, dtype, grid=grid, unit=unit, is_coord=False)
        if getattr(self, "_wide_bound", None) is not None:
            out.int_bits, self._wide_bound = self._wide_bound, None
        roles = "".join("w" if _is_wf(a) else ("c" if a is None else "s") for a in ops3)
        self._step("ew:" + roles, [int(fn), *ops3, out], "c" + roles + "W")
        return out

    def _scalar_func(self, fn, opnds, name, unit, is_coord, grid, dtype):
        """the same between per-event values: one scalar op when something first reads the result"""
        out = SExpr("func", (), name, unit, is_coord, grid)
        out.dtype = np.dtype(dtype) if dtype is not None else None
        _, res = _resolve(self, "s" * len(opnds) + "S", [*opnds, out], expression=True)
        out.args = (int(fn), *res[:-1])
        return out

    def _int_bits(self, x) -> int:
        """bits of magnitude an integer operand can hold: of a column / waveform its type's, of a constant its value's, of a result what its
        operands' bounds give (kept on the variable by _wide_wf_loop)"""
        if not (_is_wf(x) or _is_scalar(x)):
            return max(1, int(abs(float(x))).bit_length())
        v = x[1] if isinstance(x, tuple) else x
        known = getattr(v, "int_bits", None)
        if known is not None:
            return known
        dt = np.dtype(v.dtype)
        return 1 if dt.kind == "b" else dt.itemsize * 8 - (1 if dt.kind == "i" else 0) + (1 if dt.kind == "i" else 0)

    def _wide_wf_loop(self, dtype, code, opnds, src):
        """A 64-bit integer loop on WAVEFORMS (int32 beside uint32 samples: NumPy's 'll->l'): the waveform VM holds samples in the chain's
        float type, and a float64 holds every integer below 2^53.  The loop is taken when the operands' types bound the result below that --
        then nothing wraps either, so the float64 chain's exact integer arithmetic IS the int64 loop -- and refused by name otherwise.
        (Per-event 64-bit integers are exact in any case: they run in an integer program of their own, _int_island.)"""
        dtype = np.dtype(dtype)
        if dtype.itemsize < 8:
            return dtype
        ba, bb = (self._int_bits(x) for x in opnds)
        bound = {_lib.FN_IADD: max(ba, bb) + 1, _lib.FN_ISUB: max(ba, bb) + 1, _lib.FN_IMUL: ba + bb, _lib.FN_IFLOORDIV: ba}[code]
        if bound > 53:
            raise NotImplementedError(f"'{src}' is a 64-bit integer loop on waveforms whose result can exceed 2^53 ({bound} bits): the waveform "
                                      "kernels hold samples in float64 at most; cast an operand to a float (astype)")
        self._wide_bound = bound  # (picked up by _elementwise for the variable it makes)
        return dtype

    def _wf_binop(self, op, a, b, src):
        fn, sym = {ast.Add: (_lib.FN_ADD, "+"), ast.Sub: (_lib.FN_SUB, "-"), ast.Mult: (_lib.FN_MUL, "*"), ast.Div: (_lib.FN_DIV, "/"),
                   ast.FloorDiv: (_lib.FN_FLOORDIV, "//")}.get(type(op), (None, None))
        variables = [x for x in (a, b) if _is_wf(x) or _is_scalar(x)]
        int_loop = fn not in (None, _lib.FN_DIV) and all(_is_int_dtype(x) for x in variables)
        if fn is None:  # (%, **, @ ...: not in the reference's operator table either, :46-59)
            raise ProcessingChainError(f"Could not parse expression:\n  {src}")
        dtype = np.float32
        if int_loop and _all_bool(variables) and fn != _lib.FN_FLOORDIV:
            # truth values alone: numpy.add and numpy.multiply have '??->?' loops -- logical or, logical and --, numpy.subtract refuses
            if fn == _lib.FN_SUB:
                raise ProcessingChainError(f"'{src}': {_BOOL_MINUS}")
            dtype, fn = np.dtype(np.bool_), (_lib.FN_LOR if fn == _lib.FN_ADD else _lib.FN_LAND)
            a, b = (x if (_is_wf(x) or _is_scalar(x)) else _int_loop_const(x, dtype, src) for x in (a, b))
        elif int_loop:
            # every variable is an integer: the reference's first matching ufunc loop is an integer one (:1565-1572), with its wrap-around
            dtype = _int_loop_of(variables, src)
            code = {_lib.FN_ADD: _lib.FN_IADD, _lib.FN_SUB: _lib.FN_ISUB, _lib.FN_MUL: _lib.FN_IMUL, _lib.FN_FLOORDIV: _lib.FN_IFLOORDIV}[fn]
            per = next((g.period for g in (_grid_of(x) for x in (a, b) if _is_wf(x)) if g is not None), self.default_period)
            a, b = (x if (_is_wf(x) or _is_scalar(x)) else _int_loop_const(x, dtype, src, per) for x in (a, b))
            dtype = self._wide_wf_loop(dtype, code, [a, b], src)
            fn = _lib.fn_int(code, dtype)
        va, vb = _is_wf(a) or _is_scalar(a), _is_wf(b) or _is_scalar(b)
        ua, ub = (self._unit_of(a) if va else None), (self._unit_of(b) if vb else None)
        if va and vb:  # reference :848-862
            ta, tb = _time_unit_ns(ua), _time_unit_ns(ub)
            if ta is not None and tb is not None:
                unit = ua if sym in "+-" else None
            elif ua is not None and ub is not None:
                unit = f"{ua}{sym}{ub}" if sym in ("*", "/", "//") else ua
            else:
                unit = ua if ua is not None else ub
        else:
            unit = ua if va else ub
        return self._elementwise(fn, [a, b], f"({self._nm(a)}{sym}{self._nm(b)})", src, unit, dtype)

    def _where(self, cond, a, b, src):
        """where(condition, a, b) / ``a if condition else b`` (reference :1345-1430)"""
        if not (isinstance(cond, (Var, SExpr)) and getattr(cond, "dtype", None) == np.dtype(np.bool_)):
            raise ProcessingChainError(f"{self._nm(cond)} must be a boolean variable")
        is_var = lambda x: _is_wf(x) or _is_scalar(x)  # noqa: E731
        grid_of = lambda x: _grid_of(x) if _is_wf(x) else x.grid  # noqa: E731
        coord_of = lambda x: False if _is_wf(x) else x.is_coord  # noqa: E731
        for x in (a, b):
            if not is_var(x) and isinstance(x, (Var, tuple, Grid)):
                raise ProcessingChainError(f"cannot select {x!r} in '{src}'")
        name = f"where({self._nm(cond)}, {self._nm(a)}, {self._nm(b)})"
        if is_var(a) and is_var(b):
            ga, gb = grid_of(a), grid_of(b)
            if ga is not None and gb is not None and ga.period != gb.period:  # (a value without a grid goes with any)
                raise ProcessingChainError(f"Cannot select between {self._nm(a)} and {self._nm(b)} with different periods")
            if coord_of(a) is not None and coord_of(b) is not None and coord_of(a) != coord_of(b):  # (None: still open, goes with either)
                raise ProcessingChainError(f"Cannot select between {self._nm(a)} and {self._nm(b)} with different is_coord")
            if ga is not None and gb is not None and ga != gb:
                raise NotImplementedError(f"'{src}': the two values have different offsets; an offset chosen per event by the condition is "
                                          "not supported on the device path")
            grid, is_coord = (ga if ga is not None else gb), (coord_of(a) if coord_of(a) is not None else coord_of(b))
            ua, ub = self._unit_of(a), self._unit_of(b)
            same = ua == ub or (_time_unit_ns(ua) is not None and _time_unit_ns(ua) == _time_unit_ns(ub))
            if same or not ub:
                unit = ua
            elif not ua:
                unit = ub
            else:
                raise ProcessingChainError(f"{self._nm(a)} and {self._nm(b)} do not have compatible units")
        elif is_var(a) or is_var(b):
            var, const = (a, b) if is_var(a) else (b, a)
            grid, is_coord, unit = grid_of(var), coord_of(var), self._unit_of(var)
            if isinstance(const, Quantity):
                tu = _time_unit_ns(unit)
                if tu is None:
                    raise ProcessingChainError(f"{self._nm(a)} and {self._nm(b)} do not have compatible units")
                const = float(const) / (grid.period if (is_coord is True and grid is not None) else tu)
            a, b = (var, const) if is_var(a) else (const, var)
        else:
            grid, is_coord = None, False
            qa, qb = isinstance(a, Quantity), isinstance(b, Quantity)
            unit = a.unit if qa else (b.unit if qb else None)
            if unit is not None:
                a, b = (float(a) / _UNITS_NS[unit] if qa else a), (float(b) / _UNITS_NS[unit] if qb else b)
        both_bool = all(getattr(x, "dtype", None) == np.dtype(np.bool_) if is_var(x) else isinstance(x, bool) for x in (a, b))
        dtype = np.bool_ if both_bool else np.float32
        variables = [x for x in (a, b) if is_var(x)]
        wide = False
        if variables and not both_bool and all(_is_int_dtype(x) for x in variables) and not any(_is_wf(x) for x in variables):
            # integer columns select an integer signature of the reference's where (processors/where.py:11-20), the constant beside one is
            # converted to it (:1765-1768).  The value is the chosen operand's, whatever the type: only its label -- and 64-bit integers,
            # which no float register holds -- matter here
            int_dt = _int_loop_of(variables, src, _WHERE_LOOPS)
            if int_dt.itemsize == 8:
                wide, dtype = True, int_dt
                a, b = (x if is_var(x) else _int_loop_const(x, int_dt, src) for x in (a, b))
        if any(_is_wf(x) for x in (cond, a, b)):
            out = self._elementwise(_lib.FN_WHERE, [cond, a, b], name, src, unit, dtype)
            if grid is not None:
                out.grid = grid
            return out
        return self._scalar_func(_lib.FN_WHERE, [cond, a, b], name, unit, is_coord, grid, dtype if (both_bool or wide) else None)

    def _scalar_binop(self, op, a, b, src, declared=None):
        """A binary operator with a per-event variable on at least one side: the reference adds the NumPy ufunc as a processor
        (:832-891), so the operands go through the same unit handling as any processor's (`_resolve`)."""
        sym = {ast.Add: "+", ast.Sub: "-", ast.Mult: "*", ast.Div: "/"}.get(type(op))
        # every variable an integer column: the reference's first matching ufunc loop is an integer one (:1565-1572), with its wrap-around
        int_dt = None
        if type(op) in (ast.Add, ast.Sub, ast.Mult, ast.FloorDiv) and all(_is_int_dtype(x) for x in (a, b) if _is_scalar(x)):
            variables = [x for x in (a, b) if _is_scalar(x)]
            if _all_bool(variables) and not isinstance(op, ast.FloorDiv):
                # truth values alone: numpy.add / numpy.multiply run their '??->?' loops (logical or / and), numpy.subtract refuses
                if isinstance(op, ast.Sub):
                    raise ProcessingChainError(f"'{src}': {_BOOL_MINUS}")
                a, b = (x if _is_scalar(x) else _int_loop_const(x, np.dtype(np.bool_), src) for x in (a, b))
                return self._scalar_func(_lib.FN_LOR if isinstance(op, ast.Add) else _lib.FN_LAND, [a, b],
                                         f"({self._nm(a)}{'+' if isinstance(op, ast.Add) else '*'}{self._nm(b)})", None, False, None, np.bool_)
            int_dt = _int_loop_of(variables, src)
        if sym is None and not isinstance(op, ast.FloorDiv):  # (%, **, @ ...: not in the reference's operator table either, :46-59)
            raise ProcessingChainError(f"Could not parse expression:\n  {src}")
        for x in (a, b):
            if isinstance(x, (tuple, Grid)) or (isinstance(x, Var) and x.kind != "scalar"):
                raise ProcessingChainError(f"operands {a!r} and {b!r} of '{src}' are not numbers or per-event variables")
        if isinstance(op, ast.FloorDiv):  # numpy.floor_divide as a processor: len(v)//2 and the like (reference :832-847)
            v = a if _is_scalar(a) else b
            if int_dt is not None:
                a, b = (x if _is_scalar(x) else _int_loop_const(x, int_dt, src, self.default_period) for x in (a, b))
                return self._scalar_func(_lib.fn_int(_lib.FN_IFLOORDIV, int_dt), [a, b], f"({self._nm(a)}//{self._nm(b)})", v.unit, False, None, int_dt)
            _, (a, b) = _resolve(self, "ss", [a, b], expression=True)  # (a time beside the variable counts periods of its grid)
            return self._scalar_func(_lib.FN_FLOORDIV, [a, b], f"({self._nm(a)}//{self._nm(b)})", v.unit, False, None, None)
        sa, sb = _is_scalar(a), _is_scalar(b)
        name = f"({a.name if sa else a}{sym}{b.name if sb else b})"
        if sa and sb:  # reference :848-872
            ta, tb = _time_unit_ns(a.unit), _time_unit_ns(b.unit)
            if ta is not None and tb is not None:
                unit = a.unit if sym in "+-" else None  # (time * time and time / time: not a time any more)
            elif a.unit is not None and b.unit is not None:
                unit = f"{a.unit}{sym}{b.unit}" if sym in "*/" else a.unit
            else:
                unit = a.unit if a.unit is not None else b.unit
            both = a.is_coord is True and b.is_coord is True
            out = SExpr(None, (), name, unit, False if both else None, None)
        else:
            v = a if sa else b
            out = SExpr(None, (), name, v.unit, v.is_coord, None)
        if declared is not None:  # numpy.add(a, b, out) written as a processor: `out` is a declared variable with its own unit
            out = SExpr(None, (), declared.name, declared.unit, declared.is_coord, declared.grid)
        a0, b0 = a, b
        _, (a, b, _o) = _resolve(self, "ssS", [a, b, out], expression=True)
        if int_dt is not None and all(x is x0 for x, x0 in ((a, a0), (b, b0)) if _is_scalar(x0)):  # (a converted coordinate is a float)
            a, b = (x if _is_scalar(x) else _int_loop_const(x, int_dt, src) for x in (a, b))
            code = {"+": _lib.FN_IADD, "-": _lib.FN_ISUB, "*": _lib.FN_IMUL}[sym]
            out.op, out.args, out.dtype = "func", (_lib.fn_int(code, int_dt), a, b), int_dt
            return out
        if sym == "+":
            out.op, out.args = "affine", ((a, 1.0, b) if sa else (b, 1.0, a))
        elif sym == "-":
            out.op, out.args = "affine", ((a, 1.0, -b) if not sb else (b, -1.0, a))
        elif sym == "*":
            out.op, out.args = "affine", ((a, b, -0.0) if sa else (b, a, -0.0))
        else:
            e = np.frexp(abs(float(b)))[0] if not sb and float(b) != 0 else 0
            if e == 0.5:  # a power of two: multiplying by the reciprocal is the same operation bit for bit
                out.op, out.args = "affine", (a, 1.0 / float(b), -0.0)
            else:
                out.op, out.args = "div", (a, b)
        return out


def _grid_of(a):
    """coordinate grid of a waveform operand"""
    if isinstance(a, tuple) and a[0] == "slice":
        g = a[1].grid
        return g.shifted(a[2]) if g is not None else None
    if isinstance(a, Var):
        return a.grid
    return None


def _resolve(b: _Builder, roles, args, same_dim_out=False, expression=False):
    """What ProcessorManager.__init__ does with the unit information of its parameters (reference :1556-1732, 1747-1770):

    * the processor's coordinate grid is the first waveform parameter's that has one, else the first time coordinate's;
    * a per-event parameter whose ``is_coord`` is still open becomes a coordinate on that grid if its unit is a time, a plain number
      otherwise; coordinates on another grid are converted;
    * constants with time units are divided by the grid's period;
    * an output waveform of the same dimension as the input takes over its grid.

    Returns (grid, converted arguments)."""
    G = None
    for a, r in zip(args, roles):
        if r in "wW":
            g = _grid_of(a)
            if G is None and g is not None:
                G = g
    if G is None:
        for a in args:
            if _is_scalar(a) and a.is_coord is True and a.grid is not None:
                G = a.grid
                break
    out = []
    for a, r in zip(args, roles):
        if _is_scalar(a):
            if a.is_coord is True:
                if a.grid is None and G is not None:
                    a.grid = G
            elif a.is_coord is None and not (isinstance(a, Var) and a.kind is None):
                if _time_unit_ns(a.unit) is not None and G is not None:
                    a.is_coord = True
                    if a.grid is None:
                        a.grid = G
                else:
                    a.is_coord = False
            if r == "s" and a.is_coord is True and G is not None and a.grid is not None and a.grid != G:
                a = b.converted(a, G)
        elif isinstance(a, Var) and a.kind is None and r == "S":  # a new per-event output
            a.kind = "scalar"
            if a.is_coord is None:
                a.is_coord = _time_unit_ns(a.unit) is not None and G is not None
            if a.is_coord and a.grid is None:
                a.grid = G
        elif isinstance(a, Quantity) and r in "si":
            if G is not None:
                a = float(a) / G.period
            elif expression:
                # no coordinate in the expression to take a grid from (the reference refuses: "could not find valid conversion",
                # :1752-1756); a per-event input column counts samples of the input waveform here
                if b.default_period is None:
                    raise ProcessingChainError(f"could not find valid conversion for {a!r}; CoordinateGrid is None")
                a = float(a) / b.default_period
        elif r == "W" and isinstance(a, Var) and a.grid is None and same_dim_out:
            a.grid = next((_grid_of(x) for x, rx in zip(args, roles) if rx == "w"), None)
        out.append(a)
    return G, out


