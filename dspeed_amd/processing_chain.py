"""JSON/YAML recipe -> fused device chain: the host-side mirror of the reference's chain builder and runtime for
the hot path (reference src/dspeed/processing_chain.py: ``build_processing_chain`` :2363-2872,
``ProcessingChain.execute`` :665-673, ``_execute_procs`` :1144-1163).

What is kept from the reference (so existing LEGEND recipes for the energy chain run unmodified):

* the recipe schema: ``{"outputs": [...], "processors": {"a, b": {"function", "module", "args", "kwargs",
  "defaults", "unit", "prereqs"}}}``, the one-string form ``"module.func(arg, ...)"``, ``db.x.y`` lookups with
  ``defaults`` (:2555-2583), multi-output keys split on ``,``/space (:2480-2483), dependency resolution by
  depth-first search from the requested outputs with cycle detection (:2601-2651), constant folding of processors
  whose inputs are all constants -- how cusp/zac kernels are built once (:2775-2823);
* argument syntax: literals, ``'c'`` characters, ``N*us`` quantities converted to samples with the waveform period
  (:1747-1770), output declarations ``name(length, 'f')``, constant slices ``wf[a:b]``, ``len(wf)``, ``round(x)``,
  ``wf.period``, arithmetic on constants, ``var + constant`` on per-event scalars;
* the literal module string ``dspeed.processors`` (and ``numpy`` for ``amax``) resolves to this package's registry.

What is different by design: instead of calling one gufunc per processor per 16-row block, the resolved processor
list is translated into ONE device program (``dsp_chain_create``) that keeps every intermediate waveform in LDS, and
``execute`` launches it over the whole buffer.  Anything outside the supported subset raises
``ProcessingChainError``/``NotImplementedError`` -- there is no CPU fallback.
"""
from __future__ import annotations

import ast
import json
import re
import time
from collections.abc import MutableMapping
from copy import deepcopy

import numpy as np

from . import _lib
from .chain import Chain, Program, Scalar
from .device import DeviceArray, Event, HostPin, Stream
from .errors import DSPFatal, ProcessingChainError

_UNITS_NS = {"ns": 1.0, "us": 1e3, "ms": 1e6, "s": 1e9}


class Quantity(float):
    """A time in nanoseconds (the only dimension hot-path recipes use)."""

    def __repr__(self):
        return f"{float(self):g}*ns"


class WaveformInput:
    """Input column with sampling information, the role of ``lgdo.WaveformTable`` (values, dt) in the reference
    (processing_chain.py:2263-2360).  ``dt`` in nanoseconds."""

    def __init__(self, values, dt: float = 16.0, t0: float = 0.0):
        self.values = values
        self.dt = float(dt)
        self.t0 = float(t0)

    def __len__(self):
        return len(self.values)


class Var:
    """A chain variable (the subset of ProcChainVar, processing_chain.py:147-377, that the device path needs)."""

    def __init__(self, name, kind, length=None, dtype=np.float32, period=None, const=None, source=None, offset=0):
        self.name = name
        self.kind = kind          # 'wf' | 'scalar' | 'const' | 'char' | 'taps'
        self.length = length      # samples (wf/taps)
        self.dtype = np.dtype(dtype) if dtype is not None else None
        self.period = period      # ns per sample
        self.const = const        # python value for constants, ndarray for taps
        self.source = source      # input column name for chain inputs
        self.offset = offset      # first sample for sliced inputs
        self.is_input = source is not None
        self.slot = None
        self.sreg = None
        self.io = None

    def __repr__(self):
        return f"<Var {self.name} {self.kind} len={self.length}>"


# signatures of the supported processors: argument roles, in recipe order
#   w = waveform in, W = waveform out, s = float scalar in (const or per-event), i = int const, c = char const,
#   S = scalar out, t = taps in
_SIGS = {
    "bl_subtract": "wsW", "pole_zero": "wsW", "double_pole_zero": "wsssW", "trap_filter": "wiiW", "trap_norm": "wiiW",
    "asym_trap_filter": "wiiiW", "fixed_time_pickoff": "wscS", "time_point_thresh": "wsssS", "min_max": "wSSSS",
    "discrete_wavelet_transform": "wiccW", "convolve_wf": "wtcW", "fft_convolve_wf": "wtcW", "amax": "wiS",
    "mean_below_threshold": "wsS", "windower": "wsW", "avg_current": "wsW", "trap_pickoff": "wiisS",
    "upsampler": "wsW", "moving_window_multi": "wsiiW", "add": "ssS", "linear_slope_fit": "wSSSS",
}
_GENERATORS = ("cusp_filter", "zac_filter", "t0_filter", "moving_slope")
_MODULES = ("dspeed.processors", "dspeed_amd.processors", "numpy", "np")


class ProcessingChain:
    """Runs a translated recipe over a buffer of rows.  ``execute(start, stop)`` has the meaning of the reference's
    (processing_chain.py:665-673); ``__call__(tb_in, tb_out)`` relinks I/O like :675-716."""

    def __init__(self, program: Program, inputs: dict, outputs: dict, consts: dict, buffer_len: int, proc_strings: list[str],
                 loop_dtype=np.float32):
        self._program = program
        self.loop_dtype = np.dtype(loop_dtype)  # float32 or float64 gufunc loop of the whole chain
        self._in_vars = inputs      # binding name -> Var (source column)
        self._out_vars = outputs    # binding name -> (Var, length or None)
        self._consts = consts       # binding name -> ndarray (taps)
        self._buffer_len = buffer_len
        self._chain = None
        self._stream = None
        self._dev = {}
        self._tb_in = None
        self._tb_out = None
        self._pins = {}           # (address, bytes) -> HostPin of a linked host column (None: registration refused)
        self._piece_key, self._piece_bufs, self._piece_events = None, [], []  # device buffers host columns are streamed through
        self._copy_stream = None  # H2D of the next piece runs here while the compute stream works on the current one
        self._timing = {"h2d": 0.0, "kernel": 0.0, "d2h": 0.0}
        self.proc_strings = proc_strings

    # -- introspection
    @property
    def program(self) -> Program:
        return self._program

    def get_timing(self) -> dict:
        return dict(self._timing)

    def __str__(self):
        return "Input variables: " + str(list(self._in_vars)) + "\nProcessors:\n  " + "\n  ".join(self.proc_strings)

    # -- I/O
    def link(self, tb_in, tb_out):
        if tb_in is not self._tb_in or tb_out is not self._tb_out:
            for pin in self._pins.values():
                if pin is not None:
                    pin.close()
            self._pins = {}
        self._tb_in, self._tb_out = tb_in, tb_out
        self._buffer_len = len(_column(tb_in, next(iter(self._in_vars.values())).source)) if self._in_vars else self._buffer_len

    def _ensure(self):
        if self._chain is None:
            self._chain = Chain(self._program, "processing_chain", self.loop_dtype)
            self._stream = Stream()
            for name, arr in self._consts.items():
                self._dev[name] = DeviceArray.from_numpy(arr)

    #: bytes of host-resident I/O per pipelined piece: tens of MB keep PCIe transfers efficient, two pieces are in flight
    pipeline_bytes = 64 << 20

    def _pinned(self, arr: np.ndarray) -> bool:
        """Page-lock a linked host column in place, once (the reference's build_dsp refills the same buffers for every file
        chunk).  Registration can be refused (read-only or already registered memory): copies then run unpinned, only slower."""
        if arr.nbytes < (1 << 20):  # small columns: the copy is latency, not bandwidth; and they share pages with their neighbours
            return False
        key = (arr.ctypes.data, arr.nbytes)
        if key not in self._pins:
            try:
                self._pins[key] = HostPin(arr)
            except Exception:
                self._pins[key] = None
        return self._pins[key] is not None

    def execute(self, start: int = 0, stop: int | None = None) -> None:
        if stop is None:
            stop = self._buffer_len
        n = stop - start
        if n <= 0:
            return
        self._ensure()
        lib = _lib.lib()
        # ---- sort the linked columns: device-resident ones are used in place, host ones are streamed through piece buffers
        dev_in, host_in, dev_out, host_out = {}, {}, {}, {}
        for name, var in self._in_vars.items():
            col = _column(self._tb_in, var.source)
            if isinstance(col, DeviceArray):
                dev_in[name] = col
            else:
                a = np.asarray(col)
                if not a.flags.c_contiguous:
                    a = np.ascontiguousarray(a)
                else:
                    self._pinned(a)
                host_in[name] = a
        for name, (var, length) in self._out_vars.items():
            col = self._tb_out[var.name]
            if isinstance(col, DeviceArray):
                dev_out[name] = col
            else:
                direct = isinstance(col, np.ndarray) and col.flags.c_contiguous and col.dtype == self.loop_dtype
                if direct:
                    self._pinned(col)
                host_out[name] = (col, length, direct)
        row_bytes = sum(a.nbytes // max(len(a), 1) for a in host_in.values())
        row_bytes += sum(self.loop_dtype.itemsize * (1 if length is None else length) for _, length, _ in host_out.values())
        piece = n if row_bytes == 0 else int(max(1, min(n, self.pipeline_bytes // row_bytes)))
        n_slots = 2 if piece < n else 1
        # piece buffers live as long as the chain (the reference pre-allocates its ProcChainVar buffers the same way,
        # processing_chain.py:259-269): build_dsp calls execute() once per file chunk with the same shapes
        key = (piece, n_slots, tuple((nm, a.shape[1:], a.dtype.str) for nm, a in host_in.items()),
               tuple((nm, length) for nm, (_, length, _) in host_out.items()))
        if self._piece_key != key:
            self._piece_bufs = []
            for _ in range(n_slots):
                sl = {name: DeviceArray((piece, *a.shape[1:]), a.dtype) for name, a in host_in.items()}
                sl.update({name: DeviceArray((piece,) if length is None else (piece, length), self.loop_dtype)
                           for name, (_, length, _) in host_out.items()})
                self._piece_bufs.append(sl)
            self._piece_events = [Event() for _ in range(n_slots)]
            self._piece_key = key
        slots, ev_in = self._piece_bufs, self._piece_events
        if self._copy_stream is None:
            self._copy_stream = Stream()
        s_in, s_c = self._copy_stream, self._stream

        def finish(a, b, temps):
            """piece [a, b): wait for its kernel and copies, report its DSPFatal with absolute rows, deliver converted outputs"""
            t = time.perf_counter()
            try:
                self._chain.check(s_c, row_offset=a)
            except DSPFatal as e:  # the reference annotates and re-raises (processing_chain.py:1154-1159)
                if e.wf_range is None:
                    e.wf_range = range(a, b)
                raise
            self._timing["kernel"] += time.perf_counter() - t
            t = time.perf_counter()
            for col, tmp in temps:
                col[a:b] = tmp.astype(col.dtype, copy=False)
            self._timing["d2h"] += time.perf_counter() - t

        # ---- pieces: H2D(k) on the copy stream overlaps kernel(k-1) and D2H(k-1) on the compute stream
        pending = None
        for k, a in enumerate(range(start, stop, piece)):
            b = min(stop, a + piece)
            m, sl = b - a, slots[k % n_slots]
            t = time.perf_counter()
            bufs = dict(self._dev)
            for name, arr in host_in.items():  # (slot k % 2 is free: piece k-2 was finished in iteration k-1)
                d = sl[name].view_rows(0, m)
                src = arr[a:b]
                _lib.check(lib.dsp_h2d_async(d.ptr, src.ctypes.data, src.nbytes, s_in.ptr), what="h2d_async")
                bufs[name] = d
            ev_in[k % n_slots].record(s_in)
            for name, col in dev_in.items():
                bufs[name] = col.view_rows(a, b)
            for name, col in dev_out.items():
                bufs[name] = col.view_rows(a, b)
            self._timing["h2d"] += time.perf_counter() - t
            if pending is not None:
                finish(*pending)
            s_c.wait_event(ev_in[k % n_slots])
            temps = []
            for name in host_out:
                bufs[name] = sl[name].view_rows(0, m)
            self._chain.execute(bufs, m, s_c)
            for name, (col, length, direct) in host_out.items():
                d = bufs[name]
                if direct:
                    dst = col[a:b]
                else:
                    dst = np.empty(d.shape, dtype=self.loop_dtype)
                    temps.append((col, dst))
                _lib.check(lib.dsp_d2h_async(dst.ctypes.data, d.ptr, d.nbytes, s_c.ptr), what="d2h_async")
            pending = (a, b, temps)
        finish(*pending)

    def __call__(self, tb_in, tb_out, begin: int = 0, end: int | None = None):
        self.link(tb_in, tb_out)
        self.execute(begin, end)
        return tb_out


def _column(tb, name):
    col = tb[name]
    return col.values if isinstance(col, WaveformInput) else col


# ----------------------------------------------------------------------------------------------------------------
# recipe parsing
# ----------------------------------------------------------------------------------------------------------------
_db_parser = re.compile(r"(?![^\w_.])db\.[\w_.]+")


def _load(processors):
    if isinstance(processors, str):
        with open(processors) as f:
            text = f.read()
        try:
            return json.loads(text)
        except json.JSONDecodeError:
            import yaml

            return yaml.safe_load(text)
    if processors is None:
        return {}
    if isinstance(processors, MutableMapping):
        return deepcopy(dict(processors))
    raise ValueError("processors must be a dict, json/yaml file, or None")


def _normalise(key, node):
    """Bring one recipe node to {'function', 'module', 'args'} (reference :2486-2553)."""
    if isinstance(node, str):
        node = {"function": node}
    if "function" not in node:
        raise ProcessingChainError(f"no function for parameter {key}")
    function = node["function"]
    f_parse = ast.parse(function, mode="eval").body
    seg = lambda n: function[n.col_offset: n.end_col_offset]  # noqa: E731
    if isinstance(f_parse, ast.Name):
        pass
    elif isinstance(f_parse, ast.Attribute):
        if "module" in node:
            raise ProcessingChainError(f"Module specified twice for parameter {key}")
        node["function"], node["module"] = f_parse.attr, seg(f_parse.value)
    elif isinstance(f_parse, ast.Call) and isinstance(f_parse.func, (ast.Name, ast.Attribute)):
        if "args" in node:
            raise ProcessingChainError(f"Cannot specify arguments if function is expr for parameter {key}")
        if isinstance(f_parse.func, ast.Attribute):
            if "module" in node:
                raise ProcessingChainError(f"Module specified twice for parameter {key}")
            node["function"], node["module"] = f_parse.func.attr, seg(f_parse.func.value)
            node["args"] = [seg(a) for a in f_parse.args + f_parse.keywords]
        elif f_parse.func.id in ("round", "len", "float", "int") and "module" not in node:
            node["module"], node["args"] = None, [function]
        else:
            node["function"] = f_parse.func.id
            node["args"] = [seg(a) for a in f_parse.args + f_parse.keywords]
    else:  # inline expression
        if "args" in node or "module" in node:
            raise ProcessingChainError(f"Cannot specify arguments/module if function is expr for parameter {key}")
        node["module"], node["args"] = None, [function]
    if "module" not in node:
        raise ProcessingChainError(f"Could not find module for parameter {key}")
    if "args" not in node:
        raise ProcessingChainError(f"Could not find args for parameter {key}")
    return node


def _substitute_db(node, db_dict):
    args = node["args"]
    for i, arg in enumerate(args):
        if not isinstance(arg, str):
            continue
        for db_var in _db_parser.findall(arg):
            try:
                db_node = db_dict
                for k in db_var[3:].split("."):
                    db_node = db_node[k]
            except (KeyError, TypeError):
                try:
                    db_node = node["defaults"][db_var]
                except (KeyError, TypeError):
                    raise ProcessingChainError(f"did not find {db_var} in database, and could not find default value.") from None
            arg = db_node if arg == db_var else arg.replace(db_var, str(db_node))
        args[i] = arg


def _names_in(arg: str) -> list[str]:
    """Variable names an argument string refers to (get_variable(..., get_names_only=True) in the reference)."""
    try:
        tree = ast.parse(arg, mode="eval")
    except SyntaxError:
        return []
    names = []
    for n in ast.walk(tree):
        if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id not in ("round", "len", "float", "int"):
            names.append(n.func.id)  # declaration name(shape, dtype)
    called = set(names)
    for n in ast.walk(tree):
        if isinstance(n, ast.Name) and n.id not in _UNITS_NS and n.id not in ("round", "len", "float", "int", "np") and n.id not in called:
            names.append(n.id)
    seen, out = set(), []
    for n in names:
        if n not in seen:
            seen.add(n)
            out.append(n)
    return out


class _Builder:
    def __init__(self, tb_in, db_dict):
        self.tb_in = tb_in if tb_in is not None else {}
        self.db = db_dict or {}
        self.vars: dict[str, Var] = {}
        self.steps = []  # (function name, [operands], recipe key)
        self.default_period = None
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
        period = col.dt if isinstance(col, WaveformInput) else None
        vals = col.values if isinstance(col, WaveformInput) else col
        shape, dtype = vals.shape, vals.dtype
        if len(shape) == 2:
            v = Var(name, "wf", shape[1], dtype, period, source=name)
        elif len(shape) == 1:
            v = Var(name, "scalar", None, dtype, source=name)
        else:
            raise ProcessingChainError(f"input '{name}' has unsupported shape {shape}")
        self.vars[name] = v
        return v

    # ---- expression evaluation
    def eval_arg(self, arg, want_new=None):
        """Turn a recipe argument into a Var / number / Quantity / char.  ``want_new``: names this processor creates."""
        if not isinstance(arg, str):
            return arg
        tree = ast.parse(arg.strip(), mode="eval").body
        return self._eval(tree, arg, want_new or ())

    def _eval(self, n, src, new):
        if isinstance(n, ast.Constant):
            if isinstance(n.value, str):
                return ("char", n.value)
            return n.value
        if isinstance(n, ast.Name):
            if n.id in _UNITS_NS:
                return Quantity(_UNITS_NS[n.id])
            if n.id in self.vars:
                v = self.vars[n.id]
                return v.const if v.kind == "const" else v
            if n.id in new:
                v = Var(n.id, None)
                self.vars[n.id] = v
                return v
            return self.input_var(n.id)
        if isinstance(n, ast.UnaryOp) and isinstance(n.op, (ast.USub, ast.UAdd)):
            v = self._eval(n.operand, src, new)
            return -v if isinstance(n.op, ast.USub) else v
        if isinstance(n, ast.BinOp):
            a, b = self._eval(n.left, src, new), self._eval(n.right, src, new)
            return self._binop(n.op, a, b)
        if isinstance(n, ast.Attribute):
            base = self._eval(n.value, src, new)
            if isinstance(base, Var) and n.attr == "period":
                if base.period is None:
                    raise ProcessingChainError(f"'{base.name}' has no sampling period (wrap the input in WaveformInput)")
                return Quantity(base.period)
            if isinstance(n.value, ast.Name) and n.value.id == "np" and n.attr in ("pi", "e", "inf", "nan"):
                return getattr(np, n.attr)
            raise ProcessingChainError(f"unsupported attribute in '{src}'")
        if isinstance(n, ast.Subscript):
            base = self._eval(n.value, src, new)
            if not (isinstance(base, Var) and base.kind == "wf" and isinstance(n.slice, ast.Slice)):
                raise NotImplementedError(f"only constant slices of waveforms are supported: '{src}'")
            lo = self._const_int(n.slice.lower, src, new, 0)
            hi = self._const_int(n.slice.upper, src, new, base.length)
            if n.slice.step is not None:
                raise NotImplementedError(f"strided slices are not supported: '{src}'")
            lo = lo + base.length if lo < 0 else lo
            hi = hi + base.length if hi < 0 else min(hi, base.length)
            return ("slice", base, lo, hi)
        if isinstance(n, ast.Call) and isinstance(n.func, ast.Name):
            f = n.func.id
            if f in ("round", "len", "float", "int"):
                a = [self._eval(x, src, new) for x in n.args]
                if f == "len":
                    v = a[0]
                    if isinstance(v, tuple) and v[0] == "slice":
                        return v[3] - v[2]
                    if not isinstance(v, Var) or v.length is None:
                        raise ProcessingChainError(f"len() of something without a length in '{src}'")
                    return v.length
                if f == "round":
                    if isinstance(a[0], Var):
                        raise NotImplementedError("round() of per-event variables is not supported on the device")
                    return type(a[0])(round(float(a[0]))) if isinstance(a[0], Quantity) else int(round(float(a[0])))
                return {"float": float, "int": int}[f](a[0])
            # declaration:  name(length, 'f', ...)
            if f in new or f not in self.vars:
                if not n.args:
                    # name(unit='ADC') and the like: keywords only, the shape comes from the processor (reference :1076-1130)
                    if n.keywords and all(k.arg in ("unit", "period", "offset", "grid", "dtype") for k in n.keywords):
                        v = self.vars.get(f)
                        if v is None:
                            v = Var(f, None)
                            self.vars[f] = v
                        return v
                    raise ProcessingChainError(f"declaration '{src}' needs a shape")
                shape = self._eval(n.args[0], src, new)
                if isinstance(shape, Quantity):
                    raise ProcessingChainError(f"shape in '{src}' has time units; divide by a period")
                dtype = np.float32
                if len(n.args) > 1:
                    d = self._eval(n.args[1], src, new)
                    dtype = np.dtype(d[1] if isinstance(d, tuple) else d)
                v = Var(f, "wf", int(round(float(shape))), dtype)
                self.vars[f] = v
                return v
        raise ProcessingChainError(f"could not parse argument '{src}'")

    def _const_int(self, node, src, new, default):
        if node is None:
            return default
        v = self._eval(node, src, new)
        if isinstance(v, Quantity):
            raise ProcessingChainError(f"slice bound with time units in '{src}'; divide by a period")
        if isinstance(v, (Var, tuple)):
            raise NotImplementedError(f"slice bounds must be constants: '{src}'")
        return int(v)

    def _binop(self, op, a, b):
        if isinstance(a, (Var, tuple)) or isinstance(b, (Var, tuple)):
            # per-event scalar (+|-) constant  ->  SCALAR_AFFINE; anything else is outside the subset
            if isinstance(a, Var) and a.kind == "scalar" and isinstance(b, (int, float)) and isinstance(op, (ast.Add, ast.Sub)):
                return ("affine", a, 1.0, b if isinstance(op, ast.Add) else -b)
            if isinstance(b, Var) and b.kind == "scalar" and isinstance(a, (int, float)) and isinstance(op, ast.Add):
                return ("affine", b, 1.0, a)
            # per-event scalar (*|/) constant: a multiplication; a division only where multiplying by the reciprocal is the same
            # operation bit for bit (powers of two: tp_aoe_max / 16 in icpc-dsp-config.json:344)
            if isinstance(a, Var) and a.kind == "scalar" and isinstance(b, (int, float)) and isinstance(op, ast.Mult):
                return ("affine", a, float(b), 0.0)
            if isinstance(b, Var) and b.kind == "scalar" and isinstance(a, (int, float)) and isinstance(op, ast.Mult):
                return ("affine", b, float(a), 0.0)
            if isinstance(a, Var) and a.kind == "scalar" and isinstance(b, (int, float)) and isinstance(op, ast.Div) and b != 0:
                m, e = np.frexp(abs(float(b)))
                if m == 0.5:
                    return ("affine", a, 1.0 / float(b), 0.0)
            raise NotImplementedError("expressions on waveforms / between per-event variables are not supported on the device")
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
            return Quantity(r)
        if all(isinstance(x, int) and not isinstance(x, bool) for x in (a, b)) and not isinstance(op, ast.Div):
            return int(r)
        return r


def build_processing_chain(processors, tb_in=None, db_dict=None, outputs=None, block_width: int = 16):
    """Translate a dspeed recipe into a device chain.

    Returns ``(proc_chain, field_mask, tb_out)`` like the reference (processing_chain.py:2363-2369): ``tb_in`` is a
    mapping ``name -> ndarray | DeviceArray | WaveformInput``; ``tb_out`` a dict of freshly allocated NumPy arrays for
    the requested outputs; ``field_mask`` the input columns actually used.  ``block_width`` is accepted for
    signature compatibility: the device processes the whole buffer in one launch.
    """
    del block_width
    recipe = _load(processors)
    if outputs is None:
        if "outputs" not in recipe:
            raise ValueError("outputs not provided")
        outputs = recipe["outputs"]
    nodes = dict(recipe["processors"]) if "processors" in recipe else dict(recipe)
    nodes.pop("outputs", None)

    multi = {}
    for key in list(nodes):
        keys = [k for k in re.split(",| ", key) if k]
        if len(keys) > 1:
            for k in keys:
                multi[k] = key
        node = _normalise(key, nodes[key])
        nodes[key] = node
        _substitute_db(node, db_dict or {})
        if "prereqs" not in node:
            pre = []
            for arg in node["args"]:
                if isinstance(arg, str):
                    for nm in _names_in(arg):
                        if nm not in pre and nm not in keys:
                            pre.append(nm)
            node["prereqs"] = pre
    nodes.update(multi)

    order, leafs = [], []

    def resolve(par, unresolved):
        if par in order:
            return
        if par in unresolved:
            raise ProcessingChainError(f"Circular references detected for parameter '{par}'")
        node = nodes.get(par)
        if node is None:
            if par not in leafs:
                leafs.append(par)
            return
        if isinstance(node, str):
            resolve(node, unresolved)
            return
        unresolved.append(par)
        for edge in node["prereqs"]:
            resolve(edge, unresolved)
        order.append(par)
        unresolved.remove(par)

    copy_pars, out_pars = [], []
    for o in outputs:
        if o not in nodes:
            copy_pars.append(o)
        else:
            resolve(o, [])
            out_pars.append(o)

    b = _Builder(tb_in, db_dict)
    for leaf in leafs:
        if tb_in is None or leaf not in tb_in:
            raise ProcessingChainError(f"'{leaf}' not found in input table or recipe")
        b.input_var(leaf)

    proc_strings = []
    for key in order:
        node = nodes[key]
        new_vars = [k for k in re.split(",| ", key) if k]
        try:
            _add_step(b, key, node, new_vars, proc_strings)
        except (ProcessingChainError, NotImplementedError, DSPFatal):
            raise
        except Exception as e:
            raise ProcessingChainError("Exception raised while attempting to add processor:\n" + json.dumps(node, indent=2, default=str)) from e

    n_rows = 0
    if tb_in:
        n_rows = len(_column(tb_in, next(iter(tb_in))))
    chain, tb_out = _compile(b, out_pars, n_rows, proc_strings)
    for c in copy_pars:
        if tb_in is not None and c in tb_in:
            tb_out[c] = _column(tb_in, c)
    chain.link(tb_in, tb_out)
    return chain, leafs + copy_pars, tb_out


def _add_step(b: _Builder, key, node, new_vars, proc_strings):
    module, function = node["module"], node["function"]
    if module is None:  # inline expression: alias / constant
        val = b.eval_arg(node["args"][0])
        if isinstance(val, (Var, tuple)):
            b.steps.append(("alias", [val], new_vars[0]))
            b.vars[new_vars[0]] = Var(new_vars[0], "scalar") if not isinstance(val, Var) else val
        else:
            b.vars[new_vars[0]] = Var(new_vars[0], "const", const=val)
        return
    if module not in _MODULES:
        raise NotImplementedError(f"module '{module}' is not available on the device path (processor {module}.{function})")
    if module in ("numpy", "np") and function not in ("amax", "add"):
        raise NotImplementedError(f"numpy.{function} is not available on the device path")
    args = [b.eval_arg(a, new_vars) for a in node["args"]]
    if function in _GENERATORS:
        _fold_generator(b, function, args, new_vars)
        return
    if function not in _SIGS:
        raise NotImplementedError(f"processor '{function}' is not implemented on the device path")
    roles = _SIGS[function]
    if len(args) != len(roles):
        raise ProcessingChainError(f"{function} takes {len(roles)} arguments ({len(args)} given) for parameter {key}")
    # give the variables this processor creates their type now, so later recipe entries can slice / measure them
    src_len = src_period = None
    for a, r in zip(args, roles):
        if r == "w":
            if isinstance(a, tuple) and a[0] == "slice":
                src_len, src_period = a[3] - a[2], a[1].period
            elif isinstance(a, Var):
                src_len, src_period = a.length, a.period
    for a, r in zip(args, roles):
        if r == "W" and isinstance(a, Var):
            if a.kind is None:
                a.kind = "wf"
            if a.length is None and function not in ("discrete_wavelet_transform", "convolve_wf", "fft_convolve_wf", "windower", "avg_current", "upsampler"):
                a.length = src_len
            if a.period is None:
                a.period = src_period
            a.dtype = np.dtype(np.float32)
        elif r == "S" and isinstance(a, Var) and a.kind is None:
            a.kind = "scalar"
    b.steps.append((function, args, key))
    proc_strings.append(f"{function}({', '.join(str(a.name if isinstance(a, Var) else a) for a in args)})")


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
        if v.is_input and v.dtype is not None:
            if v.kind == "wf" and v.dtype in (np.dtype(np.float64), np.dtype(np.int32), np.dtype(np.uint32)):
                return np.dtype(np.float64)
            if v.kind == "scalar" and v.dtype == np.dtype(np.float64):
                return np.dtype(np.float64)
    return np.dtype(np.float32)


def _compile(b: _Builder, out_pars, n_rows, proc_strings):
    p = Program()
    ft = _loop_dtype(b)
    in_bind, out_bind, consts = {}, {}, {}
    steps = b.steps

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
    def slices_of(v):
        found, plain = set(), False
        for _fn, a2, _k in steps:
            for x in a2:
                if isinstance(x, tuple) and x[0] == "slice" and x[1] is v:
                    found.add((x[2], x[3]))
                elif x is v or (isinstance(x, tuple) and x[0] == "affine" and x[1] is v):
                    plain = True
        return found, plain

    for si, (fn, args, key) in enumerate(steps):
        if fn != "bl_subtract" or not isinstance(args[0], Var) or not isinstance(args[-1], Var) or args[-1].name in out_pars:
            continue
        src_v, dst_v = args[0], args[-1]
        found, plain = slices_of(dst_v)
        uses_of_dst = sum(1 for _fn, a2, _k in steps for x in a2 if x is dst_v)  # the producing step itself counts once
        if len(found) != 1 or uses_of_dst != 1 or not src_v.is_input or src_v.kind != "wf":
            continue
        (lo, hi), = found
        if not (0 <= lo < hi <= (src_v.length or 0)):
            continue
        new_args = list(args)
        new_args[0] = ("slice", src_v, lo, hi)
        steps[si] = (fn, new_args, key)
        dst_v.length = hi - lo
        for sj, (fn2, a2, k2) in enumerate(steps):
            if sj != si:
                steps[sj] = (fn2, [dst_v if (isinstance(x, tuple) and x[0] == "slice" and x[1] is dst_v) else x for x in a2], k2)

    last_use = {}
    for si, (fn, args, _) in enumerate(steps):
        roles = _SIGS.get(fn, "")
        for a, r in zip(args, roles):
            v = wf_of(a)
            if v is not None and r in "wts":
                last_use[v.name] = si
            if isinstance(a, tuple) and a[0] == "affine":
                last_use[a[1].name] = si
    for o in out_pars:
        last_use[o] = len(steps) + 1

    free_slots, slot_len = [], []

    def new_slot(length):
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
                    v = Var(key, "wf", hi - lo, base.dtype, base.period, source=base.source, offset=lo)
                    b.vars[key] = v
                    last_use[key] = last_use.get(base.name, si)
                return ensure_loaded(v, si)
            key = f"{base.name}[{lo}:{hi}]"
            v = b.vars.get(key)
            if v is not None and v.slot is not None:
                return v  # the same slice was materialised for an earlier processor and is still alive
            src = ensure_loaded(base, si)
            v = Var(key, "wf", hi - lo, np.float32, base.period)
            v.slot = new_slot(v.length)
            p.add_op(_lib.OP_COPY, dst=v.slot, src=src.slot, ip=(lo,))
            b.vars[key] = v
            last_use[key] = max(sj for sj, (_, a2, _k) in enumerate(steps)
                                for x in a2 if isinstance(x, tuple) and x[0] == "slice" and x[1] is base and x[2] == lo and x[3] == hi)
            return v
        v = a
        if v.kind != "wf":
            raise ProcessingChainError(f"'{v.name}' is not a waveform")
        if v.slot is None:
            if not v.is_input:
                raise ProcessingChainError(f"waveform '{v.name}' is used before it is computed")
            col = _column(b.tb_in, v.source)
            full_len = col.shape[1]
            io = p.add_io(f"in:{v.name}", _lib.IO_WF_IN, col.dtype, v.length, v.offset, full_len)
            in_bind[f"in:{v.name}"] = v
            v.slot = new_slot(v.length)
            p.add_op(_lib.OP_LOAD, dst=v.slot, io=io)
        return v

    def scalar_operand(a, args, integer=False, what=""):
        """Scalar argument -> Scalar (const / input column / register)."""
        if isinstance(a, tuple) and a[0] == "affine":
            _, base, mul, add = a
            if isinstance(add, Quantity):
                per = period_of(args)
                if per is None:
                    raise ProcessingChainError(f"{what}: time quantity without a sampling period")
                add = float(add) / per
            src = scalar_operand(base, args)
            r = p.add_sregs(1)
            p.add_op(_lib.OP_SCALAR_AFFINE, dst=r, sp=(src, Scalar.const(mul), Scalar.const(add)))
            return Scalar.reg(r)
        if isinstance(a, Var):
            if a.kind == "const":
                a = a.const
            elif a.kind == "scalar":
                if a.sreg is not None:
                    return Scalar.reg(a.sreg)
                if a.is_input:
                    if a.io is None:
                        col = _column(b.tb_in, a.source)
                        a.io = p.add_io(f"in:{a.name}", _lib.IO_SCALAR_IN, col.dtype)
                        in_bind[f"in:{a.name}"] = a
                    return Scalar.input(a.io)
                raise ProcessingChainError(f"scalar '{a.name}' is used before it is computed")
            else:
                raise ProcessingChainError(f"{what}: '{a.name}' is not a scalar")
        if isinstance(a, Quantity):
            per = period_of(args)
            if per is None:
                raise ProcessingChainError(f"{what}: time quantity without a sampling period (wrap the input in WaveformInput)")
            a = float(a) / per
            if integer:
                a = int(round(a))  # reference :1747-1770: integer parameters are rounded after the unit conversion
        if integer:
            if float(a) != int(a) and not isinstance(a, int):
                a = int(round(float(a)))
            return int(a)
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
        if src_var is not None and a.period is None:
            a.period = src_var.period
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
        if fn in ("bl_subtract", "pole_zero", "double_pole_zero"):
            src = ensure_loaded(args[0], si)
            dst = out_wf(args[-1], src.length, src)
            inplace = last_use.get(src.name, -1) <= si
            dst.slot = src.slot if inplace else new_slot(src.length)
            if fn == "bl_subtract":
                p.add_op(_lib.OP_BL_SUBTRACT, dst=dst.slot, src=src.slot, sp=(scalar_operand(args[1], args, what=what),))
            elif fn == "pole_zero":
                tau = scalar_operand(args[1], args, what=what)
                p.add_op(_lib.OP_POLE_ZERO, dst=dst.slot, src=src.slot, sp=(tau,))
            else:
                sp = tuple(scalar_operand(a, args, what=what) for a in args[1:4])
                p.add_op(_lib.OP_DOUBLE_POLE_ZERO, dst=dst.slot, src=src.slot, sp=sp)
            if not inplace:
                release(src, si)
        elif fn in trap_ops:
            src = ensure_loaded(args[0], si)
            ints = [scalar_operand(a, args, integer=True, what=what) for a in args[1:-1]]
            ints += [0] * (3 - len(ints))
            dst = out_wf(args[-1], src.length, src)
            # fusion: the trapezoid's only consumer is the next fixed_time_pickoff and it is not an output
            nxt = steps[si + 1] if si + 1 < len(steps) else None
            if (nxt and nxt[0] == "fixed_time_pickoff" and wf_of(nxt[1][0]) is dst and last_use.get(dst.name) == si + 1
                    and dst.name not in out_pars and char_of(nxt[1][2]) != ord("s")):
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
            if (users and plain and dst.name not in out_pars and sorted(kinds) in (["min_max"], ["time_point_thresh"], ["min_max", "time_point_thresh"])
                    and not any(isinstance(x, tuple) and x[0] == "slice" and x[1] is dst for _f, a2, _k in steps for x in a2)):
                pending_reduce[dst.name] = {"src": src, "ints": ints, "kind": trap_ops[fn], "emit_at": max(users), "mm_first": -1}
                last_use[src.name] = max(last_use.get(src.name, si), max(users))
                continue
            dst.slot = new_slot(src.length)
            p.add_op(trap_ops[fn], dst=dst.slot, src=src.slot, ip=ints)
            release(src, si)
        elif fn in ("min_max", "time_point_thresh") and isinstance(args[0], Var) and args[0].name in pending_reduce:
            pr = pending_reduce[args[0].name]
            if fn == "min_max":
                pr["mm_first"] = p.add_sregs(4)
                for k, a in enumerate(args[1:5]):
                    if not isinstance(a, Var):
                        raise ProcessingChainError("min_max outputs must be variable names")
                    a.kind, a.sreg = "scalar", pr["mm_first"] + k
            else:
                pr["tpt"] = (tuple(scalar_operand(a, args, what=what) for a in args[1:4]), out_scalar(args[4]))
            if si == pr["emit_at"]:
                sp, o = pr.get("tpt", ((), None))
                p.add_op(_lib.OP_TRAP_REDUCE, dst=pr["mm_first"], src=pr["src"].slot, io=(o.sreg if o is not None else -1),
                         ip=(*pr["ints"], pr["kind"]), sp=sp)
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
            dst.slot = new_slot(src.length)
            tmp = new_slot(src.length) if num > 1 else dst.slot  # ping-pong target of the passes before the last
            p.add_op(_lib.OP_MOVING_WINDOW_MULTI, dst=dst.slot, src=src.slot, ip=(typ, num, tmp), sp=(scalar_operand(args[1], args, what=what),))
            if num > 1 and tmp not in free_slots:
                free_slots.append(tmp)
            release(src, si)
        elif fn == "add":  # numpy.add on per-event scalars (icpc-dsp-config.json:341-346): a * 1 + b, exact
            o = out_scalar(args[2])
            p.add_op(_lib.OP_SCALAR_AFFINE, dst=o.sreg, sp=(scalar_operand(args[0], args, what=what), Scalar.const(1.0),
                                                            scalar_operand(args[1], args, what=what)))
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
            src = ensure_loaded(args[0], si)
            first = p.add_sregs(4)
            for k, a in enumerate(args[1:5]):
                if not isinstance(a, Var):
                    raise ProcessingChainError("linear_slope_fit outputs must be variable names")
                a.kind, a.sreg = "scalar", first + k
            p.add_op(_lib.OP_LINEAR_SLOPE_FIT, dst=first, src=src.slot)
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
            if taps.io is None:
                taps.io = p.add_io(f"taps:{taps.name}", _lib.IO_TAPS, ft, taps.length, 0, 0)
                consts[f"taps:{taps.name}"] = taps.const.astype(ft)
            dst = out_wf(args[3], None, src)
            if dst.length is None:
                raise ProcessingChainError(f"{fn}: declare the output as name(length, 'f')")
            has_nan = int(np.isnan(taps.const).any())
            # fusion: the filtered waveform's only consumer is one numpy.amax and it is not an output -> it is never stored
            users = [sj for sj, (f2, a2, _k) in enumerate(steps) if sj != si and any(wf_of(x) is dst for x in a2)]
            if (len(users) == 1 and steps[users[0]][0] == "amax" and steps[users[0]][1][0] is dst and dst.name not in out_pars
                    and users[0] > si and users[0] not in skip):
                o = out_scalar(steps[users[0]][1][2])
                p.add_op(_lib.OP_CONVOLVE_AMAX, dst=o.sreg, src=src.slot, io=taps.io, ip=(char_of(args[2]), has_nan, int(dst.length)))
                skip.add(users[0])
                last_use[src.name] = max(last_use.get(src.name, si), si)
                release(src, si)
                continue
            dst.slot = new_slot(dst.length)
            p.add_op(_lib.OP_CONVOLVE, dst=dst.slot, src=src.slot, io=taps.io, ip=(char_of(args[2]), has_nan))
            release(src, si)
        else:
            raise NotImplementedError(f"processor '{fn}' is not implemented on the device path")

    tb_out = {}
    for o in out_pars:
        v = b.vars.get(o)
        if v is None or v.kind in (None,):
            raise ProcessingChainError(f"output '{o}' was never computed")
        if v.kind == "const":
            tb_out[o] = np.full(n_rows, v.const)
            continue
        if v.kind == "taps":
            tb_out[o] = np.broadcast_to(v.const, (n_rows, v.length)).copy()
            continue
        if v.kind == "wf":
            if v.slot is None:
                raise ProcessingChainError(f"output waveform '{o}' was never computed")
            io = p.add_io(f"out:{o}", _lib.IO_WF_OUT, ft, v.length)
            p.add_op(_lib.OP_STORE, src=v.slot, io=io)
            out_bind[f"out:{o}"] = (v, v.length)
            tb_out[o] = np.empty((n_rows, v.length), dtype=ft)
        else:
            if v.sreg is None:
                if v.is_input:
                    tb_out[o] = _column(b.tb_in, v.source)
                    continue
                raise ProcessingChainError(f"output '{o}' was never computed")
            io = p.add_io(f"out:{o}", _lib.IO_SCALAR_OUT, ft)
            p.add_op(_lib.OP_STORE_SCALAR, io=io, ip=(v.sreg,))
            out_bind[f"out:{o}"] = (v, None)
            tb_out[o] = np.empty(n_rows, dtype=ft)
    p.slots = slot_len
    if len(p.ops) > _lib.MAX_OPS or len(p.slots) > _lib.MAX_SLOTS or len(p.io) > _lib.MAX_IO or p.n_sregs > _lib.MAX_SREGS:
        raise NotImplementedError("recipe is too large for one device chain (ops/slots/bindings limit)")
    chain = ProcessingChain(p, in_bind, out_bind, consts, n_rows, proc_strings, ft)
    return chain, tb_out


def shard_rows(n_rows: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous split of the event axis: rank r gets rows [r*N/G, (r+1)*N/G).  Events are independent,
    so a multi-GPU run is one chain per rank over its slice and a concatenation of the outputs -- no collective."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return (n_rows * rank) // world_size, (n_rows * (rank + 1)) // world_size
