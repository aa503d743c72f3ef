"""JSON/YAML recipe -> fused device chain: the host-side mirror of the reference's chain builder and runtime for
the hot path (reference src/dspeed/processing_chain.py: ``build_processing_chain`` :2363-2872,
``ProcessingChain.execute`` :665-673, ``_execute_procs`` :1144-1163).

What is kept from the reference (so existing LEGEND recipes for the energy chain run unmodified):

* the recipe schema: ``{"outputs": [...], "processors": {"a, b": {"function", "module", "args", "kwargs",
  "defaults", "unit", "prereqs"}}}``, the one-string form ``"module.func(arg, ...)"``, ``db.x.y`` lookups with
  ``defaults`` (:2555-2583), multi-output keys split on ``,``/space (:2480-2483), dependency resolution by
  depth-first search from the requested outputs with cycle detection (:2601-2651), constant folding of processors
  whose inputs are all constants -- how cusp/zac kernels are built once (:2775-2823);
* argument syntax (:718-1130): literals, ``'c'`` characters, ``N*us`` quantities, output declarations ``name(length, 'f',
  grid=..., unit=..., period=..., offset=...)``, constant slices ``wf[a:b]`` (bounds may be times), ``len(wf)``,
  ``round/floor/ceil/trunc(x, to_nearest)``, ``wf.period`` / ``.offset`` / ``.grid``, arithmetic on constants and ``+ - * /``
  between per-event variables, constants and times (one scalar op each, like the reference's one ufunc processor each), inline
  definitions (``"QDrift": "trapQftp * 16"``);
* units and coordinate grids (:67-144, :1556-1732, :1806-1908): a waveform has a grid (period, offset) -- the input's from
  ``WaveformInput(values, dt, t0)`` --, a processor works on the grid of its first waveform argument, per-event variables with a
  time unit are sample indices on that grid, converted when another grid reads them and written in their unit;
* the literal module string ``dspeed.processors`` (and ``numpy`` for ``amax`` / ``add``) resolves to this package's registry.

What is different by design: instead of calling one gufunc per processor per 16-row block, the resolved processor
list is translated into ONE device program (``dsp_chain_create``) that keeps every intermediate waveform in LDS, and
``execute`` launches it over the whole buffer; the processors are ordered for short waveform lifetimes (``_schedule``), which
is a reordering of pure functions within their dependencies.  Anything outside the supported subset raises
``ProcessingChainError``/``NotImplementedError`` -- there is no CPU fallback.
"""
from __future__ import annotations

import ast
import json
import queue
import re
import threading
import time
from collections.abc import MutableMapping
import math
import logging
import os
from copy import deepcopy
from types import SimpleNamespace

import numpy as np

from . import _lib
from .chain import Chain, Program, Scalar
from .device import DeviceArray, Event, HostPin, PinnedArray, Stream, dtype_code, set_device
from .errors import DSPFatal, ProcessingChainError
from .recipe import LANGUAGE_CALLS as _CALLS, Recipe

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


_COPY_POOL = None  # host threads that move rows between NumPy columns and staging buffers
_COPY_POOL_LOCK = threading.Lock()


class ProcessingChain:
    """Runs a translated recipe over a buffer of rows.  ``execute(start, stop)`` has the meaning of the reference's
    (processing_chain.py:665-673); ``__call__(tb_in, tb_out)`` relinks I/O like :675-716."""

    def __init__(self, program: Program, inputs: dict, outputs: dict, consts: dict, buffer_len: int, proc_strings: list[str],
                 loop_dtype=np.float32, aux=(), stages=(), ext_alias=None, tail=None):
        self._program = program
        self._tail = tail         # the program's all-scalar tail as a program of its own, run behind it with a row per lane (_split_scalar_tail)
        self.loop_dtype = np.dtype(loop_dtype)  # float32 or float64 gufunc loop of the whole chain
        self._in_vars = inputs      # binding name -> Var (source column)
        self._out_vars = outputs    # binding name -> (Var, length or None)
        self._consts = consts       # binding name -> ndarray (taps)
        self._aux = list(aux)       # fits done on the rows ahead of the chain (dsp_linear_slope_fit_rows), results bound as inputs
        self._aux_bufs = {}
        self._stages = list(stages)  # programs that run ahead of the main one and leave rows / columns in HBM for it (_extract_stages)
        self._ext_alias = dict(ext_alias or {})  # binding of the program -> buffer a fit or a stage filled
        self._buffer_len = buffer_len
        self._chain = None
        self._stream = None
        self._lanes = []          # [0] = the chain itself; further lanes for pieces of a batch that run at the same time (_lane)
        self._pending = None      # (start, stop, lane) of a pass queued with execute(wait=False)
        self._dev = {}
        self._tb_in = None
        self._tb_out = None
        self._pins = {}           # (address, bytes) -> HostPin of a linked host column (None: registration refused)
        self._piece_key, self._piece_bufs, self._piece_events = None, [], []  # device buffers host columns are streamed through
        self._copy_stream = None  # H2D of the next piece runs here while the compute stream works on the current one
        self._stage_key, self._stage = None, []  # page-locked staging buffers (one set per piece slot)
        self._timing = {"h2d": 0.0, "kernel": 0.0, "d2h": 0.0}
        self._copy_pars = []      # outputs that are input columns handed through
        self.vector_lens = {}     # variable-length outputs -> input column with their per-event lengths
        self.output_attrs = {}    # output -> attributes of its LGDO column (units, lh5_attrs, description)
        self.proc_strings = proc_strings
        self.device = None        # GPU ordinal the chain is bound to (None: the current device of the thread that first executes it)

    # -- introspection
    @property
    def program(self) -> Program:
        return self._program

    def get_timing(self) -> dict:
        return dict(self._timing)

    def kernels(self) -> list:
        """Which kernel every launch of a pass runs on, in launch order: [(what, kernel name)] -- the fits on the rows, the stages ahead of
        the program, the program, its scalar tail.  A stage or program that misses the specialised shapes shows up here
        as ``dsp_vm_kernel``: the place to look when a recipe is slower than its neighbours."""
        self._ensure()
        rep = [(f"linear_slope_fit x {len(g['fits'])} on the rows of {g['wf']}", "dsp_fit_rows_kernel") for g in self._aux]
        rep += [(st["what"], st["chain"].kernel_name) for st in self._stages]
        rep.append(("program", self._chain.kernel_name))
        if self._lanes and self._lanes[0].tail is not None:
            rep.append(("scalar tail of the program", self._lanes[0].tail.kernel_name))
        return rep

    def geometry(self, n_rows: int) -> dict:
        """launch geometry of the program's kernel for a pass over ``n_rows`` rows (``dsp_chain_geometry``)"""
        self._ensure()
        return self._chain.geometry(n_rows)

    def kernel_notes(self) -> list:
        """[(what, reason)] for every stage / program that runs on the generic interpreter although its ops are those of a specialised kernel
        (``dsp_chain_kernel_note``): a time constant per event, a length or alignment the kernel does not take, a kernel of fewer than 64 taps.
        Logged as a warning when the chain is first set up -- such a chain is correct and 2 - 4 x slower than its neighbour."""
        self._ensure()
        chains = [(st["what"], st["chain"]) for st in self._stages] + [("program", self._chain)]
        return [(what, ch.kernel_note) for what, ch in chains if ch.kernel_note]

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
            self._chain.set_async_check(True)  # (pieces of a host-resident batch: the check must not wait behind the next piece's transfer)
            self._stream = Stream()
            for name, arr in self._consts.items():
                self._dev[name] = DeviceArray.from_numpy(arr)
            for k, st in enumerate(self._stages):
                st["chain"] = Chain(st["program"], f"processing_chain stage {k} ({st['what']})", st.get("compute", self.loop_dtype))
                st["chain"].set_async_check(True)
                st["dev"] = {name: DeviceArray.from_numpy(arr) for name, arr in st["consts"].items()}
            self._lanes = [SimpleNamespace(stream=self._stream, chain=self._chain, stage_chains=[st["chain"] for st in self._stages],
                                           stage_bufs=[st["bufs"] for st in self._stages], aux_bufs=self._aux_bufs, tail=self._tail_chain(0),
                                           tail_bufs={})]
            self._pair_stages(self._lanes[0].stage_chains)
            for what, ch in [(st["what"], st["chain"]) for st in self._stages] + [("program", self._chain)]:
                if ch.kernel_note:
                    log.warning("%s runs on the generic interpreter (%s): %s", what, ch.kernel_name, ch.kernel_note)

    def _pair_stages(self, stage_chains) -> None:
        """a stage that writes pole-zero rows and a float16 FIR stage that reads them: the rows' scales travel with the rows (the C side
        takes the pair only if the two kernels are of those kinds, and checks at every execute that the rows are the same)"""
        if os.environ.get("DSPEED_HIP_NO_SHARED_ROW_SCALES", "0") == "1":
            return
        for i, st in enumerate(self._stages):
            made = {key for _o, key, _l in st["outs"]}
            for j in range(i + 1, len(self._stages)):
                if made & set(self._stages[j]["alias"].values()):
                    stage_chains[i].share_row_scales(stage_chains[j])

    def _tail_chain(self, lane_no: int):
        if self._tail is None:
            return None
        ch = Chain(self._tail["program"], f"processing_chain scalar tail (lane {lane_no})", self.loop_dtype)
        ch.set_async_check(True)
        return ch

    def _run_tail(self, bufs: dict, m: int, stream, lane) -> None:
        """the main program's scalar tail, behind it on its stream: the registers it takes over arrive as columns"""
        if lane.tail is None:
            return
        lane.tail.execute(bufs, m, stream)

    def _handover_bufs(self, bufs: dict, m: int, lane) -> None:
        if self._tail is None:
            return
        for name in self._tail["handover"]:
            buf = lane.tail_bufs.get(name)
            if buf is None or buf.shape[0] < m:
                buf = lane.tail_bufs[name] = DeviceArray((m,), self.loop_dtype)
            bufs[name] = buf

    def _lane(self, k: int):
        """Lane k of the chain: its own handles (error words), compute stream and intermediate buffers, so that the kernels of two pieces of
        a batch can be on the device at the same time.  Lane 0 is the chain itself; the others are made when a batch first needs them."""
        self._ensure()
        while len(self._lanes) <= k:
            ch = Chain(self._program, f"processing_chain (lane {len(self._lanes)})", self.loop_dtype)
            ch.set_async_check(True)
            stage_chains = []
            for j, st in enumerate(self._stages):
                c = Chain(st["program"], f"processing_chain stage {j} ({st['what']}, lane {len(self._lanes)})", st.get("compute", self.loop_dtype))
                c.set_async_check(True)
                stage_chains.append(c)
            self._pair_stages(stage_chains)
            self._lanes.append(SimpleNamespace(stream=Stream(), chain=ch, stage_chains=stage_chains, stage_bufs=[{} for _ in self._stages],
                                               aux_bufs={}, tail=self._tail_chain(len(self._lanes)), tail_bufs={}))
        return self._lanes[k]

    #: bytes of host-resident I/O per pipelined piece, two pieces in flight.  Tens of MB are enough for the PCIe transfers; the size is set
    #: by the kernels that run one waveform per lane (the fits of a whole recipe take 4 ms whether a piece has 4 000 rows or 60 000:
    #: tools/e2e_recipe_rate.py, 0.31 M waveforms/s with 64 MiB pieces, 0.81 M with 256 MiB)
    pipeline_bytes = 256 << 20
    #: upper bound of the intermediate rows the stages ahead of the program keep in HBM (per piece)
    stage_bytes = 16 << 30
    #: How host-resident columns reach the device.  False (default): through page-locked staging buffers the chain owns
    #: (hipHostMalloc), filled / emptied by host threads -- the device only exchanges data with memory the runtime allocated itself.
    #: True: the linked NumPy columns are page-locked in place (hipHostRegister) and copied from directly: the full PCIe rate
    #: without a host copy, for buffers that live as long as the chain (build_dsp-style refilled tables); profiles/design_diary_r01_r03.md, "Host memory and the runtime", on why it
    #: is not the default.  Environment DSPEED_HIP_PIN_IN_PLACE=1 switches it on globally.
    pin_in_place = os.environ.get("DSPEED_HIP_PIN_IN_PLACE", "0") == "1"
    #: host threads that move rows between NumPy columns and the staging buffers (tools/host_copy_rate.py: 8 threads move 79 GB/s in 32 MiB
    #: blocks, 12 threads 145 GB/s in 256 MiB blocks; the link takes 53)
    copy_threads = 12
    #: pieces of a host-resident batch whose kernels may be on the device at the same time (each on its own stream and chain handles)
    pieces_in_flight = 2

    def _host_copy(self, dst: np.ndarray, src: np.ndarray) -> None:
        """dst[...] = src for two equally shaped row blocks, split over the copy threads (NumPy releases the GIL in the copy)."""
        n = len(src)
        if n == 0:
            return
        if src.nbytes < (4 << 20) or self.copy_threads <= 1:
            np.copyto(dst, src, casting="unsafe")
            return
        global _COPY_POOL
        with _COPY_POOL_LOCK:
            if _COPY_POOL is None or _COPY_POOL._max_workers < self.copy_threads:  # (one pool for all chains of the process)
                from concurrent.futures import ThreadPoolExecutor

                _COPY_POOL = ThreadPoolExecutor(max_workers=self.copy_threads, thread_name_prefix="dspeed-copy")
        step = -(-n // self.copy_threads)
        list(_COPY_POOL.map(lambda a: np.copyto(dst[a:a + step], src[a:a + step], casting="unsafe"), range(0, n, step)))

    def _pinned(self, arr: np.ndarray) -> bool:
        """Page-lock a linked host column in place, once (the reference's build_dsp refills the same buffers for every file
        chunk).  Registration can be refused (read-only or already registered memory): copies then run unpinned, only slower."""
        if not self.pin_in_place:
            return False
        if arr.nbytes < (1 << 20):  # small columns: the copy is latency, not bandwidth; and they share pages with their neighbours
            return False
        key = (arr.ctypes.data, arr.nbytes)
        if key not in self._pins:
            try:
                self._pins[key] = HostPin(arr)
            except Exception:
                self._pins[key] = None
        return self._pins[key] is not None

    @property
    def stream(self) -> Stream:
        """the stream the chain's kernels are launched on (lane 0: what a pass over device-resident columns uses) -- for HIP events around a pass"""
        self._ensure()
        return self._stream

    def wait(self) -> None:
        """Finish a pass started with ``execute(..., wait=False)``: wait for its kernels and raise the DSPFatal a row met, as ``execute`` itself
        does otherwise."""
        pending, self._pending = self._pending, None
        if pending is None:
            return
        a, b, lane = pending
        try:
            for ch in lane.stage_chains:
                ch.check(lane.stream, row_offset=a)
            lane.chain.check(lane.stream, row_offset=a)
        except DSPFatal as e:  # the reference annotates and re-raises (processing_chain.py:1154-1159)
            if e.wf_range is None:
                e.wf_range = range(a, b)
            raise

    def execute(self, start: int = 0, stop: int | None = None, wait: bool = True) -> None:
        """``wait=False`` (columns resident on the device only): the pass is queued on ``stream`` and the call returns; ``wait()`` -- or the next
        ``execute`` -- finishes it.  The device's error word keeps the first DSPFatal of the passes queued since the last check."""
        if stop is None:
            stop = self._buffer_len
        n = stop - start
        if n <= 0:
            return
        if self.device is not None:
            set_device(self.device)
        self._ensure()
        lib = _lib.lib()
        # ---- sort the linked columns: device-resident ones are used in place, host ones are streamed through piece buffers
        dev_in, host_in, dev_out, host_out = {}, {}, {}, {}
        same_col = {}  # binding -> the binding of the same column that is sent (a stage's slice of the waveform, the fits' view of it)
        first_of = {}
        for name, var in self._in_vars.items():
            col = _column(self._tb_in, var.source)
            if isinstance(col, DeviceArray):
                dev_in[name] = col
            elif var.source in first_of:
                same_col[name] = first_of[var.source]
            else:
                first_of[var.source] = name
                a = np.asarray(col)
                if not a.flags.c_contiguous:
                    a = np.ascontiguousarray(a)
                else:
                    self._pinned(a)
                host_in[name] = a
        odt = {name: np.dtype(getattr(var, "dtype", None) or self.loop_dtype) for name, (var, _l) in self._out_vars.items()}
        for name, (var, length) in self._out_vars.items():
            col = self._tb_out[var.name]
            if isinstance(col, DeviceArray):
                dev_out[name] = col
            else:
                direct = isinstance(col, np.ndarray) and col.flags.c_contiguous and col.dtype == odt[name]
                if direct:
                    self._pinned(col)
                host_out[name] = (col, length, direct)
        row_bytes = sum(a.nbytes // max(len(a), 1) for a in host_in.values())
        row_bytes += sum(odt[nm].itemsize * (1 if length is None else length) for nm, (_, length, _) in host_out.items())
        piece = n if row_bytes == 0 else int(max(1, min(n, self.pipeline_bytes // row_bytes)))
        # what the stages ahead of the program leave in HBM (the pole-zero corrected waveform, filtered waveforms: 64 kB per row of the Ge
        # recipe) is per piece: a device-resident batch of a million rows is walked in pieces so that those buffers stay bounded
        stage_row_bytes = sum(self.loop_dtype.itemsize * (1 if length is None else length) for st in self._stages for _o, _k, length in st["outs"])
        if stage_row_bytes and piece > self.stage_bytes // stage_row_bytes:
            cap = max(1, self.stage_bytes // stage_row_bytes)
            piece = -(-n // -(-n // cap))  # (equal pieces: no short last one)
        # Host-resident batches of more than two pieces start with a quarter and a half piece and end with a half piece: the device idles
        # while the first piece crosses the link and the link idles while the last piece is processed, so both are made short.
        ramp = row_bytes > 0 and n > 2 * piece and piece >= 4
        if ramp:
            sizes = [piece // 4, piece // 2]
            tail = piece // 2
            body = n - sum(sizes) - tail
            left = body % piece
            if left <= piece - tail:  # (a short piece costs the device as much as a long one: what is left over joins the last piece)
                left, tail = 0, tail + left
            sizes += [piece] * (body // piece) + ([left] if left else []) + [tail]
        else:
            sizes = [piece] * (n // piece) + ([n % piece] if n % piece else [])
        n_pieces = len(sizes)
        if row_bytes == 0 and n_pieces == 1:
            # every column lives on the device and the stages' rows fit: the launches of one pass on the chain's own stream, no staging, no
            # feeder thread (a 3 ms pass of the energy chain does not pay for either)
            lane = self._lane(0)
            if self._pending is not None and self._pending[:2] != (start, stop):
                self.wait()  # (a DSPFatal's row is counted from the first row of its pass: queued passes must cover the same rows)
            bufs = dict(self._dev)
            for name, col in dev_in.items():
                bufs[name] = col.view_rows(start, stop)
            for name, col in dev_out.items():
                bufs[name] = col.view_rows(start, stop)
            t = time.perf_counter()
            self._run_aux(bufs, n, lane.stream, lane)
            self._handover_bufs(bufs, n, lane)
            lane.chain.execute(bufs, n, lane.stream)
            self._run_tail(bufs, n, lane.stream, lane)
            self._pending = (start, stop, lane)
            if wait:
                self.wait()
                self._timing["kernel"] += time.perf_counter() - t
            return
        if not wait:
            raise ValueError("execute(wait=False) needs every linked column on the device (host columns are streamed in pieces and finished in order)")
        self.wait()
        n_lanes = min(self.pieces_in_flight, n_pieces)  # pieces whose kernels are on the device at the same time
        n_slots = min(n_lanes + 2, n_pieces)            # ... + the one on the link + the one being copied
        # piece buffers live as long as the chain (the reference pre-allocates its ProcChainVar buffers the same way,
        # processing_chain.py:259-269): build_dsp calls execute() once per file chunk with the same shapes
        key = (piece, n_slots, tuple((nm, a.shape[1:], a.dtype.str) for nm, a in host_in.items()),
               tuple((nm, length) for nm, (_, length, _) in host_out.items()))
        if self._piece_key != key:
            self._piece_bufs = []
            for _ in range(n_slots):
                sl = {name: DeviceArray((piece, *a.shape[1:]), a.dtype) for name, a in host_in.items()}
                sl.update({name: DeviceArray((piece,) if length is None else (piece, length), odt[name])
                           for name, (_, length, _) in host_out.items()})
                self._piece_bufs.append(sl)
            self._piece_events = [Event() for _ in range(n_slots)]
            self._piece_key = key
        slots, ev_in = self._piece_bufs, self._piece_events
        # page-locked staging buffers for the columns that are not page-locked in place: one per piece slot
        in_place_in = {name for name, arr in host_in.items() if self._pinned(arr)}
        in_place_out = {name for name, (col, _len, direct) in host_out.items() if direct and self._pinned(col)}
        skey = (key, tuple(sorted(in_place_in)), tuple(sorted(in_place_out)))
        if self._stage_key != skey:
            self._stage = []
            for _ in range(n_slots):
                st = {name: PinnedArray((piece, *arr.shape[1:]), arr.dtype) for name, arr in host_in.items() if name not in in_place_in}
                st.update({name: PinnedArray((piece,) if length is None else (piece, length), odt[name])
                           for name, (_c, length, _d) in host_out.items() if name not in in_place_out})
                self._stage.append(st)
            self._stage_key = skey
        stage = self._stage
        if self._copy_stream is None:
            self._copy_stream = Stream()
        s_in = self._copy_stream
        lanes = [self._lane(j) for j in range(n_lanes)]
        pieces, at = [], start
        for size in sizes:
            pieces.append((at, at + size))
            at += size

        def finish(a, b, staged, lane):
            """piece [a, b): wait for its kernel and copies, report its DSPFatal with absolute rows, deliver the staged outputs"""
            t = time.perf_counter()
            try:
                for ch in lane.stage_chains:
                    ch.check(lane.stream, row_offset=a)
                lane.chain.check(lane.stream, row_offset=a)
            except DSPFatal as e:  # the reference annotates and re-raises (processing_chain.py:1154-1159)
                if e.wf_range is None:
                    e.wf_range = range(a, b)
                raise
            self._timing["kernel"] += time.perf_counter() - t
            t = time.perf_counter()
            for col, buf in staged:  # (a column of another dtype is converted on the way)
                self._host_copy(col[a:b], buf[:b - a])
            self._timing["d2h"] += time.perf_counter() - t

        # ---- the feeder: one host thread walks the pieces ahead of the device.  It copies a block of rows into the page-locked staging
        # buffer of the piece's slot (split over the copy threads) and queues its transfer on the copy stream, block after block: the
        # transfer of block j runs while block j + 1 is being copied, across piece boundaries, so the link sees one continuous stream of
        # rows at min(host copy rate, PCIe rate).  Slots: the pieces being processed (one per lane), one on the link, one being copied.  A
        # slot is handed back when its piece has been finished (its kernels read the slot's device buffers until then).
        # Two lanes: the kernels of piece k + 1 are queued (on their own stream, with their own chain handles) while piece k still runs --
        # the kernels that give every waveform a lane fill one wavefront per compute unit for a piece of 16 k rows and take a millisecond
        # whatever the piece's size; side by side with the other piece's kernels that time is not lost, and no launch gap opens between pieces.
        BLOCK_BYTES = 64 << 20
        slot_free = [threading.Semaphore(1) for _ in range(n_slots)]
        arrived: queue.Queue = queue.Queue()  # piece index (or the feeder's exception), in order
        stop_feeding = threading.Event()

        def feed():
            try:
                set_current = self.device
                if set_current is not None:
                    set_device(set_current)
                for k, (a, b) in enumerate(pieces):
                    while not slot_free[k % n_slots].acquire(timeout=0.05):
                        if stop_feeding.is_set():
                            return
                    if stop_feeding.is_set():
                        return
                    m, sl, st = b - a, slots[k % n_slots], stage[k % n_slots]
                    t = time.perf_counter()
                    for name, arr in host_in.items():
                        d = sl[name].view_rows(0, m)
                        row_bytes_col = arr.nbytes // max(len(arr), 1)
                        step = m if name in in_place_in else max(1, min(m, BLOCK_BYTES // max(row_bytes_col, 1)))
                        for r0 in range(0, m, step):
                            if stop_feeding.is_set():
                                return
                            r1 = min(m, r0 + step)
                            if name in in_place_in:
                                src = arr[a + r0:a + r1]
                            else:
                                src = st[name].array[r0:r1]
                                self._host_copy(src, arr[a + r0:a + r1])
                            _lib.check(lib.dsp_h2d_async(d.view_rows(r0, r1).ptr, src.ctypes.data, src.nbytes, s_in.ptr), what="h2d_async")
                    ev_in[k % n_slots].record(s_in)
                    self._timing["h2d"] += time.perf_counter() - t
                    arrived.put(k)
            except BaseException as e:  # noqa: BLE001 -- re-raised by the consumer
                arrived.put(e)

        feeder = threading.Thread(target=feed, name="dspeed-feeder", daemon=True)
        feeder.start()
        completed = False
        in_flight = []  # (piece index, a, b, staged outputs, lane), oldest first
        try:
            for k, (a, b) in enumerate(pieces):
                got = arrived.get()
                if isinstance(got, BaseException):
                    raise got
                if len(in_flight) == n_lanes:  # the lane this piece takes is the oldest one's: finish that first
                    k0, a0, b0, staged0, lane0 = in_flight.pop(0)
                    finish(a0, b0, staged0, lane0)
                    slot_free[k0 % n_slots].release()
                lane = lanes[k % n_lanes]
                s_c = lane.stream
                m, sl, st = b - a, slots[k % n_slots], stage[k % n_slots]
                bufs = dict(self._dev)
                for name in host_in:
                    bufs[name] = sl[name].view_rows(0, m)
                for name, first in same_col.items():
                    bufs[name] = bufs[first]
                for name, col in dev_in.items():
                    bufs[name] = col.view_rows(a, b)
                for name, col in dev_out.items():
                    bufs[name] = col.view_rows(a, b)
                for name in host_out:
                    bufs[name] = sl[name].view_rows(0, m)
                s_c.wait_event(ev_in[k % n_slots])
                staged = []
                self._run_aux(bufs, m, s_c, lane)
                self._handover_bufs(bufs, m, lane)
                lane.chain.execute(bufs, m, s_c)
                self._run_tail(bufs, m, s_c, lane)
                for name, (col, length, direct) in host_out.items():
                    d = bufs[name]
                    if name in in_place_out:
                        dst = col[a:b]
                    else:
                        dst = st[name].array[:m]
                        staged.append((col, dst))
                    _lib.check(lib.dsp_d2h_async(dst.ctypes.data, d.ptr, d.nbytes, s_c.ptr), what="d2h_async")
                in_flight.append((k, a, b, staged, lane))
            while in_flight:
                k0, a0, b0, staged0, lane0 = in_flight.pop(0)
                finish(a0, b0, staged0, lane0)
                slot_free[k0 % n_slots].release()
            completed = True
        finally:
            stop_feeding.set()
            feeder.join()
            if not completed:  # (a failed pass leaves transfers and kernels of later pieces behind: they must not meet the next call's)
                s_in.sync()
                for ln in lanes:
                    ln.stream.sync()
                    for ch in (*ln.stage_chains, ln.chain):  # their error words belong to the abandoned pieces
                        try:
                            ch.check(ln.stream)
                        except DSPFatal:
                            pass

    def _run_aux(self, bufs: dict, m: int, stream, lane=None) -> None:
        """linear_slope_fit of the recipe that can run on the rows of the batch, one waveform per lane, ahead of the chain on its stream:
        fills the columns the chain reads as inputs."""
        lib = _lib.lib()
        lane = lane if lane is not None else self._lane(0)
        ft_code, isz = dtype_code(self.loop_dtype), self.loop_dtype.itemsize
        for gi, g in enumerate(self._aux):
            n_cols = len(g["names"])
            out = lane.aux_bufs.get(gi)
            if out is None or out.shape[1] < m:
                out = lane.aux_bufs[gi] = DeviceArray((n_cols, m), self.loop_dtype)
            wf = bufs[g["wf"]]
            wf_ptr = (wf.ptr if isinstance(wf, DeviceArray) else int(wf)) + g["lo"] * g["itemsize"]
            sub = bufs[g["sub"]] if g["sub"] is not None else None
            sub_ptr = None if sub is None else (sub.ptr if isinstance(sub, DeviceArray) else int(sub))
            fits = (_lib.FitWindow * len(g["fits"]))(*[_lib.FitWindow(*f) for f in g["fits"]])
            _lib.check(lib.dsp_linear_slope_fit_rows(wf_ptr, g["dtype"], m, g["len"], g["stride"], ft_code, sub_ptr, g["sub_dtype"], g["sub_const"],
                                                     g["mode"], int(g["tau"] is not None), float(g["tau"] or 0.0), fits, len(g["fits"]), out.ptr,
                                                     stream.ptr), what="linear_slope_fit_rows")
            for j, name in enumerate(g["names"]):
                bufs[name] = out.ptr + j * m * isz
        # The stages, in launch order (a stage may read what an earlier one wrote) -- but not all on one stream: what a stage reads says which
        # earlier stages it waits for, and stages that do not wait for each other (the cusp filter beside the t0 chain, the reductions off the raw
        # rows, the lane-per-waveform kernels that leave most of a CU idle) go to side streams and run beside each other.  The program's stream
        # waits for all of them before the program is launched.
        plan = self._stage_plan() if self.concurrent_stages and len(self._stages) > 1 else None
        if plan is not None:
            side = getattr(lane, "side_streams", None)
            if side is None:
                side = lane.side_streams = [Stream() for _ in range(plan["n_side"])]
                lane.stage_done = [Event() for _ in self._stages]
                lane.pass_start = Event()
            lane.pass_start.record(stream)  # (behind the fits, and behind everything the previous pass of this lane left on its stream)
            for s in side:
                s.wait_event(lane.pass_start)
        for j, st in enumerate(self._stages):
            sb = dict(bufs)
            sb.update(st["dev"])
            for io_name, key in st["alias"].items():
                sb[io_name] = bufs[key]
            held = lane.stage_bufs[j]
            for out_name, key, length in st["outs"]:
                buf = held.get(key)
                if buf is None or buf.shape[0] < m:
                    buf = held[key] = DeviceArray((m,) if length is None else (m, length), st.get("out_dtypes", {}).get(key, self.loop_dtype))
                sb[out_name] = bufs[key] = buf
            if plan is None:
                lane.stage_chains[j].execute(sb, m, stream)
                continue
            k = plan["stream_of"][j]
            s = stream if k < 0 else lane.side_streams[k]
            for i in plan["deps"][j]:
                if plan["stream_of"][i] != k:
                    s.wait_event(lane.stage_done[i])
            lane.stage_chains[j].execute(sb, m, s)
            lane.stage_done[j].record(s)
        if plan is not None:
            for j in plan["sinks"]:
                if plan["stream_of"][j] >= 0:
                    stream.wait_event(lane.stage_done[j])
        for io_name, key in self._ext_alias.items():
            bufs[io_name] = bufs[key]

    #: True (DSPEED_HIP_CONCURRENT_STAGES=1): stages that do not read each other's results run on streams of their own, beside each other.  Off
    #: by default: measured on the Ge recipe it changes nothing (22.80 against 22.76 ms per 131 072 rows) -- the kernels that could overlap
    #: each fill the CUs' LDS by themselves (the interpreter 4 x 35 kB, the lane-per-waveform kernels 4 x 40 kB, the float16 FIR 84 kB), so
    #: the hardware runs them one after the other whatever stream they are on
    concurrent_stages = os.environ.get("DSPEED_HIP_CONCURRENT_STAGES", "0") == "1"

    def _stage_plan(self) -> dict:
        """Which earlier stages every stage waits for (it reads a buffer they write: the ``alias`` of its bindings against their ``outs``), and a
        stream for each: a stage continues the stream of the last stage it waits for when nothing else was put behind that one, else it takes
        the program's stream (-1) or the next side stream.  ``sinks``: stages nothing later waits for on their own stream."""
        plan = getattr(self, "_stage_plan_cache", None)
        if plan is not None:
            return plan
        n = len(self._stages)
        made = [{key for _o, key, _l in st["outs"]} for st in self._stages]
        deps = [sorted(i for i in range(j) if made[i] & set(self._stages[j]["alias"].values())) for j in range(n)]
        stream_of, tail_of, n_side = [0] * n, {}, 0   # tail_of: stream -> last stage put on it
        for j in range(n):
            k = None
            for i in reversed(deps[j]):
                if tail_of.get(stream_of[i]) == i:
                    k = stream_of[i]
                    break
            if k is None:
                if -1 not in tail_of:
                    k = -1
                else:
                    free = [q for q in range(n_side) if q not in tail_of]
                    k = free[0] if free else (n_side if n_side < 3 else min(range(n_side), key=lambda q: tail_of[q]))
                    n_side = max(n_side, k + 1)
            stream_of[j] = k
            tail_of[k] = j
        sinks = sorted(set(tail_of.values()))
        plan = self._stage_plan_cache = {"deps": deps, "stream_of": stream_of, "n_side": n_side, "sinks": sinks}
        return plan

    def __call__(self, tb_in, tb_out, begin: int = 0, end: int | None = None):
        """``proc_chain(tb_in, tb_out)`` of the reference (processing_chain.py:675-716).  LGDO tables (or stand-ins with their protocol,
        dspeed_amd/lgdo_io.py) are taken as they are: the input's columns are read through their ``.nda`` / ``values, dt, t0``, the
        results are written into the output table's columns."""
        from . import lgdo_io

        lgdo_out = tb_out if lgdo_io.is_lgdo_table(tb_out) else None
        if lgdo_io.is_lgdo_table(tb_in):
            tb_in = lgdo_io.table_columns(tb_in, set(v.source.split(".")[0] for v in self._in_vars.values()) | set(self._copy_pars))
        if lgdo_out is not None:
            n = len(_column(tb_in, next(iter(self._in_vars.values())).source)) if self._in_vars else self._buffer_len
            tb_out = {name[4:] if name.startswith("out:") else name: np.empty((n,) if length is None else (n, length), dtype=getattr(var, "dtype", None) or self.loop_dtype)
                      for name, (var, length) in self._out_vars.items()}
        self.link(tb_in, tb_out)
        self.execute(begin, end)
        if lgdo_out is not None:
            for c in self._copy_pars:
                if c in tb_in:
                    tb_out[c] = _column(tb_in, c)
            # only the rows this call computed go back into the table, at their own positions
            stop = self._buffer_len if end is None else min(int(end), self._buffer_len)
            lgdo_io.write_back(lgdo_out, {k: np.asarray(v)[begin:stop] for k, v in tb_out.items()}, begin)
            return lgdo_out
        return tb_out


class GroupedProcessingChain(ProcessingChain):
    """A recipe in which an INTEGER parameter of a processor is a per-event column -- ``trap_filter(wf, rise_col, flat_col, out)`` --, which
    the reference serves by broadcasting the column into the gufunc's "()" slot (processing_chain.py:1702-1745).  Device programs hold
    such parameters as constants (they size loops and the LDS layout), so a pass groups the rows by the values of those columns, runs each
    group through a chain built for its values (built once per distinct combination and kept) and puts the results back at the rows'
    places -- the scheme of the single-processor entry points (``gufunc.py``).  Results are those of the reference row by row; a DSPFatal
    names the first row, in table order, that met one.  Everything else -- bindings, outputs, attributes -- is the chain of the first
    combination's (``proto``)."""

    def __init__(self, proto: ProcessingChain, build, columns):
        self.__dict__.update(proto.__dict__)
        self._proto, self._build_group, self.group_columns, self._group_chains = proto, build, list(columns), {}

    def kernels(self) -> list:
        return self._proto.kernels()

    def kernel_notes(self) -> list:
        return self._proto.kernel_notes()

    @staticmethod
    def _take(col, idx):
        if isinstance(col, WaveformInput):
            t0 = col.t0 if isinstance(col.t0, float) else np.asarray(col.t0)[idx]
            return WaveformInput(GroupedProcessingChain._take(col.values, idx), col.dt, t0)
        if isinstance(col, DeviceArray):
            raise NotImplementedError("per-event integer parameters of processors: the rows are grouped by value on the host -- link host arrays")
        return np.ascontiguousarray(np.asarray(col)[idx])

    def execute(self, start: int = 0, stop: int | None = None, wait: bool = True) -> None:
        if stop is None:
            stop = self._buffer_len
        if stop <= start:
            return
        if not wait:
            raise ValueError("execute(wait=False) is for chains over device-resident columns; this one groups host rows by value")
        keys = np.stack([np.asarray(_column(self._tb_in, c))[start:stop].astype(np.int64) for c in self.group_columns], axis=1)
        uniq, first, inverse = np.unique(keys, axis=0, return_index=True, return_inverse=True)
        inverse = np.asarray(inverse).reshape(-1)
        failures = []
        for g in np.argsort(first):  # (groups in the order their first rows stand in the table)
            idx = np.flatnonzero(inverse == g) + start
            combo = tuple(int(v) for v in uniq[g])
            part = {name: self._take(col, idx) for name, col in self._tb_in.items()}
            chain = self._group_chains.get(combo)
            try:
                if chain is None:
                    chain = self._group_chains[combo] = self._build_group(dict(zip(self.group_columns, combo)), part)
                    chain.device = self.device
                out = {name: np.empty((len(idx), *np.shape(col)[1:]), dtype=np.asarray(col).dtype) for name, col in self._tb_out.items()
                       if name not in self._copy_pars}
                chain.link(part, out)
                chain.execute(0, len(idx))
            except DSPFatal as e:  # (a constant-only condition of this group's values, or a row of it: the reference meets it at the group's first such row)
                row = idx[e.wf_range.start] if isinstance(e.wf_range, range) and len(e.wf_range) else idx[0]
                e.wf_range = range(int(row), int(row) + 1)
                failures.append((int(row), e))
                continue
            for name, col in out.items():
                self._tb_out[name][idx] = col
            for k in self._timing:
                self._timing[k] += chain.get_timing()[k]
        if failures:
            raise min(failures, key=lambda f: f[0])[1]

    def __call__(self, tb_in, tb_out, begin: int = 0, end: int | None = None):
        from . import lgdo_io

        if lgdo_io.is_lgdo_table(tb_in):  # (the grouping columns are read on the host: they are part of what the table must give)
            tb_in = lgdo_io.table_columns(tb_in, set(v.source.split(".")[0] for v in self._in_vars.values()) | set(self._copy_pars) | set(self.group_columns))
        return super().__call__(tb_in, tb_out, begin, end)


def _column(tb, name):
    if name not in tb and name.endswith(".t0"):  # the per-row t0 of a WaveformInput
        return tb[name[:-3]].t0
    col = tb[name]
    return col.values if isinstance(col, WaveformInput) else col


# ----------------------------------------------------------------------------------------------------------------
# recipe parsing
# ----------------------------------------------------------------------------------------------------------------


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


def build_processing_chain(processors, tb_in=None, db_dict=None, outputs=None, block_width: int = 16, device: int | None = None):
    """``_build_chain`` -- or, where a processor's integer parameter is a column of the input table, a chain per value of that column
    (GroupedProcessingChain); documented at ``_build_chain``."""
    columns, values = [], {}
    while True:
        try:
            chain, mask, tb_out = _build_chain(processors, tb_in, db_dict, outputs, block_width, device, values)
            break
        except _PerEventInteger as e:
            col = _column(tb_in, e.column)
            if isinstance(col, DeviceArray):
                raise NotImplementedError(f"'{e.column}' is an integer parameter of a processor, given per event: the rows are grouped by its value "
                                          "on the host -- give the table's columns as host arrays") from None
            col = np.asarray(col)
            if len(col) == 0:
                raise ProcessingChainError(f"'{e.column}' is an integer parameter of a processor and the table has no rows to take its values from") from None
            columns.append(e.column)
            values[e.column] = int(col[0])
    if not columns:
        return chain, mask, tb_out
    recipe = _load(processors)

    def build(group_values, part):
        return _build_chain(recipe, part, db_dict, outputs, block_width, device, group_values)[0]

    grouped = GroupedProcessingChain(chain, build, columns)
    grouped.link(tb_in, tb_out)
    return grouped, mask, tb_out


def _build_chain(processors, tb_in, db_dict, outputs, block_width, device, group_values):
    """Translate a dspeed recipe into a device chain.

    Returns ``(proc_chain, field_mask, tb_out)`` like the reference (processing_chain.py:2363-2369): ``tb_in`` is a
    mapping ``name -> ndarray | DeviceArray | WaveformInput``; ``tb_out`` a dict of freshly allocated NumPy arrays for
    the requested outputs; ``field_mask`` the input columns actually used.  ``block_width`` is accepted for
    signature compatibility: the device processes the whole buffer in one launch.  ``device``: the GPU this chain lives on (its handle,
    streams and buffers are created there, and ``execute`` makes it the calling thread's current device); default: whatever device is
    current when the chain first runs.
    """
    del block_width
    from . import lgdo_io

    if tb_in is not None and lgdo_io.is_lgdo_table(tb_in):  # an lgdo.Table (or a stand-in with its protocol): its columns as arrays
        tb_in = lgdo_io.table_columns(tb_in)
    recipe = _load(processors)
    if outputs is None:
        if "outputs" not in recipe:
            raise ValueError("outputs not provided")
        outputs = recipe["outputs"]
    nodes = dict(recipe["processors"]) if "processors" in recipe else dict(recipe)
    nodes.pop("outputs", None)

    book = Recipe(nodes, db_dict)
    order, leafs, out_pars, copy_pars = book.plan(outputs)

    b = _Builder(tb_in, db_dict)
    b.group_values = dict(group_values)
    for leaf in leafs:
        if tb_in is None or leaf not in tb_in:
            raise ProcessingChainError(f"'{leaf}' not found in input table or recipe")
        b.input_var(leaf)

    proc_strings = []
    for entry in order:
        try:
            _add_step(b, entry.key, entry, list(entry.targets), proc_strings)
        except (ProcessingChainError, NotImplementedError, DSPFatal, _PerEventInteger):
            raise
        except Exception as e:
            raise ProcessingChainError("Exception raised while attempting to add processor:\n" + json.dumps(entry.as_dict(), indent=2, default=str)) from e
    n_rows = 0
    if tb_in:
        n_rows = len(_column(tb_in, next(iter(tb_in))))
    chain, tb_out = _compile(b, out_pars, n_rows, proc_strings)
    for c in copy_pars:
        if tb_in is not None and c in tb_in:
            tb_out[c] = _column(tb_in, c)
    chain._copy_pars = list(copy_pars)
    # what the LGDO output columns carry besides their values (reference :1990-2014 units, :2725-2740 lh5_attrs / description)
    chain.output_attrs = {}
    for o in out_pars:
        v, entry, a = b.vars.get(o), book.defined_by.get(o), {}
        unit = getattr(v, "unit", None)
        if isinstance(unit, str):
            a["units"] = unit
        if entry is not None:
            a.update(entry.get("lh5_attrs") or {})
            if entry.get("description") is not None:
                a["description"] = entry.get("description")
        chain.output_attrs[o] = a
    chain.device = None if device is None else int(device)
    chain.link(tb_in, tb_out)
    return chain, leafs + copy_pars, tb_out


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
                build(anc, [base], f"{base.name} -> HBM")
        # --- the filter itself; numpy.amax goes along when it is the only reader
        users = [x for x in steps if x is not st and any(base_of(a) is out for a, r in zip(x[1], _roles(x[0])) if r not in "WS")]
        if (len(users) == 1 and users[0][0] == "amax" and users[0][1][0] is out and out.name not in out_names and isinstance(users[0][1][2], Var)
                and mode == "v" and out.length <= 320):  # (what the amax form of the kernel takes; else the filtered waveform is kept)
            build(pre + [st, users[0]], [users[0][1][2]], f"{fn} {key} + amax")
            users[0][1][2].kind = "scalar"
            gone = [st, users[0]]
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
            p.add_op(_lib.OP_CONVOLVE, dst=dst.slot, src=src.slot, io=taps.io, ip=(char_of(args[2]), has_nan, 0, int(taps.length)))
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
    chain = ProcessingChain(p, in_bind, out_bind, consts, n_rows, proc_strings, ft, aux_desc, stages=stages, ext_alias=ext_alias, tail=tail)
    chain.vector_lens = vector_lens  # variable-length outputs -> the input column that holds their per-event lengths
    return chain, tb_out


_SCALAR_OPS = (_lib.OP_SCALAR_AFFINE, _lib.OP_SCALAR_DIV, _lib.OP_SCALAR_CONVERT, _lib.OP_SCALAR_FUNC, _lib.OP_STORE_SCALAR)
#: a tail is cut off when it has at least this many ops (a launch and a column per handed-over register have to pay for themselves)
SCALAR_TAIL_MIN_OPS = 8


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



def shard_rows(n_rows: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous split of the event axis: rank r gets rows [r*N/G, (r+1)*N/G).  Events are independent,
    so a multi-GPU run is one chain per rank over its slice and a concatenation of the outputs -- no collective."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return (n_rows * rank) // world_size, (n_rows * (rank + 1)) // world_size
