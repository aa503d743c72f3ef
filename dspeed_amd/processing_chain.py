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

import json
import queue
import threading
import time
from collections.abc import MutableMapping
import os
from copy import deepcopy
from types import SimpleNamespace

import numpy as np

from . import _lib
from .chain import Chain, Program
from .compiler import _PerEventInteger, _add_step, _compile
from .device import DeviceArray, Event, HostPin, PinnedArray, Stream, dtype_code, set_device
from .errors import DSPFatal, ProcessingChainError
from .language import Grid, Quantity, SExpr, Var, WaveformInput, _Builder, _column, _int_loop_const, _int_loop_of, log  # noqa: F401  (re-exported)
from .recipe import Recipe


_COPY_POOL = None  # host threads that move rows between NumPy columns and staging buffers
_COPY_POOL_LOCK = threading.Lock()


class ProcessingChain:
    """Runs a translated recipe over a buffer of rows.  ``execute(start, stop)`` has the meaning of the reference's
    (processing_chain.py:665-673); ``__call__(tb_in, tb_out)`` relinks I/O like :675-716."""

    def __init__(self, program: Program, inputs: dict, outputs: dict, consts: dict, buffer_len: int, proc_strings: list[str],
                 loop_dtype=np.float32, aux=(), stages=(), ext_alias=None, tail=None, walks=None):
        self._program = program
        self._tail = tail         # the program's all-scalar tail as a program of its own, run behind it with a row per lane (_split_scalar_tail)
        self._walks = walks       # the program's threshold walks as a program of their own, run behind it off the rows in HBM (_split_walks)
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
        if self._lanes and self._lanes[0].walks is not None:
            rep.append(("threshold walks behind the program", self._lanes[0].walks.kernel_name))
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
                                           tail_bufs={}, walks=self._walks_chain(0))]
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
            reads = {nm.split("[")[0] for nm in st["in_vars"]}
            for j in range(i + 1, len(self._stages)):
                # ... or one that filters a slice of the rows the pole-zero kernel reads, minus the same baseline (the cusp filter on
                # waveform[0:6092] - baseline): the kernel sees those samples anyway
                same_input = reads & {nm.split("[")[0] for nm in self._stages[j]["in_vars"]}
                if made & set(self._stages[j]["alias"].values()) or same_input:
                    stage_chains[i].share_row_scales(stage_chains[j])

    def _walks_chain(self, lane_no: int):
        if self._walks is None:
            return None
        ch = Chain(self._walks["program"], f"processing_chain threshold walks (lane {lane_no})", self.loop_dtype)
        ch.set_async_check(True)
        return ch

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
        names = (self._tail["handover"] if self._tail is not None else []) + (self._walks["handover"] if self._walks is not None else [])
        for name in names:
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
                                               aux_bufs={}, tail=self._tail_chain(len(self._lanes)), tail_bufs={}, walks=self._walks_chain(len(self._lanes))))
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
            if lane.walks is not None:
                lane.walks.check(lane.stream, row_offset=a)
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
            if lane.walks is not None:
                lane.walks.execute(bufs, n, lane.stream)
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
                if lane.walks is not None:
                    lane.walks.check(lane.stream, row_offset=a)
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
                if lane.walks is not None:
                    lane.walks.execute(bufs, m, s_c)
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
                    for ch in (*ln.stage_chains, ln.chain, *([ln.walks] if ln.walks is not None else [])):  # their error words belong to the abandoned pieces
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
        # The fits beside the first stages: one waveform per lane is eight wavefronts a CU whose recurrences wait out every instruction, the
        # kernel that writes the pole-zero rows is bound by HBM and uses no LDS -- the two share the CUs.  The fits go to a stream of their
        # own behind everything the lane's stream holds so far; the first stage that reads a fit's column (and the program) waits for them.
        beside = self.fits_beside_stages and bool(self._aux) and bool(self._stages)
        main_stream = stream
        if beside:
            if getattr(lane, "fit_stream", None) is None:
                lane.fit_stream, lane.fit_start, lane.fit_done = Stream(), Event(), Event()
            lane.fit_start.record(main_stream)
            lane.fit_stream.wait_event(lane.fit_start)
            stream = lane.fit_stream
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
        fits_pending = False
        if beside:
            lane.fit_done.record(stream)
            stream, fits_pending = main_stream, True
        # The stages, in launch order (a stage may read what an earlier one wrote) -- but not all on one stream: what a stage reads says which
        # earlier stages it waits for, and stages that do not wait for each other (the cusp filter beside the t0 chain, the reductions off the raw
        # rows, the lane-per-waveform kernels that leave most of a CU idle) go to side streams and run beside each other.  The program's stream
        # waits for all of them before the program is launched.
        plan = self._stage_plan() if self.concurrent_stages and len(self._stages) > 1 else None
        if plan is not None:
            if fits_pending:  # (stages on side streams of their own: they start behind the fits, as before)
                stream.wait_event(lane.fit_done)
                fits_pending = False
            side = getattr(lane, "side_streams", None)
            if side is None:
                side = lane.side_streams = [Stream() for _ in range(plan["n_side"])]
                lane.stage_done = [Event() for _ in self._stages]
                lane.pass_start = Event()
            lane.pass_start.record(stream)  # (behind the fits, and behind everything the previous pass of this lane left on its stream)
            for s in side:
                s.wait_event(lane.pass_start)
        for j, st in enumerate(self._stages):
            if fits_pending and any(str(key).startswith("aux:") for key in st["alias"].values()):
                stream.wait_event(lane.fit_done)
                fits_pending = False
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
        if fits_pending:
            stream.wait_event(lane.fit_done)
        for io_name, key in self._ext_alias.items():
            bufs[io_name] = bufs[key]

    #: True (default; DSPEED_HIP_FITS_BESIDE=0 switches it off): the fits on the rows run on a stream of their own beside the stages that do not read
    #: their results -- in the Ge recipe the kernel that writes the pole-zero rows
    fits_beside_stages = os.environ.get("DSPEED_HIP_FITS_BESIDE", "1") != "0"

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
                if isinstance(self._tb_out[name], DeviceArray):
                    raise NotImplementedError("per-event integer parameters of processors: the groups' results are put back on the host -- link host arrays as outputs")
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



def shard_rows(n_rows: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous split of the event axis: rank r gets rows [r*N/G, (r+1)*N/G).  Events are independent,
    so a multi-GPU run is one chain per rank over its slice and a concatenation of the outputs -- no collective."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return (n_rows * rank) // world_size, (n_rows * (rank + 1)) // world_size

