"""Python face of a fused device chain (``dsp_chain_*`` in include/dspeed_hip.h).

``Program`` collects ops / I/O bindings / slots the way the C structs want them; ``Chain`` owns the
device handle and runs batches.  The JSON-level builder lives in ``processing_chain.py``; the gufunc-style
single processors in ``processors/`` go through the C entry points instead.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .device import DeviceArray, Stream, dtype_code


@dataclass
class Scalar:
    """Scalar operand of an op: a constant, a per-waveform input column (binding index) or a scalar register."""
    kind: int
    index: int = 0
    value: float = 0.0

    @staticmethod
    def const(v) -> "Scalar":
        return Scalar(_lib.ARG_CONST, 0, float(v))

    @staticmethod
    def input(io_index: int) -> "Scalar":
        return Scalar(_lib.ARG_INPUT, io_index, 0.0)

    @staticmethod
    def reg(r: int) -> "Scalar":
        return Scalar(_lib.ARG_REG, r, 0.0)


@dataclass
class Program:
    ops: list = field(default_factory=list)
    io: list = field(default_factory=list)       # (name, kind, dtype_code, len, offset, row_stride)
    slots: list = field(default_factory=list)    # lengths
    n_sregs: int = 0

    def add_slot(self, length: int) -> int:
        self.slots.append(int(length))
        return len(self.slots) - 1

    def add_sregs(self, n: int = 1) -> int:
        first = self.n_sregs
        self.n_sregs += n
        return first

    def add_io(self, name: str, kind: int, dtype, length: int = 1, offset: int = 0, row_stride: int | None = None) -> int:
        code = dtype if isinstance(dtype, int) else dtype_code(dtype)
        if row_stride is None:
            row_stride = length + offset if kind in (_lib.IO_WF_IN, _lib.IO_WF_OUT) else (0 if kind == _lib.IO_TAPS else 1)
        self.io.append((name, kind, code, int(length), int(offset), int(row_stride)))
        return len(self.io) - 1

    def add_op(self, opcode: int, dst: int = 0, src: int = 0, io: int = 0, ip=(), sp=()) -> int:
        self.ops.append((opcode, int(dst), int(src), int(io), tuple(int(v) for v in ip), tuple(sp)))
        return len(self.ops) - 1


def _c_program(program: Program):
    """the program as the C structs of dsp_chain_create / dsp_chain_plan"""
    n_ops, n_io = len(program.ops), len(program.io)
    ops = (_lib.Op * max(n_ops, 1))()
    for i, (opcode, dst, src, io, ip, sp) in enumerate(program.ops):
        o = ops[i]
        o.opcode, o.dst, o.src, o.io = opcode, dst, src, io
        for k, v in enumerate(ip):
            o.ip[k] = v
        for k, s in enumerate(sp):
            o.sp[k].kind, o.sp[k].index, o.sp[k].value = s.kind, s.index, s.value
    ios = (_lib.IoDesc * max(n_io, 1))()
    for i, (_, kind, code, length, offset, stride) in enumerate(program.io):
        d = ios[i]
        d.kind, d.dtype, d.len, d.offset, d.row_stride = kind, code, length, offset, stride
    slots = (C.c_int32 * max(len(program.slots), 1))(*program.slots)
    return ops, n_ops, ios, n_io, slots, len(program.slots)


def plan(program: Program, compute_dtype=np.float32, name: str = "chain") -> dict:
    """What ``Chain(program)`` would be given -- kernel, note, LDS layout -- without a device (``dsp_chain_plan``): the same validation and the
    same errors as creating the chain, on a machine with or without a GPU."""
    info = _lib.PlanInfo()
    rc = _lib.lib().dsp_chain_plan(*_c_program(program), program.n_sregs, dtype_code(np.dtype(compute_dtype)), C.byref(info))
    _lib.check(rc, what=name)
    n = info.n_slots
    out = {k: getattr(info, k) for k in ("lds_bytes_per_wave", "waves_per_block", "team", "n_device_ops", "lds_elems_per_wave", "sreg_off", "scratch_off")}
    out.update(kernel=info.kernel.decode(), note=info.note.decode(),
               slots=[{k: getattr(info, "slot_" + k)[s] for k in ("base", "elems", "first_op", "last_op", "off", "pitch", "chunk")} for s in range(n)])
    return out


class Chain:
    """A compiled chain bound to the current device."""

    def __init__(self, program: Program, name: str = "chain", compute_dtype=np.float32):
        self.program = program
        self.name = name
        self.compute_dtype = np.dtype(compute_dtype)
        L = _lib.lib()
        handle = C.c_void_p()
        rc = L.dsp_chain_create(*_c_program(program), program.n_sregs, dtype_code(self.compute_dtype), C.byref(handle))
        _lib.check(rc, what=name)
        self._h = handle
        self.io_names = [io[0] for io in program.io]

    def execute(self, buffers: dict, n_wf: int, stream: Stream | None = None) -> None:
        """Enqueue one pass over n_wf rows.  ``buffers``: binding name -> DeviceArray (or raw device pointer)."""
        ptrs = (C.c_void_p * max(len(self.io_names), 1))()
        for i, nm in enumerate(self.io_names):
            b = buffers[nm]
            ptrs[i] = b.ptr if isinstance(b, DeviceArray) else int(b)
        _lib.check(_lib.lib().dsp_chain_execute(self._h, ptrs, int(n_wf), stream.ptr if stream else None), what=self.name)

    def check(self, stream: Stream | None = None, row_offset: int = 0) -> None:
        """Wait for the stream and raise DSPFatal if a data-dependent fatal condition was met."""
        row = C.c_int64(-1)
        rc = _lib.lib().dsp_chain_check(self._h, stream.ptr if stream else None, C.byref(row))
        _lib.check(rc, row=(row.value + row_offset) if row.value >= 0 else None, what=self.name)

    def geometry(self, n_wf: int) -> dict:
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        _lib.check(_lib.lib().dsp_chain_geometry(self._h, int(n_wf), C.byref(a), C.byref(b), C.byref(c)))
        return {"lds_bytes_per_wave": a.value, "waves_per_block": b.value, "blocks": c.value}

    def set_fused(self, enable) -> bool:
        """False/0: the generic waveform VM; True/1: the default specialised energy kernel where the chain has that shape (the
        register-resident kernel for 1024/2048/4096 samples); 13: the register-resident kernel explicitly; 15: the classic
        specialised kernel (VM layout, bit-identical to the VM; float32 rows only).  Returns whether a specialised kernel is in use."""
        return bool(_lib.lib().dsp_chain_set_fused(self._h, int(enable)))

    def set_async_check(self, enable: bool = True) -> None:
        """the error word travels with every launch (a page-locked mirror): ``check`` then issues no transfer of its own -- for pipelines
        that send the next buffer to the device meanwhile"""
        _lib.check(_lib.lib().dsp_chain_set_async_check(self._h, int(bool(enable))), what=self.name)

    def profile(self, enable: bool = True) -> None:
        """Switch the in-kernel per-op timing on (zeroing the counters) or off; see ``profile_read``."""
        _lib.check(_lib.lib().dsp_chain_profile(self._h, int(bool(enable))), what=self.name)

    def profile_read(self) -> dict:
        """Per op of the device program: opcode, the slot it reads and the shader-clock cycles summed over the sampled waveforms
        (the first wavefront of every workgroup), plus that number of waveforms."""
        L = _lib.lib()
        cap = _lib.MAX_OPS + _lib.MAX_SLOTS
        opcodes, slots, cycles = (C.c_int32 * cap)(), (C.c_int32 * cap)(), (C.c_uint64 * cap)()
        n, nwf = C.c_int(), C.c_uint64()
        _lib.check(L.dsp_chain_profile_read(self._h, cap, opcodes, slots, cycles, C.byref(n), C.byref(nwf)), what=self.name)
        return {"opcodes": list(opcodes[:n.value]), "slots": list(slots[:n.value]), "cycles": list(cycles[:n.value]),
                "waveforms": int(nwf.value)}

    @property
    def kernel_name(self) -> str:
        return _lib.lib().dsp_chain_kernel_name(self._h).decode()

    @property
    def kernel_note(self) -> str:
        """"" or why this chain runs on the generic interpreter although its ops are those of a specialised kernel"""
        return _lib.lib().dsp_chain_kernel_note(self._h).decode()

    def share_row_scales(self, consumer: "Chain") -> bool:
        """This chain writes pole-zero corrected rows, ``consumer`` runs a float16 matrix-core FIR over them: let the rows' scales and flags
        travel with the rows instead of being read off them again (``dsp_chain_share_row_scales``).  False: not such a pair, nothing changed."""
        rc = _lib.lib().dsp_chain_share_row_scales(self._h, consumer._h)
        if rc < 0:
            _lib.check(rc, what=self.name)
        return rc == 1

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().dsp_chain_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def energy_chain_program(wf_len: int, tau: float, rise: int, flat: int, mode: str = "l", wf_dtype=np.float32,
                         row_stride: int | None = None, trap: str = "trap_filter") -> Program:
    """BASELINE.json config 2/4: bl_subtract -> pole_zero -> trap_filter -> fixed_time_pickoff, fused.

    Bindings: ``waveform`` (n_wf, wf_len), ``baseline`` (n_wf,), ``t_pick`` (n_wf,) -> ``trapEftp`` (n_wf,).
    The waveform is read once; nothing but the 4-byte energy is written (16 396 algorithmic bytes per waveform).
    """
    p = Program()
    s = p.add_slot(wf_len)
    r = p.add_sregs(1)
    io_wf = p.add_io("waveform", _lib.IO_WF_IN, wf_dtype, wf_len, 0, row_stride)
    io_bl = p.add_io("baseline", _lib.IO_SCALAR_IN, np.float32)
    io_tp = p.add_io("t_pick", _lib.IO_SCALAR_IN, np.float32)
    io_e = p.add_io("trapEftp", _lib.IO_SCALAR_OUT, np.float32)
    kind = {"trap_filter": _lib.OP_TRAP_FILTER, "trap_norm": _lib.OP_TRAP_NORM}[trap]
    p.add_op(_lib.OP_LOAD, dst=s, io=io_wf)
    p.add_op(_lib.OP_BL_SUBTRACT, dst=s, src=s, sp=(Scalar.input(io_bl),))
    p.add_op(_lib.OP_POLE_ZERO, dst=s, src=s, sp=(Scalar.const(tau),))
    p.add_op(_lib.OP_TRAP_PICKOFF, dst=r, src=s, io=ord(mode), ip=(rise, flat, 0, kind), sp=(Scalar.input(io_tp),))
    p.add_op(_lib.OP_STORE_SCALAR, io=io_e, ip=(r,))
    return p
