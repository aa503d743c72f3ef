#!/usr/bin/env python3
"""What one op costs on the waveform interpreter, whatever it computes: programs LOAD -> MIN_MAX -> N x SCALAR_AFFINE -> STORE_SCALAR on
8192-sample rows (one row per SIMD: a lone wavefront), N = 8 and 40, operands a register or constants -- the slope is the dispatch
(op fetch through the scalar cache, decode, ~ 150 instructions of a wavefront that issues one per ~ 12 cycles).  Measured: 45 us per op
and 65 536 rows = 64 rounds of 1024 rows = 0.7 us = ~ 1 700 cycles per op and row.  Why recipes' programs shed ops: _split_scalar_head,
_split_scalar_tail, the planner's threshold fold and merged stores.   python tools/vm_op_cost.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dspeed_amd import _lib
from dspeed_amd.chain import Chain, Program, Scalar
from dspeed_amd.device import DeviceArray, Event, Stream, sync
rows, n = 65536, 8192
st = Stream()
wf = DeviceArray.zeros((rows, n), np.float32)
res = []
for n_sc in (8, 40):
    for kind in ("reg", "const"):
        p = Program(); p.slots = [n]; p.n_sregs = 6
        io = p.add_io("wf", _lib.IO_WF_IN, np.float32, n, 0, n)
        col = p.add_io("col", _lib.IO_SCALAR_IN, np.float32)
        p.add_op(_lib.OP_LOAD, dst=0, io=io)
        p.add_op(_lib.OP_MIN_MAX, dst=0, src=0)
        for k in range(n_sc):
            a = Scalar.reg(3) if kind == "reg" else Scalar.const(2.0)
            p.add_op(_lib.OP_SCALAR_AFFINE, dst=4, sp=(a, Scalar.const(0.5), Scalar.const(1.0)))
        o = p.add_io("o", _lib.IO_SCALAR_OUT, np.float32)
        p.add_op(_lib.OP_STORE_SCALAR, io=o, ip=(4 if n_sc else 3,))
        ch = Chain(p, "t", np.float32)
        bufs = {"wf": wf, "col": DeviceArray.zeros((rows,), np.float32), "o": DeviceArray.zeros((rows,), np.float32)}
        for _ in range(2): ch.execute(bufs, rows, st)
        e0, e1 = Event(), Event(); e0.record(st)
        for _ in range(5): ch.execute(bufs, rows, st)
        e1.record(st); sync()
        res.append({"scalar_ops": n_sc, "operand": kind, "ms": e0.elapsed_ms(e1) / 5, "kernel": ch.kernel_name, **ch.geometry(rows)})
        print(json.dumps(res[-1]))
