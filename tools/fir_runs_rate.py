#!/usr/bin/env python3
"""Rate of the run-length FIR (dsp_fir_runs.hip) on a device-resident batch of float32 rows, in its three forms: the filtered waveform
kept, kept with the per-event values (min_max + a threshold walk from the maximum: the t0 estimate of the Ge recipes), and the per-event
values alone (nothing but the input rows touches HBM).  The matrix-core FIR on the same rows beside it (DSPEED_HIP_NO_FIR_RUNS=1).
python tools/fir_runs_rate.py [rows] [samples] [steps]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util  # noqa: E402
from dspeed_amd import _lib  # noqa: E402
from dspeed_amd.chain import Chain, Program, Scalar  # noqa: E402
from dspeed_amd.device import DeviceArray, Event, Stream, sync  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
taps = golden_util.recipe_kernel("t0")
m = len(taps)


def program(keep, red):
    p = Program()
    p.slots = [n, n]
    wf = p.add_io("wf", _lib.IO_WF_IN, np.float32, n, 0, n)
    tp = p.add_io("taps", _lib.IO_TAPS, np.float32, 144, 0, 0)
    p.add_op(_lib.OP_LOAD, dst=0, io=wf)
    p.add_op(_lib.OP_CONVOLVE, dst=1, src=0, io=tp, ip=(ord("s"), 0, 1, m))
    outs = []
    if keep:
        p.add_op(_lib.OP_STORE, src=1, io=p.add_io("filtered", _lib.IO_WF_OUT, np.float32, n, 0, n))
    if red:
        thr = p.add_io("thr", _lib.IO_SCALAR_IN, np.float32)
        p.n_sregs = 5
        p.add_op(_lib.OP_MIN_MAX, dst=0, src=1)
        p.add_op(_lib.OP_TIME_POINT_THRESH, dst=4, src=1, sp=(Scalar.input(thr), Scalar.reg(1), Scalar.const(0.0)))
        for r in range(5):
            outs.append(f"o{r}")
            p.add_op(_lib.OP_STORE_SCALAR, io=p.add_io(f"o{r}", _lib.IO_SCALAR_OUT, np.float32), ip=(r,))
    return p, outs


st = Stream()
wf = DeviceArray((rows, n), np.float32)
bl, tp = DeviceArray((rows,), np.float32), DeviceArray((rows,), np.float32)
_lib.check(_lib.lib().dsp_synth_waveforms(wf.ptr, _lib.F32, rows, n, n, bl.ptr, tp.ptr, 0xD5BEED, 0, 1716.28, 5.0, 625 + 0.8 * 188,
                                          -30.0, 30.0, 500.0, 15000.0, st.ptr), what="synth")
sync()
thr = DeviceArray.from_numpy(np.full(rows, 20.0, np.float32))
tapbuf = DeviceArray.from_numpy(np.concatenate([taps, np.zeros(144 - m, np.float32)]))
filtered = DeviceArray((rows, n), np.float32)
for name, keep, red in (("kept", 1, 0), ("kept + per-event values", 1, 1), ("per-event values only", 0, 1)):
    prog, outs = program(keep, red)
    ch = Chain(prog, name, np.float32)
    bufs = {"wf": wf, "taps": tapbuf, "thr": thr, "filtered": filtered}
    bufs.update({o: DeviceArray((rows,), np.float32) for o in outs})
    bufs = {k: v for k, v in bufs.items() if k in [io[0] for io in prog.io]}
    for _ in range(2):
        ch.execute(bufs, rows, st)
    e0, e1 = Event(), Event()
    e0.record(st)
    for _ in range(steps):
        ch.execute(bufs, rows, st)
    e1.record(st)
    sync()
    ch.check(st)
    dt = e0.elapsed_ms(e1) * 1e-3 / steps
    byts = rows * n * 4 * (1 + keep)
    print(json.dumps({"form": name, "kernel": ch.kernel_name, "rows": rows, "samples": n, "taps": m, "ms": round(dt * 1e3, 4), "waveforms_per_s": round(rows / dt),
                      "algorithmic_GBps": round(byts / dt / 1e9, 1), "frac_of_8TBps": round(byts / dt / 8e12, 3), **ch.geometry(rows)}))
