#!/usr/bin/env python3
"""Rate of a stored FIR (the Ge recipes' t0 filter: 133 taps, 'same', 8192 outputs) on a device-resident batch:
python tools/fir_store_rate.py [rows] [fused 0|1] [taps] [mode]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dspeed_amd import _lib  # noqa: E402
from dspeed_amd.device import DeviceArray, Event, Stream, sync  # noqa: E402
from dspeed_amd.processing_chain import build_processing_chain  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
fused = int(sys.argv[2]) if len(sys.argv) > 2 else 1
m = int(sys.argv[3]) if len(sys.argv) > 3 else 133
mode = sys.argv[4] if len(sys.argv) > 4 else "s"
n = 8192
p = {"v": n - m + 1, "s": n, "f": n + m - 1}[mode]
M = "dspeed.processors"
rec = {"outputs": ["wf_f"], "processors": {
    "wf_bl": f"{M}.bl_subtract(waveform, baseline, wf_bl)",
    "k": {"function": "t0_filter", "module": M, "args": [str(m // 3), str(m - m // 3), f"k({m}, 'f')"]},
    "wf_f": {"function": "convolve_wf", "module": M, "args": ["wf_bl", "k", f"'{mode}'", f"wf_f({p}, 'f')"]}}}
st = Stream()
wf = DeviceArray((rows, n), np.float32)
bl, tp = DeviceArray((rows,), np.float32), DeviceArray((rows,), np.float32)
_lib.check(_lib.lib().dsp_synth_waveforms(wf.ptr, _lib.F32, rows, n, n, bl.ptr, tp.ptr, 0xD5BEED, 0, 1716.28, 5.0, 625 + 0.8 * 188,
                                          9000.0, 11000.0, 500.0, 15000.0, st.ptr), what="synth")
sync()
tb = {"waveform": wf, "baseline": bl}
chain, _, _ = build_processing_chain(rec, tb)
chain.link(tb, {"wf_f": DeviceArray((rows, p), np.float32)})
chain._ensure()
chain._chain.set_fused(fused)
for _ in range(2):
    chain.execute()
steps = 3
e0, e1 = Event(), Event()
e0.record(chain._stream)
for _ in range(steps):
    chain.execute()
e1.record(chain._stream)
sync()
dt = e0.elapsed_ms(e1) * 1e-3 / steps
flop = 2 * m * p
print(json.dumps({"kernel": chain._chain.kernel_name, "rows": rows, "taps": m, "mode": mode, "ms": dt * 1e3, "waveforms_per_s": rows / dt,
                  "algorithmic_TFLOPs": rows * flop / dt / 1e12, "GBps": rows * (n + p) * 4 / dt / 1e9}))
