#!/usr/bin/env python3
"""C5 (BASELINE.json configs[4]) rate on a device-resident synthetic batch: python tools/c5_rate.py [rows] [fused 0|1]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import recipes  # noqa: E402
from dspeed_amd import _lib  # noqa: E402
from dspeed_amd.device import DeviceArray, Event, Stream, sync  # noqa: E402
from dspeed_amd.processing_chain import build_processing_chain  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
fused = int(sys.argv[2]) if len(sys.argv) > 2 else 1
st = Stream()
wf = DeviceArray((rows, 8192), np.int16)
bl, tp = DeviceArray((rows,), np.float32), DeviceArray((rows,), np.float32)
_lib.check(_lib.lib().dsp_synth_waveforms(wf.ptr, _lib.I16, rows, 8192, 8192, bl.ptr, tp.ptr, 0xD5BEED, 0, 1716.28, 5.0, 625 + 0.8 * 188,
                                          -3000.0, 3000.0, 500.0, 15000.0, st.ptr), what="synth")
thr = DeviceArray.from_numpy(np.full(rows, 20.0, dtype=np.float32))
sync()
outs = {k: DeviceArray((rows,), np.float32) for k in ("tp_0", "tp_min", "tp_max", "wf_min", "wf_max")}
outs["dwt_haar"] = DeviceArray((rows, 256), np.float32)
tb = {"waveform": wf, "thr": thr}
chain, _, _ = build_processing_chain(recipes.C5, tb)
chain.link(tb, outs)
chain._ensure()
chain._chain.set_fused(fused)
for _ in range(2):
    chain.execute()
steps = 5
e0, e1 = Event(), Event()
e0.record(chain._stream)
for _ in range(steps):
    chain.execute()
e1.record(chain._stream)
sync()
dt = e0.elapsed_ms(e1) * 1e-3 / steps
bytes_wf = 8192 * 2 + 4 + 5 * 4 + 256 * 4
print(json.dumps({"config": "C5", "kernel": chain._chain.kernel_name, "rows": rows, "ms": dt * 1e3, "waveforms_per_s": rows / dt,
                  "bytes_per_waveform": bytes_wf, "achieved_GBps": rows * bytes_wf / dt / 1e9, "frac_hbm": rows * bytes_wf / dt / 8e12}))
