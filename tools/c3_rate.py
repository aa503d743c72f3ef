#!/usr/bin/env python3
"""C3 (BASELINE.json configs[2]) rate on a device-resident synthetic batch: python tools/c3_rate.py [rows] [fused 0|1]"""
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

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000
fused = int(sys.argv[2]) if len(sys.argv) > 2 else 1
st = Stream()
wf = DeviceArray((rows, 8192), np.float32)
bl, tp = DeviceArray((rows,), np.float32), DeviceArray((rows,), np.float32)
_lib.check(_lib.lib().dsp_synth_waveforms(wf.ptr, _lib.F32, rows, 8192, 8192, bl.ptr, tp.ptr, 0xD5BEED, 0, 1716.28, 5.0, 625 + 0.8 * 188,
                                          9000.0, 11000.0, 500.0, 15000.0, st.ptr), what="synth")
sync()
tb = {"waveform": wf, "baseline": bl}
chain, _, _ = build_processing_chain(recipes.C3, tb)
chain.link(tb, {"cuspEmax": DeviceArray((rows,), np.float32), "zacEmax": DeviceArray((rows,), np.float32)})
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
flop = 2 * 5792 * 301 * 2  # direct-form flops of the two kernels
name = chain._chain.kernel_name
rec = {"config": "C3", "kernel": name, "rows": rows, "ms": dt * 1e3, "waveforms_per_s": rows / dt, "algorithmic_TFLOPs": rows * flop / dt / 1e12}
if "f16" in name:
    # three float16 products per multiply-add (two-way split operands), over the 64 x 320 x 6144 padded tile of every row block and kernel
    issued = rows * 3 * 2 * 6144 * 320 * 2 / dt
    rec.update({"bound": "mfma-f16", "issued_TFLOPs_f16_mfma": issued / 1e12, "frac_f16_mfma_peak": issued / 2.5e15,
                "algorithmic_over_fp32_mfma_peak": rows * flop / dt / 157.3e12,
                "row_bytes_read_GBps": rows * 6092 * 4 * 2 / dt / 1e9})  # (the scale pass and the product, both kernels sharing the rows in L2 at best)
elif fused:
    rec.update({"bound": "mfma-f32", "frac_fp32_peak": rows * flop / dt / 157.3e12, "issued_TFLOPs_mfma": rows * 2 * 6112 * 320 * 2 / dt / 1e12})
print(json.dumps(rec))
