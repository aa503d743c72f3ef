#!/usr/bin/env python3
"""Marginal cost of each stage of the C5 chain (int16 x 8192) on the waveform VM: builds the chain stage by stage."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dspeed_amd import _lib
from dspeed_amd.device import DeviceArray, Event, Stream, sync
from dspeed_amd.processing_chain import build_processing_chain
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
st = Stream()
wf = DeviceArray((rows, 8192), np.int16); bl = DeviceArray((rows,), np.float32); tp = DeviceArray((rows,), np.float32)
_lib.check(_lib.lib().dsp_synth_waveforms(wf.ptr, _lib.I16, rows, 8192, 8192, bl.ptr, tp.ptr, 1234, 0, 1716.28, 5.0, 775.0, -3000.0, 3000.0, 500.0, 15000.0, st.ptr))
thr = DeviceArray.from_numpy(np.full(rows, 20.0, np.float32)); sync()
M = "dspeed.processors"
P = {"wf_pz": f"{M}.double_pole_zero(waveform, 1716.28, 62.5, 0.02, wf_pz)",
     "wf_atrap": f"{M}.asym_trap_filter(wf_pz, 8, 4, 125, wf_atrap)",
     "tp_min, tp_max, wf_min, wf_max": {"function": "min_max", "module": M, "args": ["wf_atrap", "tp_min", "tp_max", "wf_min", "wf_max"]},
     "tp_0": f"{M}.time_point_thresh(wf_atrap, thr, tp_max, 0, tp_0)",
     "dwt_haar": {"function": "discrete_wavelet_transform", "module": M, "args": ["wf_pz", 5, "'h'", "'a'", "dwt_haar(256, 'f')"]}}
S = lambda: DeviceArray((rows,), np.float32)
stages = [(["wf_max_dummy"], None)]
def run(label, keys, outs):
    rec = {"outputs": list(outs), "processors": {k: P[k] for k in keys}}
    tb = {"waveform": wf, "thr": thr}
    chain, _, _ = build_processing_chain(rec, tb); chain.link(tb, outs)
    chain.execute(); e0, e1 = Event(), Event(); e0.record(chain._stream)
    for _ in range(3): chain.execute()
    e1.record(chain._stream); sync(); dt = e0.elapsed_ms(e1) / 3
    g = chain._chain.geometry(rows)
    print(f"{label:34s} {dt:8.2f} ms  {rows/dt/1e3:8.2f} M wf/s  lds/wave {g['lds_bytes_per_wave']:6d}  waves/block {g['waves_per_block']}  blocks {g['blocks']}", flush=True)
mm = "tp_min, tp_max, wf_min, wf_max"
run("dpz -> amax-ish (min_max of pz)", ["wf_pz", ], {"wf_pz": DeviceArray((rows, 8192), np.float32)})
run("dpz+atrap+minmax", ["wf_pz", "wf_atrap", mm], {"tp_max": S(), "wf_max": S()})
run("dpz+atrap+minmax+tpt", ["wf_pz", "wf_atrap", mm, "tp_0"], {"tp_0": S(), "tp_max": S()})
run("full C5", ["wf_pz", "wf_atrap", mm, "tp_0", "dwt_haar"], {"tp_0": S(), "tp_min": S(), "tp_max": S(), "wf_min": S(), "wf_max": S(), "dwt_haar": DeviceArray((rows, 256), np.float32)})
run("dpz+dwt", ["wf_pz", "dwt_haar"], {"dwt_haar": DeviceArray((rows, 256), np.float32)})
