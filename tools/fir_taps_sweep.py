#!/usr/bin/env python3
"""Cycles of the short-FIR op ('same' mode on an 8192-sample waveform the trapezoids also read: chunk-padded layout) against the number
of taps (in-kernel op timers).  Usage (GPU box): python tools/fir_taps_sweep.py [rows]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench_configs import synth  # noqa: E402
from dspeed_amd import _lib  # noqa: E402
from dspeed_amd.device import DeviceArray, Stream, sync  # noqa: E402
from dspeed_amd.processing_chain import build_processing_chain  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
st = Stream()
wf, bl, tp = synth(rows, 8192, np.int16, st, bl_lo=-3000.0, bl_hi=3000.0)
sync()
M = "dspeed.processors"
for m in (16, 64, 80, 128, 133, 144, 192, 256):
    rec = {"outputs": ["a", "t"], "processors": {
        "wf_blsub": f"{M}.bl_subtract(waveform, baseline, wf_blsub)",
        "k": f"{M}.t0_filter(8, 125, k({m}, 'f'))" if m == 133 else f"{M}.moving_slope(k({m}, 'f'))",
        "wf_f": f"{M}.convolve_wf(wf_blsub, k, 's', wf_f(8192, 'f'))",
        "t_a, t, lo, a": {"function": "min_max", "module": M, "args": ["wf_f", "t_a", "t", "lo", "a"]},
        "wf_t": f"{M}.trap_norm(wf_blsub, 100, 20, wf_t)", "e": f"{M}.fixed_time_pickoff(wf_t, 4000, 'n', e)"}}
    rec["outputs"].append("e")
    tb = {"waveform": wf, "baseline": bl}
    chain, _, _ = build_processing_chain(rec, tb)
    outs = {v.name: DeviceArray((rows,), np.float32) for v, _l in chain._out_vars.values()}
    chain.link(tb, outs)
    chain.execute()
    chain._chain.profile(True)
    chain.execute()
    pr = chain._chain.profile_read()
    n = max(pr["waveforms"], 1)
    conv = [c / n for o, c in zip(pr["opcodes"], pr["cycles"]) if o == _lib.OP_CONVOLVE]
    geo = chain._chain.geometry(rows)
    print(json.dumps({"taps": m, "convolve_cycles_per_waveform": round(conv[0]), "cycles_per_16_taps_per_320_outputs": round(conv[0] / 25.6 / (m / 16)),
                      "lds_bytes_per_wave": geo["lds_bytes_per_wave"]}), flush=True)
