#!/usr/bin/env python3
"""How far the rise-time walks of the Ge recipe go on the synthetic rows of tools/icpc_rate.py (samples between a walk's start and its result)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import recipes  # noqa: E402
from bench_configs import synth  # noqa: E402
from dspeed_amd.device import Stream, sync  # noqa: E402
from dspeed_amd.processing_chain import WaveformInput, build_processing_chain  # noqa: E402

rows = 16384
st = Stream()
wf, bl, _tp = synth(rows, 8192, np.int16, st, bl_lo=-3000.0, bl_hi=3000.0)
sync()
tb = {"waveform": WaveformInput(wf, 16.0, 48000.0), "baseline": bl}
chain, _, out = build_processing_chain(recipes.ICPC, tb)
chain.execute()
o = {k: np.asarray(v) for k, v in out.items()}
rec = {}
for a, b in (("tp_0_est", "tp_99"), ("tp_99", "tp_90"), ("tp_90", "tp_50"), ("tp_50", "tp_10"), ("tp_0_est", "tp_100")):
    d = np.abs(o[b] - o[a]) / 16.0
    ok = ~np.isnan(d)
    rec[f"{a}->{b}"] = {"nan": int((~ok).sum()), "median": float(np.median(d[ok])) if ok.any() else None,
                        "p90": float(np.percentile(d[ok], 90)) if ok.any() else None, "max": float(d[ok].max()) if ok.any() else None,
                        "mean": float(d[ok].mean()) if ok.any() else None}
rec["trapTmax"] = [float(np.nanmin(o["trapTmax"])), float(np.nanmax(o["trapTmax"]))]
print(json.dumps(rec, indent=1))
