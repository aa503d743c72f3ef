#!/usr/bin/env python3
"""The whole Ge recipe on a device-resident batch, nothing else: for rocprofv3 --kernel-trace --stats (which launches make up a pass) and for
A/B runs.  python tools/icpc_rate.py [rows] [steps]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import recipes  # noqa: E402
from bench_configs import synth, timed  # noqa: E402
from dspeed_amd.device import DeviceArray, Stream, sync  # noqa: E402
from dspeed_amd.processing_chain import WaveformInput, build_processing_chain  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
st = Stream()
# ICPC_RISE="lo,hi": pulses with a charge-collection time of lo .. hi samples (6 .. 60 = 0.1 .. 1 us at 16 ns) instead of one-sample steps
rise = tuple(float(x) for x in os.environ["ICPC_RISE"].split(",")) if os.environ.get("ICPC_RISE") else None
wf, bl, _tp = synth(rows, 8192, np.int16, st, bl_lo=-3000.0, bl_hi=3000.0, rise=rise)
sync()
tb = {"waveform": WaveformInput(wf, 16.0, 48000.0), "baseline": bl}
drop = set(filter(None, os.environ.get("ICPC_DROP", "").split(",")))  # (outputs left out: what the rest of the recipe costs without them)
outs = [o for o in recipes.ICPC["outputs"] if o not in drop]
chain, _, _ = build_processing_chain(recipes.ICPC, tb, outputs=outs)
chain.link(tb, {k: DeviceArray((rows,), np.float32) for k in outs})
chain._ensure()
dt = timed(chain, steps=steps, warmup=2)
print(json.dumps({"recipe": "ICPC", "dropped": sorted(drop), "rise_samples": rise, "rows": rows, "steps": steps, "ms_per_pass": dt * 1e3, "waveforms_per_s": round(rows / dt),
                  "kernels": [s["chain"].kernel_name for s in chain._stages] + [chain._chain.kernel_name],
                  "stages": [s["what"] for s in chain._stages]}))
