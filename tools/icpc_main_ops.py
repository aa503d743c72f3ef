#!/usr/bin/env python3
"""Per-op cycles of the Ge recipe's MAIN program (in-kernel timers, dsp_chain_profile): where the interpreter's share of a pass goes.
python tools/icpc_main_ops.py [rows]      (DSPEED_HIP_NO_TEAMS=1: one wavefront per row; ICPC_RISE=6,60: pulses with a rise time)"""
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
from dspeed_amd import _lib  # noqa: E402
from dspeed_amd.device import DeviceArray, Stream, sync  # noqa: E402
from dspeed_amd.processing_chain import WaveformInput, build_processing_chain  # noqa: E402

NAMES = {getattr(_lib, k): k[3:] for k in dir(_lib) if k.startswith("OP_")}
NAMES.update({100: "(clear shared LDS)", 101: "(nop)", 102: "(stores)"})
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
st = Stream()
rise = tuple(float(x) for x in os.environ["ICPC_RISE"].split(",")) if os.environ.get("ICPC_RISE") else None
wf, bl, _tp = synth(rows, 8192, np.int16, st, bl_lo=-3000.0, bl_hi=3000.0, rise=rise)
sync()
tb = {"waveform": WaveformInput(wf, 16.0, 48000.0), "baseline": bl}
chain, _, _ = build_processing_chain(recipes.ICPC, tb)
chain.link(tb, {k: DeviceArray((rows,), np.float32) for k in recipes.ICPC["outputs"]})
chain._ensure()
dt = timed(chain, steps=3, warmup=2)
chain._chain.profile(True)
chain.execute()
pr = chain._chain.profile_read()
chain._chain.profile(False)
n = max(pr["waveforms"], 1)
geo = chain._chain.geometry(rows)
print(json.dumps({"rows": rows, "rise_samples": rise, "ms_per_pass": dt * 1e3, "kernel": chain._chain.kernel_name, **geo, "cycles_per_row_all_members": round(sum(pr["cycles"]) / n),
                  "ops": [[NAMES.get(o, str(o)), round(c / n)] for o, c in zip(pr["opcodes"], pr["cycles"])]}))
