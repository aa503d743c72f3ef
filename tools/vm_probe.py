#!/usr/bin/env python3
"""One small program on the waveform VM (fused kernels off), for counter runs that tell which op an LDS bank conflict or a stall belongs to:
    python tools/vm_probe.py <probe> [rows]        probes: load, bl, pz, trap, trap_pick, c2, minmax, tpt
Prints one JSON line (rate); run it under rocprofv3 --pmc ... to get the counters of its dsp_vm_kernel launches."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_configs import synth, timed  # noqa: E402
from dspeed_amd.device import DeviceArray, Stream, sync  # noqa: E402
from dspeed_amd.processing_chain import WaveformInput, build_processing_chain  # noqa: E402

M = "dspeed.processors"
probe = sys.argv[1] if len(sys.argv) > 1 else "c2"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
P = {"wf_bl": f"{M}.bl_subtract(waveform, baseline, wf_bl)", "wf_pz": f"{M}.pole_zero(wf_bl, 1716.28, wf_pz)",
     "wf_trap": f"{M}.trap_filter(wf_pz, 625, 188, wf_trap)", "e": f"{M}.fixed_time_pickoff(wf_trap, t_pick, 'l', e)",
     "a, b, lo, hi": f"{M}.min_max(waveform, a, b, lo, hi)", "lo_bl, hi_bl, c, d": f"{M}.min_max(wf_bl, lo_bl, hi_bl, c, d)",
     "s0": "waveform[100]", "s_bl": "wf_bl[100]", "s_pz": "wf_pz[100]", "s_tr": "wf_trap[100]",
     "tp": f"{M}.time_point_thresh(wf_pz, 500, 2000, 1, tp)"}
OUTS = {"load": ["s0"], "bl": ["s_bl"], "pz": ["s_pz"], "trap": ["s_tr"], "trap_pick": ["e"], "c2": ["e"], "minmax": ["hi"], "tpt": ["tp"]}
st = Stream()
wf, bl, tp = synth(rows, 4096, np.float32, st)
sync()
tb = {"waveform": WaveformInput(wf, 16.0, 0.0), "baseline": bl, "t_pick": tp}
outs = OUTS[probe]
chain, _, _ = build_processing_chain({"outputs": outs, "processors": P}, tb)
chain.link(tb, {k: DeviceArray((rows,), np.float32) for k in outs})
chain._ensure()
chain._chain.set_fused(0)
dt = timed(chain, steps=5, warmup=2)
print(json.dumps({"probe": probe, "rows": rows, "kernel": chain._chain.kernel_name, "ops": len(chain.program.ops), "ms": dt * 1e3,
                  "waveforms_per_s": rows / dt, "lds_bytes_per_wave": chain._chain.geometry(rows)["lds_bytes_per_wave"]}))
