#!/usr/bin/env python3
"""The whole Ge recipe (tests/recipes.py ICPC, the structure of the reference's icpc-dsp-config.json) on a device-resident synthetic batch
of 8192-sample 16-bit rows: throughput of the full recipe and of sub-recipes that request fewer outputs (the dependency resolution drops
what they do not need), to see where the time goes.  One JSON object per line.  Secondary measurement, not the contract bench.

Usage (GPU box): python tools/icpc_breakdown.py [rows]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import recipes  # noqa: E402
from bench_configs import synth, timed  # noqa: E402
from dspeed_amd import _lib  # noqa: E402
from dspeed_amd.device import DeviceArray, Stream, sync  # noqa: E402
from dspeed_amd.processing_chain import WaveformInput, build_processing_chain  # noqa: E402

SUBSETS = [
    ("load+min_max", ["tp_max", "wf_max"]),
    ("baseline fit (700 samples)", ["bl_std"]),
    ("+ pole_zero + tail fit (6592 samples)", ["pz_std"]),
    ("+ pole_zero + trap_norm amax", ["trapTmax"]),
    ("t0: 133-tap FIR, min_max, threshold walk", ["tp_0_est"]),
    ("t0 + asym trap walk", ["tp_0_est", "tp_0_atrap"]),
    ("rise-time points", ["tp_10", "tp_50", "tp_90", "tp_99", "tp_100"]),
    ("energy: trap pick-off on the grid", ["trapEftp", "trapEmax"]),
    ("cusp: 5792-tap FIR on wf[:6092]", ["cuspEmax", "cuspEftp"]),
    ("drift time", ["QDrift", "dt_eff"]),
    ("current: window, upsample x16, 3 moving averages", ["A_max", "tp_aoe_samp"]),
    ("whole recipe", list(recipes.ICPC["outputs"])),
]


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    st = Stream()
    wf, bl, _tp = synth(rows, 8192, np.int16, st, bl_lo=-3000.0, bl_hi=3000.0)
    sync()
    tb = {"waveform": WaveformInput(wf, 16.0, 48000.0), "baseline": bl}
    for label, outs in SUBSETS:
        chain, _, _ = build_processing_chain(recipes.ICPC, tb, outputs=outs)
        chain.link(tb, {k: DeviceArray((rows,), np.float32) for k in outs})
        chain._ensure()
        lds, wpb, blocks = C.c_int(), C.c_int(), C.c_int()
        _lib.check(_lib.lib().dsp_chain_geometry(chain._chain._h, rows, C.byref(lds), C.byref(wpb), C.byref(blocks)), what="geometry")
        dt = timed(chain, steps=3, warmup=1)
        print(json.dumps({"recipe": "ICPC", "outputs": label, "n_outputs": len(outs), "ops": len(chain.program.ops),
                          "slots": len(chain.program.slots), "lds_bytes_per_waveform": lds.value, "waves_per_block": wpb.value,
                          "rows": rows, "waveforms_per_s": round(rows / dt), "us_per_waveform_per_cu": round(dt / rows * 256 * 1e6, 2)}), flush=True)


if __name__ == "__main__":
    main()
