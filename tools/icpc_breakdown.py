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
    per_op(recipes.ICPC, tb, list(recipes.ICPC["outputs"]), rows, "ICPC")
    wf5, _bl, _tp = synth(rows, 8192, np.int16, st, bl_lo=-3000.0, bl_hi=3000.0)
    thr = DeviceArray.from_numpy(np.full(rows, 20.0, dtype=np.float32))
    sync()
    per_op(recipes.C5, {"waveform": wf5, "thr": thr}, list(recipes.C5["outputs"]), rows, "C5")


def per_op(recipe, tb, outs, rows, label):
    """cycles per op of the device program, from the in-kernel timers (dsp_chain_profile)"""
    names = {getattr(_lib, k): k[3:] for k in dir(_lib) if k.startswith("OP_")}
    names[100] = "(clear shared LDS)"
    names[101] = "(bl_subtract, done by the load)"
    names[102] = "(STORE_SCALAR run as one op)"
    chain, _, _ = build_processing_chain(recipe, tb, outputs=outs)
    out_cols = {}
    for k in outs:
        v = chain._out_vars[f"out:{k}"][1]
        out_cols[k] = DeviceArray((rows,) if v is None else (rows, v), np.float32)
    chain.link(tb, out_cols)
    chain._ensure()
    chain._chain.set_fused(0)
    chain.execute()
    chain._chain.profile(True)
    chain.execute()
    pr = chain._chain.profile_read()
    total = sum(pr["cycles"])
    agg = {}
    for oc, cyc in zip(pr["opcodes"], pr["cycles"]):
        agg[names.get(oc, str(oc))] = agg.get(names.get(oc, str(oc)), 0) + cyc
    print(json.dumps({"recipe": label, "per_op_profile": True, "waveforms_sampled": pr["waveforms"],
                      "cycles_per_waveform": round(total / max(pr["waveforms"], 1)),
                      "share_by_opcode": {k: round(v / total, 4) for k, v in sorted(agg.items(), key=lambda kv: -kv[1])},
                      "ops": [[names.get(oc, str(oc)), round(c / max(pr["waveforms"], 1))] for oc, c in zip(pr["opcodes"], pr["cycles"])]}), flush=True)
    chain._chain.profile(False)


if __name__ == "__main__":
    main()
