#!/usr/bin/env python3
"""Per-op cycles of the generic waveform VM on the BASELINE recipes (in-kernel timers, dsp_chain_profile).  One JSON line per recipe.
Usage (GPU box): python tools/vm_op_profile.py [rows]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import recipes  # noqa: E402
from bench_configs import synth  # noqa: E402
from dspeed_amd import _lib  # noqa: E402
from dspeed_amd.device import DeviceArray, Stream, sync  # noqa: E402
from dspeed_amd.processing_chain import build_processing_chain  # noqa: E402

NAMES = {getattr(_lib, k): k[3:] for k in dir(_lib) if k.startswith("OP_")}
NAMES[100] = "(clear shared LDS)"
NAMES[101] = "(bl_subtract, done by the load)"
NAMES[102] = "(STORE_SCALAR run as one op)"


def profile(label, recipe, tb, rows):
    chain, _, _ = build_processing_chain(recipe, tb)
    outs = {}
    for k, (v, length) in chain._out_vars.items():
        outs[v.name] = DeviceArray((rows,) if length is None else (rows, length), np.float32)
    chain.link(tb, outs)
    chain._ensure()
    chain._chain.set_fused(0)
    chain.execute()
    chain._chain.profile(True)
    chain.execute()
    pr = chain._chain.profile_read()
    geo = chain._chain.geometry(rows)
    n = max(pr["waveforms"], 1)
    print(json.dumps({"recipe": label, "rows": rows, **geo, "cycles_per_waveform": round(sum(pr["cycles"]) / n),
                      "ops": [[NAMES.get(o, str(o)), round(c / n)] for o, c in zip(pr["opcodes"], pr["cycles"])]}), flush=True)


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    st = Stream()
    wf, bl, tp = synth(rows, 4096, np.float32, st)
    sync()
    profile("C2 (float32 x 4096)", recipes.C2, {"waveform": wf, "baseline": bl, "t_pick": tp}, rows)
    del wf
    wf, bl, tp = synth(rows, 4096, np.int16, st, bl_lo=-3000.0, bl_hi=3000.0)
    sync()
    profile("C2 (int16 x 4096)", recipes.C2, {"waveform": wf, "baseline": bl, "t_pick": tp}, rows)
    del wf
    r3 = max(1000, rows // 4)
    wf, bl, tp = synth(r3, 8192, np.float32, st)
    sync()
    profile("C3 (float32 x 8192, 2 x 5792-tap FIR)", recipes.C3, {"waveform": wf, "baseline": bl}, r3)
    del wf
    wf, bl, tp = synth(rows, 8192, np.int16, st, bl_lo=-3000.0, bl_hi=3000.0)
    thr = DeviceArray.from_numpy(np.full(rows, 20.0, dtype=np.float32))
    sync()
    profile("C5 (int16 x 8192)", recipes.C5, {"waveform": wf, "thr": thr}, rows)


if __name__ == "__main__":
    main()
