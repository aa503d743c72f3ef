"""Per output of the whole Ge recipe: in how many rows the device equals the all-oracle run bit for bit, and the largest deviation -- the
numbers the tolerances of tests/test_gpu_icpc_recipe.py are set from.  Usage (GPU box): python tools/icpc_parity_measure.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import recipes
from test_gpu_icpc_recipe import _synth, _expected, F
from dspeed_amd.processing_chain import WaveformInput, build_processing_chain
res = {}
for t0_kind, rows_dtype in (("per_row", np.uint16), ("constant", np.uint16), ("per_row", np.int32)):
    rng = np.random.default_rng(2026)
    n = 48
    wf, bl = _synth(rng, n)
    wf = wf.astype(rows_dtype)
    ft = np.float64 if rows_dtype == np.int32 else np.float32
    t0_ns = (rng.integers(2900, 3100, n) * 16).astype(F) if t0_kind == "per_row" else np.full(n, 48000.0, dtype=F)
    tb = {"waveform": WaveformInput(wf, 16.0, t0_ns if t0_kind == "per_row" else 48000.0), "baseline": bl}
    chain, mask, out = build_processing_chain(recipes.ICPC, tb)
    chain.execute()
    want, tp0 = _expected(wf, bl, t0_ns, ft)
    r = {}
    for k in out:
        a, b = out[k].astype(np.float64), want[k].astype(np.float64)
        same = (a == b) | (np.isnan(a) & np.isnan(b))
        scale = np.maximum(np.abs(b), 1e-300)
        r[k] = {"rows_equal": int(same.sum()), "max_rel": float(np.nanmax(np.abs(a - b) / scale)), "max_abs": float(np.nanmax(np.abs(a - b)))}
    res[f"{t0_kind}/{np.dtype(rows_dtype).name}"] = r
print(json.dumps(res, indent=1))
