#!/usr/bin/env python3
"""Accuracy of the kept-output float16 FIR (dsp_fir_f16.hip, STORE form) for short kernels against float64, relative to each filtered
waveform's peak: the Ge recipe's t0 filter (133 taps, differentiating) on pole-zero corrected pulses, 's' mode, and a 96-tap and a 250-tap
one.  Run with DSPEED_HIP_LIB pointing at a build with -DF16_KFLUSH_SHORT=0 for the float64-every-128-samples form beside it."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dspeed_amd.processors as P  # noqa: E402
from dspeed_amd.processing_chain import build_processing_chain  # noqa: E402

M = "dspeed.processors"
rng = np.random.default_rng(31)
n_wf, L = 256, 8192
i = np.arange(L)[None, :]
A = rng.uniform(500, 15000, (n_wf, 1))
t0 = np.floor(rng.uniform(0.3, 0.6, (n_wf, 1)) * L)
rise = rng.uniform(5, 60, (n_wf, 1))
wf = (A * np.clip((i - t0) / rise, 0, 1) + 5.0 * rng.standard_normal((n_wf, L)) + rng.uniform(-50, 50, (n_wf, 1))).astype(np.float32)
rec = {}
for rise_k, fall_k in ((8, 125), (16, 80), (50, 200)):
    m = rise_k + fall_k
    recipe = {"outputs": ["wf_f"], "processors": {
        "wf_c": "waveform + 0",
        "kern": {"function": "t0_filter", "module": M, "args": [rise_k, fall_k, f"kern({m}, 'f')"]},
        "wf_f": {"function": "convolve_wf", "module": M, "args": ["wf_c", "kern", "'s'", f"wf_f({L}, 'f')"]}}}
    chain, _, out = build_processing_chain(recipe, {"waveform": wf})
    chain.execute()
    kern = np.zeros(m, dtype=np.float32)
    P.t0_filter(rise_k, fall_k, kern)
    full = np.stack([np.convolve(r.astype(np.float64), kern.astype(np.float64), "full") for r in wf])
    lo = (m - 1) // 2  # numpy 'same'
    ref = full[:, lo:lo + L]
    got = np.asarray(out["wf_f"], dtype=np.float64)
    # the reference's own 's' is np.convolve(..., 'same'); if the alignment differs by the kernel's parity this shows as a gross error
    err = np.max(np.abs(got - ref), axis=1) / np.abs(ref).max(axis=1)
    rec[f"t0_filter({rise_k},{fall_k})"] = {"kernels": [k for _w, k in chain.kernels()], "max_rel_to_peak": float(err.max()), "median": float(np.median(err))}
print(json.dumps(rec))
