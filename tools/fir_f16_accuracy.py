#!/usr/bin/env python3
"""Accuracy of the two forms of the 'valid' FIR + amax kernel on C3's geometry (5792-tap cusp / zac kernels, 6092-sample slices) against
float64, relative to each filtered waveform's peak: the float16 matrix instructions on two-way split operands (dsp_fir_f16.hip) and the
float32 ones (dsp_fir_mfma.hip).  The bar is 1e-6."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import recipes  # noqa: E402
from dspeed_amd.processing_chain import build_processing_chain  # noqa: E402

rng = np.random.default_rng(30)
n_wf, L = 96, 8192
i = np.arange(L)[None, :]
B = rng.uniform(9000, 11000, (n_wf, 1))
A = rng.uniform(500, 15000, (n_wf, 1))
t0 = np.floor(rng.uniform(0.45, 0.55, (n_wf, 1)) * L)
wf = (B + A * np.exp(-(i - t0) / 1716.28) * (i >= t0) + 5.0 * rng.standard_normal((n_wf, L))).astype(np.float32)
bl = B[:, 0].astype(np.float32)
xb = (wf - bl[:, None]).astype(np.float32)
rec = {}
for kind in ("f16", "f32"):
    if kind == "f32":
        os.environ["DSPEED_HIP_FIR_F32"] = "1"
    else:
        os.environ.pop("DSPEED_HIP_FIR_F32", None)
    chain, _, out = build_processing_chain(recipes.C3, {"waveform": wf, "baseline": bl})
    chain.execute()
    for nm in ("cusp", "zac"):
        k64 = np.asarray(chain._consts[f"taps:{nm}_kernel"][:5792], dtype=np.float64)[::-1]
        x64 = xb[:, :6092].astype(np.float64)
        win = np.lib.stride_tricks.sliding_window_view(x64, 5792, axis=1)
        ref = win @ k64
        err = np.abs(out[f"{nm}Emax"] - ref.max(axis=1)) / np.abs(ref).max(axis=1)
        rec[f"{kind}:{nm}"] = {"kernel": chain._chain.kernel_name, "max_rel_to_peak": float(err.max()), "median": float(np.median(err))}
print(json.dumps(rec, indent=1))
