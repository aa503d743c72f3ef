#!/usr/bin/env python3
"""Why the matrix-core FIR adds its partial sums in float64: error of ONE float32 accumulation chain over the 5792 taps of the cusp / zac
kernels on BASELINE's synthetic waveforms, relative to the filtered waveform's peak (CPU, NumPy; no GPU).  The bar is 1e-6."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dspeed_amd.processors import _cusp_filter, _zac_filter  # host-side generators (no device call)

n, m = 6092, 5792
kern = {}
for name, gen in (("cusp", _cusp_filter), ("zac", _zac_filter)):
    k = np.zeros(m, np.float32)
    gen(None, np.float32(1250), np.float32(188), np.float32(28125), k)
    kern[name] = k
rng = np.random.default_rng(1)
worst = {"chain": 0.0, "chunk128": 0.0}
for trial in range(6):
    i = np.arange(8192)
    B, A, t0 = rng.uniform(9000, 11000), rng.uniform(500, 15000), np.floor(rng.uniform(.45, .55) * 8192)
    x = (B + A * np.exp(-(i - t0) / 1716.28) * (i >= t0) + 5 * rng.standard_normal(8192)).astype(np.float32)
    xb = (x - np.float32(B)).astype(np.float32)[:n]
    for name, k in kern.items():
        kr = k[::-1].copy()
        js = range(0, 301, 10)
        ref = np.array([np.dot(xb[j:j + m].astype(np.float64), kr.astype(np.float64)) for j in js])
        prods = [(xb[j:j + m] * kr).astype(np.float32) for j in js]
        chain = np.array([np.cumsum(pr, dtype=np.float32)[-1] for pr in prods])
        chunk = np.array([sum(float(np.cumsum(pr[c:c + 128], dtype=np.float32)[-1]) for c in range(0, m, 128)) for pr in prods])
        peak = np.abs(ref).max()
        worst["chain"] = max(worst["chain"], np.abs(chain - ref).max() / peak)
        worst["chunk128"] = max(worst["chunk128"], np.abs(chunk - ref).max() / peak)
print({k: f"{v:.2e}" for k, v in worst.items()})
