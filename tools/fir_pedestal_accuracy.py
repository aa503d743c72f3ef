#!/usr/bin/env python3
"""The zero-area cusp on rows whose pedestal was NOT subtracted (uint16 samples around 10 000 ADC, pulses of 500 - 15 000): the kernel removes
the pedestal itself, so the filtered waveform's peak is far below the products that make it -- the case where a float16 split (22 bits of each
operand, dsp_fir_f16.hip) could fall short of the float32 form.  Every one of the 301 'valid' outputs against float64, relative to the filtered
waveform's peak, for the device's float16 and float32 forms ('valid' + amax, and the kept output) and the CPU oracle.  The bar is 1e-6."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util  # noqa: E402
import oracle  # noqa: E402
from dspeed_amd.processing_chain import build_processing_chain  # noqa: E402

M = "dspeed.processors"
rng = np.random.default_rng(77)
n, L = 96, 8192
i = np.arange(L)[None, :]
B = rng.uniform(9000, 11000, (n, 1))
A = rng.uniform(500, 15000, (n, 1))
A[:8] = 500.0  # (the smallest pulses on the largest pedestals are the hard rows)
B[:8] = 11000.0
t0 = np.floor(rng.uniform(0.45, 0.55, (n, 1)) * L)
rec = {"rows": n, "what": "convolve_wf(waveform[:6092] (pedestal 9 000 - 11 000 ADC, not subtracted), zero-area cusp of 5792 taps, 'v'): 301 samples per row"}
for dt_name, dt in (("uint16", np.uint16), ("float32", np.float32)):
    wf = (B + A * np.exp(-(i - t0) / 1716.28) * (i >= t0) + 5.0 * rng.standard_normal((n, L)))
    wf = np.rint(wf).astype(dt) if dt == np.uint16 else wf.astype(dt)
    x64 = wf.astype(np.float64)[:, :6092]
    win = np.lib.stride_tricks.sliding_window_view(x64, 5792, axis=1)
    for nm in ("zac", "cusp"):
        k = golden_util.recipe_kernel(nm)
        ref = win @ np.asarray(k, np.float64)[::-1]
        peak = np.abs(ref).max(axis=1, keepdims=True)
        o = oracle.convolve_wf(wf.astype(np.float32), k, "v", 301, in_len=6092)[0]
        rec[f"{dt_name}:{nm}:oracle"] = {"worst_any_sample": float((np.abs(o - ref) / peak).max()), "peak_over_pedestal_min": float((peak[:, 0] / B[:, 0]).min())}
        for kind in ("f16", "f32"):
            if kind == "f32":
                os.environ["DSPEED_HIP_FIR_F32"] = "1"
            else:
                os.environ.pop("DSPEED_HIP_FIR_F32", None)
            procs = {"kern": {"function": f"{nm}_filter", "module": M, "args": ["1250", "188", "28125", "kern(5792, 'f')"]},
                     "wf_f": {"function": "convolve_wf", "module": M, "args": ["waveform[:6092]", "kern", "'v'", "wf_f(301, 'f')"]},
                     "emax": "numpy.amax(wf_f, 1, emax)"}
            for form, outs in (("kept", ["wf_f"]), ("amax", ["emax"])):
                chain, _, out = build_processing_chain({"outputs": outs, "processors": procs}, {"waveform": wf})
                chain.execute()
                if form == "kept":
                    e = np.abs(out["wf_f"] - ref) / peak
                    r = {"worst_any_sample": float(e.max()), "vs_oracle_worst": float((np.abs(out["wf_f"] - o) / peak).max())}
                else:
                    r = {"amax_rel_to_peak": float((np.abs(out["emax"] - ref.max(axis=1)) / peak[:, 0]).max()),
                         "amax_vs_oracle": float((np.abs(out["emax"] - o.max(axis=1)) / peak[:, 0]).max())}
                r["kernels"] = [kname for _w, kname in chain.kernels()]
                rec[f"{dt_name}:{nm}:{kind}:{form}"] = r
print(json.dumps(rec, indent=1))
