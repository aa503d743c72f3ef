#!/usr/bin/env python3
"""Every sample of the zero-area cusp's 301 'valid' outputs on the Ge recipe's rows (uint16, baseline subtracted while staging) against float64,
relative to the filtered waveform's peak: the device's float16 and float32 matrix forms and the CPU oracle (np.convolve's float32 summation
restated).  zacEftp picks sample 50, on the filter's flank -- the place where a zero-area kernel's cancellation shows.  The bar is 1e-6."""
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
from test_gpu_icpc_recipe import _synth  # noqa: E402

M = "dspeed.processors"
rng = np.random.default_rng(2031)
n = 64
wf, bl = _synth(rng, n)
rec = {"rows": n, "what": "wf_zac / wf_cusp = fft_convolve_wf((waveform - baseline)[:6092], 5792 taps, 'v'): 301 samples per row"}
x64 = (wf.astype(np.float32) - bl[:, None]).astype(np.float64)[:, :6092]
win = np.lib.stride_tricks.sliding_window_view(x64, 5792, axis=1)
for nm in ("cusp", "zac"):
    k = golden_util.recipe_kernel(nm)
    ref = win @ np.asarray(k, np.float64)[::-1]
    peak = np.abs(ref).max(axis=1, keepdims=True)
    o = oracle.convolve_wf(oracle.bl_subtract(wf.astype(np.float32), bl)[0], k, "v", 301, in_len=6092)[0]
    e = np.abs(o - ref) / peak
    rec[f"oracle:{nm}"] = {"worst_any_sample": float(e.max()), "sample_50": float(e[:, 50].max()), "at_the_peak": float(np.take_along_axis(e, np.abs(ref).argmax(axis=1)[:, None], 1).max())}
    for kind in ("f16", "f32"):
        if kind == "f32":
            os.environ["DSPEED_HIP_FIR_F32"] = "1"
        else:
            os.environ.pop("DSPEED_HIP_FIR_F32", None)
        procs = {"wf_blsub": f"{M}.bl_subtract(waveform, baseline, wf_blsub)", "kern": {"function": f"{nm}_filter", "module": M, "args": ["1250", "188", "28125", "kern(5792, 'f')"]},
                 "wf_f": {"function": "fft_convolve_wf", "module": M, "args": ["wf_blsub[:6092]", "kern", "'v'", "wf_f(301, 'f')"]}}
        chain, _, out = build_processing_chain({"outputs": ["wf_f"], "processors": procs}, {"waveform": wf, "baseline": bl})
        chain.execute()
        e = np.abs(out["wf_f"] - ref) / peak
        rec[f"{kind}:{nm}"] = {"kernels": [kname for _w, kname in chain.kernels()], "worst_any_sample": float(e.max()), "sample_50": float(e[:, 50].max()),
                               "at_the_peak": float(np.take_along_axis(e, np.abs(ref).argmax(axis=1)[:, None], 1).max()),
                               "device_vs_oracle_worst": float((np.abs(out["wf_f"] - o) / peak).max())}
print(json.dumps(rec, indent=1))
