#!/usr/bin/env python3
"""PCIe-inclusive rate of the whole Ge recipe (tests/recipes.py ICPC): uint16 rows of 8192 samples in host memory (NumPy), 27 result
columns back in host memory, through ProcessingChain.execute() -- what a build_dsp-style loop sees per file chunk.
Usage (GPU box): python tools/e2e_recipe_rate.py [rows]   (default 400 000 rows = 6.5 GB of host memory)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import recipes  # noqa: E402
from dspeed_amd.processing_chain import WaveformInput, build_processing_chain  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
rng = np.random.default_rng(3)
base_rows = min(rows, 16384)  # (distinct rows; a longer batch repeats them: the rate does not depend on what the rows hold)
i = np.arange(8192, dtype=np.float32)[None, :]
A = rng.uniform(500, 15000, (base_rows, 1)).astype(np.float32)
t0 = np.floor(rng.uniform(0.4, 0.5, (base_rows, 1)) * 8192).astype(np.float32)
wf = np.empty((rows, 8192), dtype=np.uint16)
for a in range(0, base_rows, 8192):  # (in slabs: the float32 intermediate of the whole batch would not fit next to it)
    b = min(base_rows, a + 8192)
    x = 10000.0 + A[a:b] * np.exp(-(i - t0[a:b]) / 1716.28) * (i >= t0[a:b]) + 5.0 * rng.standard_normal((b - a, 8192), dtype=np.float32)
    wf[a:b] = np.rint(x).astype(np.uint16)
for a in range(base_rows, rows, base_rows):
    wf[a:a + base_rows] = wf[:min(base_rows, rows - a)]
tb = {"waveform": WaveformInput(wf, 16.0, 48000.0), "baseline": np.full(rows, 10000.0, np.float32)}
res = {}
chain, _, out = build_processing_chain(recipes.ICPC, tb)
configs = [("128 MiB pieces, 2 in flight", 128 << 20, 2), ("256 MiB pieces (default), 2 in flight (default)", 256 << 20, 2),
           ("512 MiB pieces, 2 in flight", 512 << 20, 2), ("256 MiB pieces, 1 in flight", 256 << 20, 1), ("1 GiB pieces, 1 in flight", 1 << 30, 1)]
for label, piece, lanes in configs:
    chain.pipeline_bytes, chain.pieces_in_flight = piece, lanes
    chain.execute()
    before = chain.get_timing()
    t = time.perf_counter()
    for _ in range(2):
        chain.execute()
    dt = (time.perf_counter() - t) / 2
    after = chain.get_timing()
    res[label] = {"waveforms_per_s": round(rows / dt), "GB_per_s_over_pcie": round(rows * 16384 / dt / 1e9, 2), "ms_per_pass": round(dt * 1e3, 1),
                  "host_seconds_per_pass": {k: round((after[k] - before[k]) / 2, 3) for k in after}}
print(json.dumps({"recipe": "ICPC", "rows": rows, "wf_len": 8192, "row_dtype": "uint16", "pcie_GB_per_s_of_the_simple_chain": 53.0, "results": res}, indent=1))
