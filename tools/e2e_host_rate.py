#!/usr/bin/env python3
"""PCIe-inclusive rate of the C2 recipe when the waveforms live in host memory: NumPy columns in,
NumPy column out, through ProcessingChain.execute() (pinned in place, overlapped pieces).  Not part of bench.py's `value`.
Usage (on the GPU box): python tools/e2e_host_rate.py [rows]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import recipes  # noqa: E402
from dspeed_amd.processing_chain import build_processing_chain  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
rng = np.random.default_rng(1)
wf = (10000 + 5 * rng.standard_normal((rows, 4096), dtype=np.float32)).astype(np.float32)
wf[:, 2048:] += 3000
tb = {"waveform": wf, "baseline": np.full(rows, 10000, np.float32), "t_pick": np.full(rows, 2048 + 625 + 150.4, np.float32)}
res = {}
for mode, in_place in (("staged through page-locked buffers (default)", False), ("columns page-locked in place (pin_in_place)", True)):
    chain, _, out = build_processing_chain(recipes.C2, tb)
    chain.pin_in_place = in_place
    for label, piece in (("one piece", 1 << 62), ("64 MiB pieces, overlapped", 64 << 20), ("256 MiB pieces, overlapped", 256 << 20)):
        chain.pipeline_bytes = piece
        chain.execute()  # (first call allocates the piece / staging buffers, or pins the columns)
        t = time.perf_counter()
        for _ in range(3):
            chain.execute()
        dt = (time.perf_counter() - t) / 3
        res[f"{mode}: {label}"] = {"waveforms_per_s": rows / dt, "GB_per_s_over_pcie": rows * 16396 / dt / 1e9}
    del chain
print(json.dumps({"rows": rows, "wf_len": 4096, "host_copy_threads": 8, "results": res}, indent=1))
