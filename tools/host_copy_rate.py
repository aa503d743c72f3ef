#!/usr/bin/env python3
"""How fast host threads move rows from a NumPy array into a page-locked staging buffer (the first leg of ProcessingChain.execute for
host-resident columns), by number of threads and block size -- the rate the PCIe link (about 53 GB/s) has to be fed at.
Usage (GPU box): python tools/host_copy_rate.py"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dspeed_amd.device import PinnedArray  # noqa: E402

src = np.random.default_rng(0).integers(0, 60000, (65536, 8192), dtype=np.uint16)  # 1 GiB
res = {"cpus": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "results": []}
for block_mib in (32, 256):
    rows = (block_mib << 20) // (8192 * 2)
    pinned = PinnedArray((rows, 8192), np.uint16)  # (the view below lives only as long as this object)
    dst = pinned.array
    for threads in (1, 4, 8, 12, 16, 24):
        pool = ThreadPoolExecutor(max_workers=threads)
        step = -(-rows // threads)

        def copy(base):
            list(pool.map(lambda a: np.copyto(dst[a:min(rows, a + step)], src[base + a:base + min(rows, a + step)]), range(0, rows, step)))

        copy(0)
        t = time.perf_counter()
        reps = 0
        for base in range(0, len(src) - rows + 1, rows):
            copy(base)
            reps += 1
        dt = time.perf_counter() - t
        res["results"].append({"block_MiB": block_mib, "threads": threads, "GB_per_s": round(reps * dst.nbytes / dt / 1e9, 1)})
        pool.shutdown()
print(json.dumps(res, indent=1))
