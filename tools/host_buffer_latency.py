import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import recipes
from dspeed_amd.processing_chain import build_processing_chain
rows = 3200
rng = np.random.default_rng(1)
wf = (10000 + 5 * rng.standard_normal((rows, 4096), dtype=np.float32)).astype(np.float32); wf[:, 2048:] += 3000
tb = {"waveform": wf, "baseline": np.full(rows, 10000, np.float32), "t_pick": np.full(rows, 2048 + 775.4, np.float32)}
chain, _, out = build_processing_chain(recipes.C2, tb)
chain.execute(); chain.execute()
t = time.perf_counter(); n = 50
for _ in range(n): chain.execute()
dt = (time.perf_counter() - t) / n
print(f"host columns, {rows} rows per execute(): {dt*1e6:.0f} us -> {rows/dt/1e6:.2f} M wf/s, {rows*16396/dt/1e9:.1f} GB/s")
