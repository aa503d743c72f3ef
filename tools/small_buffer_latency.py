import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import recipes
from dspeed_amd import _lib
from dspeed_amd.device import DeviceArray, Stream, sync
from dspeed_amd.processing_chain import build_processing_chain
for rows in (3200, 32000, 320000):
    st = Stream()
    wf = DeviceArray((rows, 4096), np.float32); bl = DeviceArray((rows,), np.float32); tp = DeviceArray((rows,), np.float32)
    _lib.check(_lib.lib().dsp_synth_waveforms(wf.ptr, _lib.F32, rows, 4096, 4096, bl.ptr, tp.ptr, 1, 0, 1716.28, 5.0, 775.0, 9000.0, 11000.0, 500.0, 15000.0, st.ptr)); sync()
    tb = {"waveform": wf, "baseline": bl, "t_pick": tp}
    chain, _, _ = build_processing_chain(recipes.C2, tb); chain.link(tb, {"trapEftp": DeviceArray((rows,), np.float32)})
    chain.execute(); n = 200
    t = time.perf_counter()
    for _ in range(n): chain.execute()
    dt = (time.perf_counter() - t) / n
    print(f"rows {rows:7d}: {dt*1e6:8.1f} us per execute()  -> {rows/dt/1e6:7.1f} M wf/s")
