import sys, os, time, json
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from dspeed_amd import _lib
from dspeed_amd.device import DeviceArray, Event, Stream, sync
from dspeed_amd.processing_chain import build_processing_chain
rows = 100000
st = Stream()
def synth(rows, wf_len, dtype):
    wf = DeviceArray((rows, wf_len), dtype); bl = DeviceArray((rows,), np.float32); tp = DeviceArray((rows,), np.float32)
    code = _lib.I16 if np.dtype(dtype) == np.int16 else _lib.F32
    _lib.check(_lib.lib().dsp_synth_waveforms(wf.ptr, code, rows, wf_len, wf_len, bl.ptr, tp.ptr, 1234, 0, 1716.28, 5.0, 775.0, -3000.0 if code==_lib.I16 else 9000.0, 3000.0 if code==_lib.I16 else 11000.0, 500.0, 15000.0, st.ptr))
    sync(); return wf, bl, tp
def timed(chain, steps=3):
    chain.execute(); e0, e1 = Event(), Event(); e0.record(chain._stream)
    for _ in range(steps): chain.execute()
    e1.record(chain._stream); sync(); return e0.elapsed_ms(e1)*1e-3/steps
def run(label, recipe, tb, outs):
    chain,_,_ = build_processing_chain(recipe, tb); chain.link(tb, outs); dt = timed(chain)
    print(f"{label:28s} {rows/dt/1e6:9.2f} M wf/s   {dt*1e3:8.2f} ms", flush=True)
for dtype in (np.float32, np.int16):
    L = 8192
    wf, bl, tp = synth(rows, L, dtype)
    thr = DeviceArray.from_numpy(np.full(rows, 20.0, np.float32))
    W = lambda: DeviceArray((rows, L), np.float32)
    S = lambda: DeviceArray((rows,), np.float32)
    M = "dspeed.processors"
    print("dtype", np.dtype(dtype).name)
    run("load+store (copy via pz?)", {"outputs": ["o"], "processors": {"o": f"{M}.bl_subtract(waveform, 0, o)"}}, {"waveform": wf}, {"o": W()})
    run("pole_zero", {"outputs": ["o"], "processors": {"o": f"{M}.pole_zero(waveform, 1716.28, o)"}}, {"waveform": wf}, {"o": W()})
    run("double_pole_zero", {"outputs": ["o"], "processors": {"o": f"{M}.double_pole_zero(waveform, 1716.28, 62.5, 0.02, o)"}}, {"waveform": wf}, {"o": W()})
    run("asym_trap", {"outputs": ["o"], "processors": {"o": f"{M}.asym_trap_filter(waveform, 8, 4, 125, o)"}}, {"waveform": wf}, {"o": W()})
    run("trap_filter", {"outputs": ["o"], "processors": {"o": f"{M}.trap_filter(waveform, 1250, 376, o)"}}, {"waveform": wf}, {"o": W()})
    run("min_max", {"outputs": ["a","b","c","d"], "processors": {"a, b, c, d": {"function": "min_max", "module": M, "args": ["waveform","a","b","c","d"]}}}, {"waveform": wf}, {"a": S(), "b": S(), "c": S(), "d": S()})
    run("tpt backward from 6000", {"outputs": ["o"], "processors": {"o": f"{M}.time_point_thresh(waveform, thr, 6000, 0, o)"}}, {"waveform": wf, "thr": thr}, {"o": S()})
    run("dwt level 5", {"outputs": ["o"], "processors": {"o": {"function": "discrete_wavelet_transform", "module": M, "args": ["waveform", 5, "'h'", "'a'", "o(256, 'f')"]}}}, {"waveform": wf}, {"o": DeviceArray((rows, 256), np.float32)})
    run("pickoff l", {"outputs": ["o"], "processors": {"o": f"{M}.fixed_time_pickoff(waveform, 4000.5, 'l', o)"}}, {"waveform": wf}, {"o": S()})
    run("mean_below", {"outputs": ["o"], "processors": {"o": f"{M}.mean_below_threshold(waveform, 100000, o)"}}, {"waveform": wf}, {"o": S()})
