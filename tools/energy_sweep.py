#!/usr/bin/env python3
"""Throughput of the specialised energy kernel across its parameter space (lengths, shaping filters, pick-off modes, lags, input types):
no configuration should fall off a cliff.  One line per configuration; device-resident synthetic rows."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dspeed_amd import _lib
from dspeed_amd.chain import Chain, energy_chain_program, Program, Scalar
from dspeed_amd.device import DeviceArray, Event, Stream, sync

def synth(rows, wf_len, dtype, st):
    wf = DeviceArray((rows, wf_len), dtype); bl = DeviceArray((rows,), np.float32); tp = DeviceArray((rows,), np.float32)
    code = _lib.I16 if np.dtype(dtype) == np.int16 else _lib.F32
    lo, hi = (-3000.0, 3000.0) if code == _lib.I16 else (9000.0, 11000.0)
    _lib.check(_lib.lib().dsp_synth_waveforms(wf.ptr, code, rows, wf_len, wf_len, bl.ptr, tp.ptr, 7, 0, 1716.28, 5.0, 0.2 * wf_len, lo, hi, 500.0, 15000.0, st.ptr))
    sync(); return wf, bl, tp

def main():
    total = 1 << 32  # bytes of float32 waveform data per configuration
    st = Stream()
    for wf_len in (4096, 2048, 1024):
        for dtype in (np.float32, np.int16):
            rows = total // (wf_len * 4)
            wf, bl, tp = synth(rows, wf_len, dtype, st)
            out = DeviceArray((rows,), np.float32)
            for trap in ("trap_filter", "trap_norm"):
                for mode in ("l", "h", "n"):
                    for rise, flat in ((625 * wf_len // 4096, 188 * wf_len // 4096), (wf_len // 64, 3), (wf_len // 3, wf_len // 4)):
                        if dtype == np.int16 and (trap, mode) != ("trap_filter", "l"):
                            continue
                        ch = Chain(energy_chain_program(wf_len, 1716.28, rise, flat, mode, wf_dtype=dtype, trap=trap), "sweep")
                        bufs = {"waveform": wf, "baseline": bl, "t_pick": tp, "trapEftp": out}
                        ch.execute(bufs, rows, st); ch.check(st)
                        e0, e1 = Event(), Event(); e0.record(st)
                        for _ in range(5): ch.execute(bufs, rows, st)
                        e1.record(st); sync()
                        dt = e0.elapsed_ms(e1) * 1e-3 / 5
                        print(json.dumps({"wf_len": wf_len, "dtype": np.dtype(dtype).name, "trap": trap, "mode": mode, "rise": rise, "flat": flat,
                                          "kernel": ch.kernel_name, "M_wf_per_s": round(rows / dt / 1e6, 1),
                                          "GBps_of_samples": round(rows * wf_len * np.dtype(dtype).itemsize / dt / 1e9)}), flush=True)
            del wf
main()
