#!/usr/bin/env python3
"""Secondary measurements (NOT the contract bench, which is bench.py): the other BASELINE.json configs on device-resident
synthetic batches, each against the roofline that bounds it.  One JSON object per line.

  C2-vm   the energy chain on the generic waveform VM (what an arbitrary recipe gets)          HBM, 16 396 B / waveform
  C2-cls  ... on the classic specialised kernel                                                HBM
  C3      8192-sample float32 rows: bl_subtract -> 2 x 5792-tap FIR 'v' on wf[:6092] -> amax    FP32 FMA, 6.97 MFLOP / waveform
  C5      8192-sample int16 rows: double_pole_zero -> asym_trap -> min_max -> time_point_thresh, Haar DWT level 5
                                                                                                HBM, 16 384 B read + 1 044 B written
Usage (GPU box): python tools/bench_configs.py [rows]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import recipes  # noqa: E402
from dspeed_amd import _lib  # noqa: E402
from dspeed_amd.device import DeviceArray, Event, Stream, sync  # noqa: E402
from dspeed_amd.processing_chain import build_processing_chain  # noqa: E402

HBM_PEAK, FMA_PEAK = 8000.0, 157.3  # GB/s, TFLOP/s (MI355X_MICROARCH.md)
TAU, SIGMA, SEED = 1716.28, 5.0, 0xD5BEED


def synth(rows, wf_len, dtype, stream, bl_lo=9000.0, bl_hi=11000.0, rise=None):
    """rise=(lo, hi): pulses that take lo .. hi samples to reach their height (dsp_synth_pulses) instead of one-sample steps"""
    wf = DeviceArray((rows, wf_len), dtype)
    bl, tp = DeviceArray((rows,), np.float32), DeviceArray((rows,), np.float32)
    code = _lib.I16 if np.dtype(dtype) == np.int16 else _lib.F32
    if rise is not None:
        _lib.check(_lib.lib().dsp_synth_pulses(wf.ptr, code, rows, wf_len, wf_len, bl.ptr, tp.ptr, SEED, 0, TAU, SIGMA, 625 + 0.8 * 188,
                                               bl_lo, bl_hi, 500.0, 15000.0, float(rise[0]), float(rise[1]), stream.ptr), what="synth")
        return wf, bl, tp
    _lib.check(_lib.lib().dsp_synth_waveforms(wf.ptr, code, rows, wf_len, wf_len, bl.ptr, tp.ptr, SEED, 0, TAU, SIGMA, 625 + 0.8 * 188,
                                              bl_lo, bl_hi, 500.0, 15000.0, stream.ptr), what="synth")
    return wf, bl, tp


def timed(chain, steps=5, warmup=2):
    st = chain._stream
    for _ in range(warmup):
        chain.execute()
    e0, e1 = Event(), Event()
    e0.record(st)
    for _ in range(steps):
        chain.execute()
    e1.record(st)
    sync()
    return e0.elapsed_ms(e1) * 1e-3 / steps


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    st = Stream()
    out = []
    # ---- C2 on the VM and on the classic kernel
    wf, bl, tp = synth(rows, 4096, np.float32, st)
    sync()
    for label, fused in (("C2-vm", 0), ("C2-cls", 15), ("C2-default", 1)):
        tb = {"waveform": wf, "baseline": bl, "t_pick": tp}
        chain, _, _ = build_processing_chain(recipes.C2, tb)
        chain.link(tb, {"trapEftp": DeviceArray((rows,), np.float32)})
        chain._ensure()
        chain._chain.set_fused(fused)
        dt = timed(chain)
        gbps = rows * 16396 / dt / 1e9
        out.append({"config": label, "kernel": chain._chain.kernel_name, "rows": rows, "waveforms_per_s": rows / dt, "bound": "hbm",
                    "achieved_GBps": gbps, "frac": gbps / HBM_PEAK})
    # ---- C2 with the pole-zero time constant as a per-event column (the register-resident kernel's TAU build; the interpreter before)
    import copy
    rec_tau = copy.deepcopy(recipes.C2)
    rec_tau["processors"]["wf_pz"] = "dspeed.processors.pole_zero(wf_blsub, tau, wf_pz)"
    tau = DeviceArray.from_numpy(np.full(rows, TAU, dtype=np.float32))
    tb = {"waveform": wf, "baseline": bl, "t_pick": tp, "tau": tau}
    chain, _, _ = build_processing_chain(rec_tau, tb)
    chain.link(tb, {"trapEftp": DeviceArray((rows,), np.float32)})
    dt = timed(chain)
    gbps = rows * 16400 / dt / 1e9
    out.append({"config": "C2-tau-per-event", "kernel": chain._chain.kernel_name, "rows": rows, "waveforms_per_s": rows / dt, "bound": "hbm",
                "bytes_per_waveform": 16400, "achieved_GBps": gbps, "frac": gbps / HBM_PEAK})
    del wf
    # ---- C2 on 16-bit rows (what the digitisers write): 8 kB per waveform instead of 16
    wf, bl, tp = synth(rows, 4096, np.int16, st, bl_lo=-3000.0, bl_hi=3000.0)
    sync()
    tb = {"waveform": wf, "baseline": bl, "t_pick": tp}
    chain, _, _ = build_processing_chain(recipes.C2, tb)
    chain.link(tb, {"trapEftp": DeviceArray((rows,), np.float32)})
    dt = timed(chain)
    gbps = rows * (4096 * 2 + 12) / dt / 1e9
    out.append({"config": "C2-int16", "kernel": chain._chain.kernel_name, "rows": rows, "waveforms_per_s": rows / dt, "bound": "hbm",
                "bytes_per_waveform": 4096 * 2 + 12, "achieved_GBps": gbps, "frac": gbps / HBM_PEAK})
    del wf
    # ---- C3: long FIR
    r3 = max(1000, rows // 4)
    wf, bl, tp = synth(r3, 8192, np.float32, st)
    sync()
    tb = {"waveform": wf, "baseline": bl}
    chain, _, _ = build_processing_chain(recipes.C3, tb)
    chain.link(tb, {"cuspEmax": DeviceArray((r3,), np.float32), "zacEmax": DeviceArray((r3,), np.float32)})
    dt = timed(chain, steps=3, warmup=1)
    tflops = r3 * 6.97e6 / dt / 1e12
    if "f16" in chain._chain.kernel_name:  # three float16 products per multiply-add over the padded 64 x 320 x 6144 tiles of the two kernels
        issued = r3 * 3 * 2 * 6144 * 320 * 2 / dt / 1e12
        out.append({"config": "C3", "kernel": chain._chain.kernel_name, "rows": r3, "waveforms_per_s": r3 / dt, "bound": "f16-mfma",
                    "achieved_TFLOPs": issued, "frac": issued / 2500.0, "algorithmic_TFLOPs": tflops, "achieved_GBps_read": r3 * 32768 / dt / 1e9})
    else:
        out.append({"config": "C3", "kernel": chain._chain.kernel_name, "rows": r3, "waveforms_per_s": r3 / dt, "bound": "fp32-fma",
                    "achieved_TFLOPs": tflops, "frac": tflops / FMA_PEAK, "achieved_GBps_read": r3 * 32768 / dt / 1e9})
    del wf
    # ---- C5: int16 rows
    wf, bl, tp = synth(rows, 8192, np.int16, st, bl_lo=-3000.0, bl_hi=3000.0)
    thr = DeviceArray.from_numpy(np.full(rows, 20.0, dtype=np.float32))
    sync()
    outs = {k: DeviceArray((rows,), np.float32) for k in ("tp_0", "tp_min", "tp_max", "wf_min", "wf_max")}
    outs["dwt_haar"] = DeviceArray((rows, 256), np.float32)
    tb = {"waveform": wf, "thr": thr}
    chain, _, _ = build_processing_chain(recipes.C5, tb)
    chain.link(tb, outs)
    dt = timed(chain)
    bytes_wf = 8192 * 2 + 4 + 5 * 4 + 256 * 4
    gbps = rows * bytes_wf / dt / 1e9
    out.append({"config": "C5", "kernel": chain._chain.kernel_name, "rows": rows, "waveforms_per_s": rows / dt, "bound": "hbm",
                "bytes_per_waveform": bytes_wf, "achieved_GBps": gbps, "frac": gbps / HBM_PEAK})
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
