"""ctypes front-end of the CPU oracle (oracle/dsp_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package -- as the checker and the reported CPU baseline, never as part of the product path.
``dspeed_amd`` does not import it.

Every wrapper takes/returns NumPy arrays shaped like the reference gufunc arguments
(rows = waveforms) and returns ``(outputs..., rc)`` where ``rc`` is 0 or the ORC_E_* code of the
DSPFatal the reference would raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("DSP_ORACLE_LIB") or os.path.join(_HERE, "libdsp_oracle.so")  # (override: the sanitizer build, tests/test_oracle_sanitizers.py)
_lib = None

E_NAMES = {
    0: "OK", 1: "PZ_NAN", 2: "DPZ_SHORT", 3: "TRAP_RISE", 4: "TRAP_FLAT", 5: "TRAP_FALL", 6: "TRAP_WIDE", 7: "FTP_INT",
    8: "FTP_MODE", 9: "TPT_START_INT", 10: "TPT_WALK_INT", 11: "TPT_RANGE", 12: "CONV_LONG", 13: "CONV_OUTLEN",
    14: "CONV_MODE", 15: "DWT_LEVEL", 16: "DWT_OUTLEN", 17: "ZERODIV",
}


def build(force: bool = False) -> str:
    if os.environ.get("DSP_ORACLE_LIB"):
        return _LIB_PATH
    src = [os.path.join(_HERE, f) for f in ("dsp_oracle.c", "dsp_oracle_impl.h", "dsp_oracle.h")]
    if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def _sfx(dt):
    dt = np.dtype(dt)
    if dt == np.float32:
        return "f32", C.c_float
    if dt == np.float64:
        return "f64", C.c_double
    raise TypeError(f"oracle supports float32/float64 loops, got {dt}")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _rows(w):
    w = np.ascontiguousarray(w)
    if w.ndim == 1:
        w = w[None, :]
    return w


def _vec(v, n, dt):
    """Per-row vector or broadcast constant -> (array, stride)."""
    a = np.asarray(v, dtype=dt)
    if a.ndim == 0:
        return a.reshape(1).copy(), 0
    a = np.ascontiguousarray(a)
    assert a.shape == (n,)
    return a, 1


def _call(name, dt, *args):
    sfx, _ = _sfx(dt)
    fn = getattr(lib(), f"orc_{name}_{sfx}")
    fn.restype = C.c_int
    err_row = C.c_long(-1)
    rc = fn(*args, C.byref(err_row))
    return rc


def bl_subtract(w, baseline):
    w = _rows(w)
    dt = w.dtype
    bl, st = _vec(baseline, w.shape[0], dt)
    out = np.empty_like(w)
    rc = _call("bl_subtract", dt, _p(w), C.c_long(w.shape[0]), C.c_int(w.shape[1]), _p(bl), C.c_int(st), _p(out))
    return out, rc


def pole_zero(w, tau):
    w = _rows(w)
    sfx, ct = _sfx(w.dtype)
    out = np.empty_like(w)
    rc = _call("pole_zero", w.dtype, _p(w), C.c_long(w.shape[0]), C.c_int(w.shape[1]), ct(float(w.dtype.type(tau))), _p(out))
    return out, rc


def double_pole_zero(w, tau1, tau2, frac):
    w = _rows(w)
    sfx, ct = _sfx(w.dtype)
    f = lambda v: ct(float(w.dtype.type(v)))  # noqa: E731
    out = np.empty_like(w)
    rc = _call("double_pole_zero", w.dtype, _p(w), C.c_long(w.shape[0]), C.c_int(w.shape[1]), f(tau1), f(tau2), f(frac), _p(out))
    return out, rc


def _trap(name, w, *ints):
    w = _rows(w)
    out = np.empty_like(w)
    rc = _call(name, w.dtype, _p(w), C.c_long(w.shape[0]), C.c_int(w.shape[1]), *[C.c_int(int(v)) for v in ints], _p(out))
    return out, rc


def trap_filter(w, rise, flat):
    return _trap("trap_filter", w, rise, flat)


def trap_norm(w, rise, flat):
    return _trap("trap_norm", w, rise, flat)


def asym_trap_filter(w, rise, flat, fall):
    return _trap("asym_trap_filter", w, rise, flat, fall)


def fixed_time_pickoff(w, t_in, mode):
    w = _rows(w)
    t, st = _vec(t_in, w.shape[0], w.dtype)
    out = np.empty(w.shape[0], dtype=w.dtype)
    m = ord(mode) if isinstance(mode, str) else int(mode)
    rc = _call("fixed_time_pickoff", w.dtype, _p(w), C.c_long(w.shape[0]), C.c_int(w.shape[1]), _p(t), C.c_int(st), C.c_int(m), _p(out))
    return out, rc


def time_point_thresh(w, a_threshold, t_start, walk_forward):
    w = _rows(w)
    sfx, ct = _sfx(w.dtype)
    thr, s1 = _vec(a_threshold, w.shape[0], w.dtype)
    ts, s2 = _vec(t_start, w.shape[0], w.dtype)
    out = np.empty(w.shape[0], dtype=w.dtype)
    rc = _call("time_point_thresh", w.dtype, _p(w), C.c_long(w.shape[0]), C.c_int(w.shape[1]), _p(thr), C.c_int(s1), _p(ts),
               C.c_int(s2), ct(float(walk_forward)), _p(out))
    return out, rc


def interpolated_time_point_thresh(w, a_threshold, t_start, walk_forward, mode):
    w = _rows(w)
    thr, s1 = _vec(a_threshold, w.shape[0], w.dtype)
    ts, s2 = _vec(t_start, w.shape[0], w.dtype)
    out = np.empty(w.shape[0], dtype=w.dtype)
    m = ord(mode) if isinstance(mode, str) else int(mode)
    rc = _call("interpolated_time_point_thresh", w.dtype, _p(w), C.c_long(w.shape[0]), C.c_int(w.shape[1]), _p(thr), C.c_int(s1), _p(ts),
               C.c_int(s2), C.c_long(int(walk_forward)), C.c_int(m), _p(out))
    return out, rc


def min_max(w):
    w = _rows(w)
    o = [np.empty(w.shape[0], dtype=w.dtype) for _ in range(4)]
    rc = _call("min_max", w.dtype, _p(w), C.c_long(w.shape[0]), C.c_int(w.shape[1]), *[_p(x) for x in o])
    return (*o, rc)


def min_max_norm(w, a_min, a_max):
    w = _rows(w)
    lo, s1 = _vec(a_min, w.shape[0], w.dtype)
    hi, s2 = _vec(a_max, w.shape[0], w.dtype)
    out = np.empty_like(w)
    rc = _call("min_max_norm", w.dtype, _p(w), C.c_long(w.shape[0]), C.c_int(w.shape[1]), _p(lo), C.c_int(s1), _p(hi), C.c_int(s2), _p(out))
    return out, rc


def windower(w, t0, out_len):
    w = _rows(w)
    t, st = _vec(t0, w.shape[0], w.dtype)
    out = np.empty((w.shape[0], int(out_len)), dtype=w.dtype)
    rc = _call("windower", w.dtype, _p(w), C.c_long(w.shape[0]), C.c_int(w.shape[1]), _p(t), C.c_int(st), _p(out), C.c_int(int(out_len)))
    return out, rc


def avg_current(w, length, out_len=None):
    w = _rows(w)
    _, ct = _sfx(w.dtype)
    m = w.shape[1] - int(length) if out_len is None else int(out_len)
    out = np.empty((w.shape[0], max(m, 0)), dtype=w.dtype)
    rc = _call("avg_current", w.dtype, _p(w), C.c_long(w.shape[0]), C.c_int(w.shape[1]), ct(float(length)), _p(out), C.c_int(m))
    return out, rc


def trap_pickoff(w, rise, flat, t_pickoff):
    w = _rows(w)
    t, st = _vec(t_pickoff, w.shape[0], w.dtype)
    out = np.empty(w.shape[0], dtype=w.dtype)
    rc = _call("trap_pickoff", w.dtype, _p(w), C.c_long(w.shape[0]), C.c_int(w.shape[1]), C.c_int(int(rise)), C.c_int(int(flat)), _p(t),
               C.c_int(st), _p(out))
    return out, rc


def upsampler(w, upsample, out_len):
    w = _rows(w)
    _, ct = _sfx(w.dtype)
    out = np.empty((w.shape[0], int(out_len)), dtype=w.dtype)
    rc = _call("upsampler", w.dtype, _p(w), C.c_long(w.shape[0]), C.c_int(w.shape[1]), ct(float(upsample)), _p(out), C.c_int(int(out_len)))
    return out, rc


def moving_window_multi(w, length, num_mw, mw_type):
    w = _rows(w)
    _, ct = _sfx(w.dtype)
    out = np.empty_like(w)
    rc = _call("moving_window_multi", w.dtype, _p(w), C.c_long(w.shape[0]), C.c_int(w.shape[1]), ct(float(length)), ct(float(num_mw)),
               C.c_int(int(mw_type)), _p(out))
    return out, rc


def linear_slope_fit(w):
    """PARITY UNPINNED (see dsp_oracle_impl.h): numba's typing of this body is restated from its rules, not executed"""
    w = _rows(w)
    o = [np.empty(w.shape[0], dtype=w.dtype) for _ in range(4)]
    rc = _call("linear_slope_fit", w.dtype, _p(w), C.c_long(w.shape[0]), C.c_int(w.shape[1]), *[_p(x) for x in o])
    return (*o, rc)


def mean_below_threshold(w, threshold):
    w = _rows(w)
    thr, st = _vec(threshold, w.shape[0], w.dtype)
    out = np.empty(w.shape[0], dtype=w.dtype)
    rc = _call("mean_below_threshold", w.dtype, _p(w), C.c_long(w.shape[0]), C.c_int(w.shape[1]), _p(thr), C.c_int(st), _p(out))
    return out, rc


def convolve_wf(w, kernel, mode, out_len, in_len=None):
    """``in_len``: use only the first in_len samples of each row (the reference's ``wf[:in_len]`` slice view)."""
    w = _rows(w)
    k = np.ascontiguousarray(kernel, dtype=w.dtype)
    n = w.shape[1] if in_len is None else int(in_len)
    out = np.empty((w.shape[0], out_len), dtype=w.dtype)
    m = ord(mode) if isinstance(mode, str) else int(mode)
    rc = _call("convolve", w.dtype, _p(w), C.c_long(w.shape[0]), C.c_int(n), C.c_long(w.shape[1]), _p(k), C.c_int(len(k)), C.c_int(m),
               _p(out), C.c_int(out_len))
    return out, rc


def dwt_haar(w, level, part, out_len):
    w = _rows(w)
    out = np.empty((w.shape[0], out_len), dtype=w.dtype)
    pc = ord(part) if isinstance(part, str) else int(part)
    rc = _call("dwt_haar", w.dtype, _p(w), C.c_long(w.shape[0]), C.c_int(w.shape[1]), C.c_int(int(level)), C.c_int(pc), _p(out),
               C.c_int(out_len))
    return out, rc


def chain_energy(wf, baseline, t_pick, tau, rise, flat, mode="l", block_width=16, n_threads=1):
    """bl_subtract -> pole_zero -> trap_filter -> fixed_time_pickoff in 16-row blocks (config C2/C4)."""
    wf = np.ascontiguousarray(wf, dtype=np.float32)
    bl = np.ascontiguousarray(baseline, dtype=np.float32)
    tp = np.ascontiguousarray(t_pick, dtype=np.float32)
    out = np.empty(wf.shape[0], dtype=np.float32)
    fn = lib().orc_chain_energy_f32
    fn.restype = C.c_int
    rc = fn(_p(wf), C.c_long(wf.shape[0]), C.c_int(wf.shape[1]), _p(bl), _p(tp), C.c_float(tau), C.c_int(rise), C.c_int(flat),
            C.c_int(ord(mode)), _p(out), C.c_int(block_width), C.c_int(n_threads))
    return out, rc


def chain_pz_trap(wf, tau, rise, flat, block_width=16, n_threads=1):
    wf = np.ascontiguousarray(wf, dtype=np.float32)
    out = np.empty_like(wf)
    fn = lib().orc_chain_pz_trap_f32
    fn.restype = C.c_int
    rc = fn(_p(wf), C.c_long(wf.shape[0]), C.c_int(wf.shape[1]), C.c_float(tau), C.c_int(rise), C.c_int(flat), _p(out),
            C.c_int(block_width), C.c_int(n_threads))
    return out, rc


def max_threads() -> int:
    fn = lib().orc_max_threads
    fn.restype = C.c_int
    return fn()
