"""ctypes binding of libdspeed_hip.so (include/dspeed_hip.h).

This is the whole Python <-> device boundary: plain pointers and sizes.  The library is REQUIRED:
there is no CPU fallback anywhere in dspeed_amd -- if the shared object is missing or a HIP call fails,
the caller gets an exception.
"""
from __future__ import annotations

import ctypes as C
import os

from .errors import DSPFatal

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSPEED_HIP_LIB") or os.path.join(_HERE, "libdspeed_hip.so")  # override: A/B kernel builds

# ---- constants of include/dspeed_hip.h
OK, ERR_HIP, ERR_ARG, ERR_UNSUPPORTED, ERR_TOO_LONG = 0, -1, -2, -3, -4
E_ZERODIV = 17
F32, F64, I16, U16, I32, U32, BOOL, I64, U64 = range(9)
IO_WF_IN, IO_WF_OUT, IO_SCALAR_IN, IO_SCALAR_OUT, IO_TAPS = range(5)
ARG_CONST, ARG_INPUT, ARG_REG = range(3)
(OP_LOAD, OP_STORE, OP_STORE_SCALAR, OP_BL_SUBTRACT, OP_POLE_ZERO, OP_DOUBLE_POLE_ZERO, OP_TRAP_FILTER, OP_TRAP_NORM,
 OP_ASYM_TRAP, OP_PICKOFF, OP_TIME_POINT_THRESH, OP_MIN_MAX, OP_DWT_HAAR, OP_CONVOLVE, OP_COPY, OP_TRAP_PICKOFF, OP_AMAX,
 OP_SCALAR_AFFINE, OP_MEAN_BELOW, OP_CONVOLVE_AMAX, OP_WINDOWER, OP_AVG_CURRENT, OP_TRAP_WINDOW_PICKOFF, OP_TRAP_REDUCE, OP_UPSAMPLER, OP_MOVING_WINDOW_MULTI, OP_LINEAR_SLOPE_FIT,
 OP_SCALAR_CONVERT, OP_SCALAR_DIV, OP_INTERP_TIME_POINT_THRESH, OP_MIN_MAX_NORM, OP_ELEMENTWISE, OP_SCALAR_FUNC) = range(1, 34)
(FN_ADD, FN_SUB, FN_MUL, FN_DIV, FN_LT, FN_LE, FN_GT, FN_GE, FN_EQ, FN_NE, FN_WHERE, FN_ISNAN, FN_ISFINITE, FN_NEG, FN_COPY, FN_FLOORDIV,
 FN_IADD, FN_ISUB, FN_IMUL, FN_IFLOORDIV, FN_ICAST, FN_LOR, FN_LAND, FN_RINT, FN_FLOOR, FN_CEIL, FN_TRUNC) = range(27)


def fn_int(code, dtype):
    """ip[0] of an integer loop: DSP_FN_I* | DSP_FN_INT(bits, signed)  (dspeed_hip.h)"""
    import numpy as np

    dtype = np.dtype(dtype)
    return code | (dtype.itemsize * 8) << 8 | (1 << 16 if dtype.kind == "i" else 0)


MAX_OPS, MAX_SLOTS, MAX_IO, MAX_SREGS = 192, 32, 128, 128


class IoDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dtype", C.c_int32), ("len", C.c_int32), ("offset", C.c_int32), ("row_stride", C.c_int64)]


class ScalarArg(C.Structure):
    _fields_ = [("kind", C.c_int32), ("index", C.c_int32), ("value", C.c_double)]


class FitWindow(C.Structure):  # dsp_fit_window
    _fields_ = [("stage", C.c_int32), ("first", C.c_int32), ("count", C.c_int32)]


FIT_MAX = 4


class Op(C.Structure):
    _fields_ = [("opcode", C.c_int32), ("dst", C.c_int32), ("src", C.c_int32), ("io", C.c_int32), ("ip", C.c_int32 * 4),
                ("sp", ScalarArg * 4)]


class PlanInfo(C.Structure):  # dsp_plan_info
    _fields_ = [("kernel", C.c_char * 64), ("note", C.c_char * 256), ("lds_bytes_per_wave", C.c_int32), ("waves_per_block", C.c_int32),
                ("team", C.c_int32), ("n_device_ops", C.c_int32), ("lds_elems_per_wave", C.c_int32), ("sreg_off", C.c_int32),
                ("scratch_off", C.c_int32), ("n_slots", C.c_int32), ("slot_base", C.c_int32 * MAX_SLOTS), ("slot_elems", C.c_int32 * MAX_SLOTS),
                ("slot_first_op", C.c_int32 * MAX_SLOTS), ("slot_last_op", C.c_int32 * MAX_SLOTS), ("slot_off", C.c_int32 * MAX_SLOTS),
                ("slot_pitch", C.c_int32 * MAX_SLOTS), ("slot_chunk", C.c_int32 * MAX_SLOTS)]


_lib = None


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m dspeed_amd.build` (hipcc, gfx950). "
                           "dspeed_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    pi64 = C.POINTER(C.c_int64)
    sig = {
        "dsp_device_count": [C.POINTER(C.c_int)],
        "dsp_set_device": [C.c_int],
        "dsp_get_device": [C.POINTER(C.c_int)],
        "dsp_device_info": [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), pi64, C.POINTER(C.c_int)],
        "dsp_malloc": [C.POINTER(vp), i64],
        "dsp_free": [vp],
        "dsp_host_alloc": [C.POINTER(vp), i64],
        "dsp_host_free": [vp],
        "dsp_memset": [vp, C.c_int, i64, vp],
        "dsp_h2d": [vp, vp, i64],
        "dsp_d2h": [vp, vp, i64],
        "dsp_h2d_async": [vp, vp, i64, vp],
        "dsp_d2h_async": [vp, vp, i64, vp],
        "dsp_stream_create": [C.POINTER(vp)],
        "dsp_stream_destroy": [vp],
        "dsp_stream_sync": [vp],
        "dsp_sync": [],
        "dsp_event_create": [C.POINTER(vp)],
        "dsp_event_destroy": [vp],
        "dsp_event_record": [vp, vp],
        "dsp_stream_wait_event": [vp, vp],
        "dsp_host_register": [vp, i64],
        "dsp_host_unregister": [vp],
        "dsp_event_sync": [vp],
        "dsp_event_elapsed_ms": [vp, vp, C.POINTER(C.c_float)],
        "dsp_chain_create": [C.POINTER(Op), C.c_int, C.POINTER(IoDesc), C.c_int, C.POINTER(i32), C.c_int, C.c_int, C.c_int,
                             C.POINTER(vp)],
        "dsp_chain_plan": [C.POINTER(Op), C.c_int, C.POINTER(IoDesc), C.c_int, C.POINTER(i32), C.c_int, C.c_int, C.c_int, C.POINTER(PlanInfo)],
        "dsp_chain_execute": [vp, C.POINTER(vp), i64, vp],
        "dsp_chain_check": [vp, vp, pi64],
        "dsp_chain_destroy": [vp],
        "dsp_chain_geometry": [vp, i64, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "dsp_chain_set_fused": [vp, C.c_int],
        "dsp_chain_set_async_check": [vp, C.c_int],
        "dsp_bl_subtract_f32": [vp, C.c_int, i64, i32, i64, vp, f32, vp, i64, vp, pi64],
        "dsp_pole_zero_f32": [vp, C.c_int, i64, i32, i64, f32, vp, i64, vp, pi64],
        "dsp_double_pole_zero_f32": [vp, C.c_int, i64, i32, i64, f32, f32, f32, vp, i64, vp, pi64],
        "dsp_pole_zero_col_f32": [vp, C.c_int, i64, i32, i64, vp, f32, vp, i64, vp, pi64],
        "dsp_double_pole_zero_col_f32": [vp, C.c_int, i64, i32, i64, vp, f32, vp, f32, vp, f32, vp, i64, vp, pi64],
        "dsp_trap_filter_f32": [vp, C.c_int, i64, i32, i64, i32, i32, vp, i64, vp, pi64],
        "dsp_trap_norm_f32": [vp, C.c_int, i64, i32, i64, i32, i32, vp, i64, vp, pi64],
        "dsp_asym_trap_filter_f32": [vp, C.c_int, i64, i32, i64, i32, i32, i32, vp, i64, vp, pi64],
        "dsp_fixed_time_pickoff_f32": [vp, C.c_int, i64, i32, i64, vp, f32, i32, vp, vp, pi64],
        "dsp_min_max_norm_f32": [vp, C.c_int, i64, i32, i64, vp, f32, vp, f32, vp, i64, vp, pi64],
        "dsp_install_abort_trace": [C.c_int],
        "dsp_uninstall_abort_trace": [],
        "dsp_chain_profile": [vp, C.c_int],
        "dsp_chain_profile_read": [vp, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.POINTER(C.c_int),
                                   C.POINTER(C.c_uint64)],
        "dsp_time_point_thresh_f32": [vp, C.c_int, i64, i32, i64, vp, f32, vp, f32, f32, vp, vp, pi64],
        "dsp_interpolated_time_point_thresh_f32": [vp, C.c_int, i64, i32, i64, vp, f32, vp, f32, i64, i32, vp, vp, pi64],
        "dsp_min_max_f32": [vp, C.c_int, i64, i32, i64, vp, vp, vp, vp, vp, pi64],
        "dsp_mean_below_threshold_f32": [vp, C.c_int, i64, i32, i64, vp, f32, vp, vp, pi64],
        "dsp_linear_slope_fit_f32": [vp, C.c_int, i64, i32, i64, vp, vp, vp, vp, vp, pi64],
        "dsp_upsampler_f32": [vp, C.c_int, i64, i32, i64, f32, vp, i32, i64, vp, pi64],
        "dsp_moving_window_multi_f32": [vp, C.c_int, i64, i32, i64, f32, f32, i32, vp, i64, vp, pi64],
        "dsp_windower_f32": [vp, C.c_int, i64, i32, i64, vp, f32, vp, i32, i64, vp, pi64],
        "dsp_avg_current_f32": [vp, C.c_int, i64, i32, i64, f32, vp, i32, i64, vp, pi64],
        "dsp_trap_pickoff_f32": [vp, C.c_int, i64, i32, i64, i32, i32, vp, f32, vp, vp, pi64],
        "dsp_dwt_haar_f32": [vp, C.c_int, i64, i32, i64, i32, i32, vp, i32, i64, vp, pi64],
        "dsp_convolve_wf_f32": [vp, C.c_int, i64, i32, i64, vp, i32, i32, vp, i32, i64, vp, pi64],
        "dsp_synth_waveforms": [vp, C.c_int, i64, i32, i64, vp, vp, C.c_uint64, i64, f32, f32, f32, f32, f32, f32, f32, vp],
        "dsp_synth_pulses": [vp, C.c_int, i64, i32, i64, vp, vp, C.c_uint64, i64, f32, f32, f32, f32, f32, f32, f32, f32, f32, vp],
        "dsp_stream_read": [vp, i64, vp, vp],
        "dsp_linear_slope_fit_rows": [vp, C.c_int, i64, i32, i64, C.c_int, vp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double,
                                      C.POINTER(FitWindow), C.c_int, vp, vp],
    }
    f64 = C.c_double
    for name in list(sig):  # the float64 loops: same argument order, double scalars
        if name.endswith("_f32") and name not in ("dsp_synth_waveforms", "dsp_synth_pulses"):
            sig[name[:-4] + "_f64"] = [f64 if t is f32 else t for t in sig[name]]
    for name, argtypes in sig.items():
        fn = getattr(L, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    for name in ("dsp_last_error", "dsp_version"):
        getattr(L, name).restype = C.c_char_p
        getattr(L, name).argtypes = []
    L.dsp_fatal_message.restype = C.c_char_p
    L.dsp_fatal_message.argtypes = [C.c_int]
    L.dsp_chain_kernel_name.restype = C.c_char_p
    L.dsp_chain_kernel_name.argtypes = [vp]
    L.dsp_chain_kernel_note.restype = C.c_char_p
    L.dsp_chain_kernel_note.argtypes = [vp]
    L.dsp_chain_share_row_scales.restype = C.c_int
    L.dsp_chain_share_row_scales.argtypes = [vp, vp]
    _lib = L
    return L


EXPORTS = [
    "dsp_device_count", "dsp_set_device", "dsp_get_device", "dsp_device_info", "dsp_malloc", "dsp_free", "dsp_host_alloc", "dsp_host_register", "dsp_host_unregister", "dsp_stream_wait_event",
    "dsp_host_free", "dsp_memset", "dsp_h2d", "dsp_d2h", "dsp_h2d_async", "dsp_d2h_async", "dsp_stream_create", "dsp_stream_destroy",
    "dsp_stream_sync", "dsp_sync", "dsp_event_create", "dsp_event_destroy", "dsp_event_record", "dsp_event_sync",
    "dsp_event_elapsed_ms", "dsp_last_error", "dsp_fatal_message", "dsp_version", "dsp_chain_create", "dsp_chain_plan", "dsp_chain_execute",
    "dsp_chain_check", "dsp_chain_destroy", "dsp_chain_geometry", "dsp_chain_kernel_name", "dsp_chain_kernel_note", "dsp_chain_share_row_scales", "dsp_chain_set_fused", "dsp_chain_set_async_check", "dsp_bl_subtract_f32", "dsp_pole_zero_f32",
    "dsp_double_pole_zero_f32", "dsp_pole_zero_col_f32", "dsp_double_pole_zero_col_f32", "dsp_pole_zero_col_f64", "dsp_double_pole_zero_col_f64", "dsp_trap_filter_f32", "dsp_trap_norm_f32", "dsp_asym_trap_filter_f32", "dsp_fixed_time_pickoff_f32",
    "dsp_install_abort_trace", "dsp_uninstall_abort_trace", "dsp_chain_profile", "dsp_chain_profile_read", "dsp_min_max_norm_f32", "dsp_min_max_norm_f64", "dsp_time_point_thresh_f32", "dsp_interpolated_time_point_thresh_f32", "dsp_interpolated_time_point_thresh_f64", "dsp_min_max_f32", "dsp_mean_below_threshold_f32", "dsp_mean_below_threshold_f64", "dsp_windower_f32", "dsp_windower_f64", "dsp_avg_current_f32",
    "dsp_avg_current_f64", "dsp_trap_pickoff_f32", "dsp_trap_pickoff_f64", "dsp_upsampler_f32", "dsp_upsampler_f64",
    "dsp_moving_window_multi_f32", "dsp_moving_window_multi_f64", "dsp_linear_slope_fit_f32", "dsp_linear_slope_fit_f64", "dsp_dwt_haar_f32", "dsp_convolve_wf_f32", "dsp_synth_waveforms", "dsp_synth_pulses", "dsp_stream_read",
    "dsp_bl_subtract_f64", "dsp_pole_zero_f64", "dsp_double_pole_zero_f64", "dsp_trap_filter_f64", "dsp_trap_norm_f64",
    "dsp_asym_trap_filter_f64", "dsp_fixed_time_pickoff_f64", "dsp_time_point_thresh_f64", "dsp_min_max_f64", "dsp_dwt_haar_f64",
    "dsp_convolve_wf_f64", "dsp_linear_slope_fit_rows",
]


def last_error() -> str:
    return lib().dsp_last_error().decode()


def fatal_message(code: int) -> str:
    return lib().dsp_fatal_message(code).decode()


def check(rc: int, row: int | None = None, what: str = ""):
    """Map a status code to the reference's error behaviour: DSP_E_* -> DSPFatal (ZeroDivisionError for the numba
    division-by-zero case), negative -> RuntimeError / NotImplementedError / ValueError."""
    if rc == OK:
        return
    if rc == E_ZERODIV:
        raise ZeroDivisionError("division by zero")
    if rc > 0:
        msg = last_error() or fatal_message(rc)
        err = DSPFatal(msg)
        if row is not None and row >= 0:
            err.wf_range = range(row, row + 1)
        if what:
            err.processor = what
        raise err
    msg = f"{what + ': ' if what else ''}{last_error()}"
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc in (ERR_ARG, ERR_TOO_LONG):
        raise ValueError(msg)
    raise RuntimeError(msg)
