"""Processor registry: the same names as ``dspeed.processors`` (reference processors/__init__.py:66-172) for the
hot-path processors, each a :class:`~dspeed_amd.gufunc.HipGUFunc` with the reference's gufunc layout and type
strings, executed by hand-written HIP kernels through the C ABI.

A dspeed JSON recipe that says ``"module": "dspeed.processors"`` resolves to this module in
``dspeed_amd.processing_chain`` (the reference's own Python never runs on the GPU box).
"""
from __future__ import annotations

import numpy as np

from ..device import DeviceArray
from ..errors import DSPFatal
from ..gufunc import HipGUFunc, Staging, entry, loop_suffix, run

_F = {"f32": np.float32, "f64": np.float64}


def _dtype_of(x):
    return x.dtype if hasattr(x, "dtype") else np.asarray(x).dtype


def _split(g, args):
    """Split positional arguments into inputs and (optional) in-place outputs, as NumPy gufuncs do."""
    if len(args) == g.nargs:
        return list(args[: g.nin]), list(args[g.nin:])
    if len(args) == g.nin:
        return list(args), [None] * g.nout
    raise TypeError(f"{g.__name__}() takes {g.nin} inputs and {g.nout} outputs ({len(args)} given)")


def _result(outs, squeeze):
    res = []
    for o in outs:
        if isinstance(o, np.ndarray) and squeeze:
            o = o.reshape(o.shape[1:]) if o.ndim >= 1 and o.shape[0] == 1 else o
            if o.ndim == 0:
                o = o[()]
        res.append(o)
    return res[0] if len(res) == 1 else tuple(res)


# --------------------------------------------------------------------------------------------------- waveform -> waveform
def _wf2wf(name, n_scalars=0, n_ints=0, scalar_cols=()):
    """Implementation factory for '(n),<scalars>->(n)' processors."""

    def impl(g, *args):
        ins, outs = _split(g, args)
        st = Staging()
        try:
            ptr, code, n_wf, n, stride, one_d = st.wf_in(ins[0])
            sfx = loop_suffix(_dtype_of(ins[0]))
            fn = entry(name, sfx)
            ft = _F[sfx]
            cargs = []
            for k, v in enumerate(ins[1:]):
                if k in scalar_cols:
                    p, val = st.scalar_in(v, n_wf, ft)
                    cargs += [p, val]
                elif k < n_ints:
                    cargs.append(_as_int(v, g.__name__))
                else:
                    cargs.append(float(ft(v)))
            shape = (n,) if one_d else (n_wf, n)
            optr, res = st.out(outs[0], shape, ft)
            run(fn, g.__name__, ptr, code, n_wf, n, stride, *cargs, optr, n)
            st.finish()
            return res
        finally:
            st.release()

    return impl


def _as_int(v, what):
    a = np.asarray(v)
    if a.size != 1:
        raise NotImplementedError(f"{what}: per-waveform integer parameters are not supported on the device")
    f = float(a.reshape(-1)[0])
    if np.isnan(f):
        raise NotImplementedError(f"{what}: NaN integer parameter")
    return int(f)


bl_subtract = HipGUFunc("bl_subtract", "(n),()->(n)", ["ff->f", "dd->d"], _wf2wf("bl_subtract", scalar_cols=(0,)),
                        "w_out = w_in - a_baseline (reference processors/bl_subtract.py:11-46)")
min_max_norm = HipGUFunc("min_max_norm", "(n),(),()->(n)", ["fff->f", "ddd->d"], _wf2wf("min_max_norm", scalar_cols=(0, 1)),
                         "waveform over the larger of |a_min|, |a_max| (reference processors/min_max.py:85-140)")
pole_zero = HipGUFunc("pole_zero", "(n),()->(n)", ["ff->f", "dd->d"], _wf2wf("pole_zero_col", scalar_cols=(0,)),
                      "single pole-zero cancellation (reference processors/pole_zero.py:24-77)")
double_pole_zero = HipGUFunc("double_pole_zero", "(n),(),(),()->(n)", ["ffff->f", "dddd->d"], _wf2wf("double_pole_zero_col", scalar_cols=(0, 1, 2)),
                             "double pole-zero cancellation (reference processors/pole_zero.py:82-198)")
trap_filter = HipGUFunc("trap_filter", "(n),(),()->(n)", ["fii->f", "dii->d"], _wf2wf("trap_filter", n_ints=2),
                        "symmetric trapezoidal filter (reference processors/trap_filters.py:12-76)")
trap_norm = HipGUFunc("trap_norm", "(n),(),()->(n)", ["fii->f", "dii->d"], _wf2wf("trap_norm", n_ints=2),
                      "normalised trapezoidal filter (reference processors/trap_filters.py:79-149)")
asym_trap_filter = HipGUFunc("asym_trap_filter", "(n),(),(),()->(n)", ["fiii->f", "diii->d"], _wf2wf("asym_trap_filter", n_ints=3),
                             "asymmetric trapezoidal filter (reference processors/trap_filters.py:152-227)")


# --------------------------------------------------------------------------------------------------- waveform -> scalars
def _mode_char(m):
    if isinstance(m, str):
        return ord(m)
    a = np.asarray(m)
    if a.dtype.kind in "SU":
        return ord(str(a.reshape(-1)[0])[0])
    return int(a.reshape(-1)[0])


def _fixed_time_pickoff(g, *args):
    ins, outs = _split(g, args)
    st = Staging()
    try:
        ptr, code, n_wf, n, stride, one_d = st.wf_in(ins[0])
        sfx = loop_suffix(_dtype_of(ins[0]))
        ft = _F[sfx]
        tp, tv = st.scalar_in(ins[1], n_wf, ft)
        optr, res = st.out(outs[0], () if one_d else (n_wf,), ft)
        run(entry("fixed_time_pickoff", sfx), g.__name__, ptr, code, n_wf, n, stride, tp, tv, _mode_char(ins[2]), optr)
        st.finish()
        return res[()] if isinstance(res, np.ndarray) and res.ndim == 0 else res
    finally:
        st.release()


def _time_point_thresh(g, *args):
    ins, outs = _split(g, args)
    st = Staging()
    try:
        ptr, code, n_wf, n, stride, one_d = st.wf_in(ins[0])
        sfx = loop_suffix(_dtype_of(ins[0]))
        ft = _F[sfx]
        ap, av = st.scalar_in(ins[1], n_wf, ft)
        sp, sv = st.scalar_in(ins[2], n_wf, ft)
        walk = float(np.asarray(ins[3]).reshape(-1)[0])
        optr, res = st.out(outs[0], () if one_d else (n_wf,), ft)
        run(entry("time_point_thresh", sfx), g.__name__, ptr, code, n_wf, n, stride, ap, av, sp, sv, walk, optr)
        st.finish()
        return res[()] if isinstance(res, np.ndarray) and res.ndim == 0 else res
    finally:
        st.release()


def _interpolated_time_point_thresh(g, *args):
    ins, outs = _split(g, args)
    st = Staging()
    try:
        ptr, code, n_wf, n, stride, one_d = st.wf_in(ins[0])
        sfx = loop_suffix(_dtype_of(ins[0]))
        ft = _F[sfx]
        ap, av = st.scalar_in(ins[1], n_wf, ft)
        sp, sv = st.scalar_in(ins[2], n_wf, ft)
        walk = int(np.asarray(ins[3]).reshape(-1)[0])  # an int64 argument of the gufunc
        optr, res = st.out(outs[0], () if one_d else (n_wf,), ft)
        run(entry("interpolated_time_point_thresh", sfx), g.__name__, ptr, code, n_wf, n, stride, ap, av, sp, sv, walk, _mode_char(ins[4]), optr)
        st.finish()
        return res[()] if isinstance(res, np.ndarray) and res.ndim == 0 else res
    finally:
        st.release()


def _linear_slope_fit(g, *args):
    ins, outs = _split(g, args)
    st = Staging()
    try:
        ptr, code, n_wf, n, stride, one_d = st.wf_in(ins[0])
        sfx = loop_suffix(_dtype_of(ins[0]))
        ft = _F[sfx]
        pr = [st.out(o, () if one_d else (n_wf,), ft) for o in outs]
        run(entry("linear_slope_fit", sfx), g.__name__, ptr, code, n_wf, n, stride, *[p for p, _ in pr])
        st.finish()
        return tuple(r[()] if isinstance(r, np.ndarray) and r.ndim == 0 else r for _, r in pr)
    finally:
        st.release()


def _mean_below_threshold(g, *args):
    ins, outs = _split(g, args)
    st = Staging()
    try:
        ptr, code, n_wf, n, stride, one_d = st.wf_in(ins[0])
        sfx = loop_suffix(_dtype_of(ins[0]))
        ft = _F[sfx]
        tp, tv = st.scalar_in(ins[1], n_wf, ft)
        optr, res = st.out(outs[0], () if one_d else (n_wf,), ft)
        run(entry("mean_below_threshold", sfx), g.__name__, ptr, code, n_wf, n, stride, tp, tv, optr)
        st.finish()
        return res[()] if isinstance(res, np.ndarray) and res.ndim == 0 else res
    finally:
        st.release()


def _min_max(g, *args):
    ins, outs = _split(g, args)
    st = Staging()
    try:
        ptr, code, n_wf, n, stride, one_d = st.wf_in(ins[0])
        sfx = loop_suffix(_dtype_of(ins[0]))
        ft = _F[sfx]
        pr = [st.out(o, () if one_d else (n_wf,), ft) for o in outs]
        run(entry("min_max", sfx), g.__name__, ptr, code, n_wf, n, stride, *[p for p, _ in pr])
        st.finish()
        return tuple(r[()] if isinstance(r, np.ndarray) and r.ndim == 0 else r for _, r in pr)
    finally:
        st.release()


fixed_time_pickoff = HipGUFunc("fixed_time_pickoff", "(n),(),()->()", ["ffb->f", "ddb->d"], _fixed_time_pickoff,
                               "value at a (fractional) sample index, modes i n f c l h (reference processors/fixed_time_pickoff.py:12-125)")
time_point_thresh = HipGUFunc("time_point_thresh", "(n),(),(),()->()", ["ffff->f", "dddd->d"], _time_point_thresh,
                              "first threshold crossing walking forward/backward (reference processors/time_point_thresh.py:12-92)")
interpolated_time_point_thresh = HipGUFunc("interpolated_time_point_thresh", "(n),(),(),(),()->()", ["ffflb->f", "dddlb->d"], _interpolated_time_point_thresh,
                                           "threshold crossing placed between samples, modes i b c a f r n l (reference processors/time_point_thresh.py:95-222)")
linear_slope_fit = HipGUFunc("linear_slope_fit", "(n)->(),(),(),()", ["f->ffff", "d->dddd"], _linear_slope_fit,
                             "Welford mean / standard deviation and least-squares slope / intercept (reference processors/linear_slope_fit.py:11-91)")
mean_below_threshold = HipGUFunc("mean_below_threshold", "(n),()->()", ["ff->f", "dd->d"], _mean_below_threshold,
                                 "mean of the samples below a threshold (reference processors/arithmetic.py:9-62)")
min_max = HipGUFunc("min_max", "(n)->(),(),(),()", ["f->ffff", "d->dddd"], _min_max,
                    "first-occurrence argmin/argmax and values (reference processors/min_max.py:11-82)")


# --------------------------------------------------------------------------------------------------- '(n),...,(m)' in-place outputs
def _dwt(g, *args):
    if len(args) != 5:
        raise TypeError("discrete_wavelet_transform(w_in, level, wave_type, coeff, w_out): w_out must be passed (its length is the output size)")
    w_in, level, wave_type, coeff, w_out = args
    st = Staging()
    try:
        ptr, code, n_wf, n, stride, one_d = st.wf_in(w_in)
        sfx = loop_suffix(_dtype_of(w_in))
        ft = _F[sfx]
        if _mode_char(wave_type) not in (ord("h"), ord("d")):
            raise NotImplementedError("only the Haar wavelet ('h' / 'd' = db1) is implemented")
        m = w_out.shape[-1]
        optr, res = st.out(w_out, (m,) if one_d else (n_wf, m), ft)
        run(entry("dwt_haar", sfx), g.__name__, ptr, code, n_wf, n, stride, _as_int(level, g.__name__), _mode_char(coeff), optr, m, m)
        st.finish()
        return res
    finally:
        st.release()


def _windower(g, *args):
    if len(args) != 3:
        raise TypeError("windower(w_in, t0_in, w_out): w_out must be passed (its length is the window size)")
    w_in, t0_in, w_out = args
    st = Staging()
    try:
        ptr, code, n_wf, n, stride, one_d = st.wf_in(w_in)
        sfx = loop_suffix(_dtype_of(w_in))
        ft = _F[sfx]
        tp, tv = st.scalar_in(t0_in, n_wf, ft)
        m = w_out.shape[-1]
        optr, res = st.out(w_out, (m,) if one_d else (n_wf, m), ft)
        run(entry("windower", sfx), g.__name__, ptr, code, n_wf, n, stride, tp, tv, optr, m, m)
        st.finish()
        return res
    finally:
        st.release()


def _avg_current(g, *args):
    if len(args) != 3:
        raise TypeError("avg_current(w_in, length, w_out): w_out must be passed (len(w_in) - int(length) samples)")
    w_in, length, w_out = args
    st = Staging()
    try:
        ptr, code, n_wf, n, stride, one_d = st.wf_in(w_in)
        sfx = loop_suffix(_dtype_of(w_in))
        ft = _F[sfx]
        m = w_out.shape[-1]
        optr, res = st.out(w_out, (m,) if one_d else (n_wf, m), ft)
        run(entry("avg_current", sfx), g.__name__, ptr, code, n_wf, n, stride, float(np.asarray(length).reshape(-1)[0]), optr, m, m)
        st.finish()
        return res
    finally:
        st.release()


def _upsampler(g, *args):
    if len(args) != 3:
        raise TypeError("upsampler(w_in, upsample, w_out): w_out must be passed (its length is the output size)")
    w_in, upsample, w_out = args
    st = Staging()
    try:
        ptr, code, n_wf, n, stride, one_d = st.wf_in(w_in)
        sfx = loop_suffix(_dtype_of(w_in))
        ft = _F[sfx]
        m = w_out.shape[-1]
        optr, res = st.out(w_out, (m,) if one_d else (n_wf, m), ft)
        run(entry("upsampler", sfx), g.__name__, ptr, code, n_wf, n, stride, float(np.asarray(upsample).reshape(-1)[0]), optr, m, m)
        st.finish()
        return res
    finally:
        st.release()


def _moving_window_multi(g, *args):
    ins, outs = _split(g, args)
    st = Staging()
    try:
        ptr, code, n_wf, n, stride, one_d = st.wf_in(ins[0])
        sfx = loop_suffix(_dtype_of(ins[0]))
        ft = _F[sfx]
        optr, res = st.out(outs[0], (n,) if one_d else (n_wf, n), ft)
        run(entry("moving_window_multi", sfx), g.__name__, ptr, code, n_wf, n, stride, float(np.asarray(ins[1]).reshape(-1)[0]),
            float(np.asarray(ins[2]).reshape(-1)[0]), _as_int(ins[3], g.__name__), optr, n)
        st.finish()
        return res
    finally:
        st.release()


def _trap_pickoff(g, *args):
    ins, outs = _split(g, args)
    st = Staging()
    try:
        ptr, code, n_wf, n, stride, one_d = st.wf_in(ins[0])
        sfx = loop_suffix(_dtype_of(ins[0]))
        ft = _F[sfx]
        tp, tv = st.scalar_in(ins[3], n_wf, ft)
        optr, res = st.out(outs[0], () if one_d else (n_wf,), ft)
        run(entry("trap_pickoff", sfx), g.__name__, ptr, code, n_wf, n, stride, _as_int(ins[1], g.__name__), _as_int(ins[2], g.__name__),
            tp, tv, optr)
        st.finish()
        return res[()] if isinstance(res, np.ndarray) and res.ndim == 0 else res
    finally:
        st.release()


def _convolve(g, *args):
    if len(args) != 4:
        raise TypeError(f"{g.__name__}(w_in, kernel, mode_in, w_out): w_out must be passed (its length selects the output size)")
    w_in, kernel, mode_in, w_out = args
    st = Staging()
    try:
        ptr, code, n_wf, n, stride, one_d = st.wf_in(w_in)
        sfx = loop_suffix(_dtype_of(w_in))
        ft = _F[sfx]
        if isinstance(kernel, DeviceArray):
            kd = kernel
        else:
            k = np.ascontiguousarray(kernel, dtype=ft)
            if k.ndim != 1:
                raise NotImplementedError("per-waveform kernels are not supported")
            kd = DeviceArray.from_numpy(k)
            st.keep.append(kd)
        p = w_out.shape[-1]
        optr, res = st.out(w_out, (p,) if one_d else (n_wf, p), ft)
        run(entry("convolve_wf", sfx), g.__name__, ptr, code, n_wf, n, stride, kd.ptr, kd.shape[-1], _mode_char(mode_in), optr, p, p)
        st.finish()
        return res
    finally:
        st.release()


discrete_wavelet_transform = HipGUFunc("discrete_wavelet_transform", "(n),(),(),(),(m)", ["fibbf", "dlbbd"], _dwt,
                                       "Haar DWT approximation/detail coefficients (reference processors/dwt.py:13-81)")
convolve_wf = HipGUFunc("convolve_wf", "(n),(m),(),(p)", ["ffbf", "ddbd"], _convolve,
                        "FIR filter, np.convolve semantics, modes f v s (reference processors/convolutions.py:14-72)")
fft_convolve_wf = HipGUFunc("fft_convolve_wf", "(n),(m),(),(p)", ["ffbf", "ddbd"], _convolve,
                            "same filter as convolve_wf (the reference routes it through scipy fftconvolve, "
                            "processors/convolutions.py:75-119); evaluated in direct form on the device")


# --------------------------------------------------------------------------------------------------- kernel generators (host, run once)
def _check_kernel_params(sigma, flat, decay):
    # messages and order: reference processors/energy_kernels.py:51-61 / :115-125
    if sigma < 0:
        raise DSPFatal("The curvature parameter must be positive")
    if flat < 0:
        raise DSPFatal("The length of the flat section must be positive")
    if np.floor(flat) != flat:
        raise DSPFatal("The length of the flat section must be an integer")
    if decay < 0:
        raise DSPFatal("The decay constant must be positive")


def _scalar_for(kernel, v):
    # the reference's object-mode gufunc boxes float32 scalars into Python floats carrying the float32 value
    return float(np.float32(v)) if kernel.dtype == np.float32 else float(v)


def _cusp_shape(length, sigma, flat):
    """float64 cusp with flat top: sinh(i/sigma)/sinh(lt/sigma) rising, 1 on [lt, lt+flat], mirrored falling edge."""
    lt = int((length - flat) / 2)
    fl = int(flat)
    ind = np.arange(length, dtype=np.float64)
    cusp = np.zeros(length, dtype=np.float64)
    den = np.sinh(lt / sigma)
    cusp[:lt] = np.sinh(ind[:lt] / sigma) / den
    cusp[lt: lt + fl + 1] = 1.0
    cusp[lt + fl + 1:] = np.sinh((length - ind[lt + fl + 1:]) / sigma) / den
    return cusp, lt, fl


def _cusp_filter(g, sigma, flat, decay, kernel):
    sigma, flat, decay = (_scalar_for(kernel, v) for v in (sigma, flat, decay))
    _check_kernel_params(sigma, flat, decay)
    n = len(kernel)
    cusp, _, _ = _cusp_shape(n, sigma, flat)
    k = cusp.astype(kernel.dtype)  # the reference builds the cusp inside the output array (energy_kernels.py:63-70)
    kernel[:] = np.convolve(k, [1, -np.exp(-1 / decay)], "same")
    return kernel


def _zac_filter(g, sigma, flat, decay, kernel):
    sigma, flat, decay = (_scalar_for(kernel, v) for v in (sigma, flat, decay))
    _check_kernel_params(sigma, flat, decay)
    n = len(kernel)
    cusp, lt, fl = _cusp_shape(n, sigma, flat)
    ind = np.arange(n, dtype=np.float64)
    par = np.zeros(n, dtype=np.float64)
    par[:lt] = np.power(ind[:lt] - lt / 2, 2) - np.power(lt / 2, 2)
    par[lt + fl + 1:] = np.power(n - ind[lt + fl + 1:] - lt / 2, 2) - np.power(lt / 2, 2)
    areapar = areacusp = 0.0
    for a, b in zip(par.tolist(), cusp.tolist()):  # sequential sums, as the reference accumulates them (energy_kernels.py:146-149)
        areapar += a
        areacusp += b
    zac = cusp + (-par / areapar * areacusp)
    kernel[:] = np.convolve(zac, [1, -np.exp(-1 / decay)], "same")
    return kernel


windower = HipGUFunc("windower", "(n),(),(m)", ["fff", "ddd"], _windower,
                     "window of len(w_out) samples starting at int(t0_in), NaN outside the input (reference processors/windower.py:12-54)")
avg_current = HipGUFunc("avg_current", "(n),(),(m)", ["fff", "ddd"], _avg_current,
                        "(w_in[L:] - w_in[:-L]) / length (reference processors/moving_windows.py:206-249)")
upsampler = HipGUFunc("upsampler", "(n),(),(m)", ["fff", "ddd"], _upsampler,
                      "every sample repeated int(upsample) times (reference processors/upsampler.py:13-56)")
moving_window_multi = HipGUFunc("moving_window_multi", "(n),(),(),()->(n)", ["fffi->f", "dddi->d"], _moving_window_multi,
                                "moving averages applied alternately from the left and the right (reference processors/moving_windows.py:117-204)")
trap_pickoff = HipGUFunc("trap_pickoff", "(n),(),(),()->()", ["fiif->f", "diid->d"], _trap_pickoff,
                         "normalised difference of two rise-long window sums at an integer pick-off sample (reference processors/trap_filters.py:230-293)")
def _t0_filter(g, rise, fall, kernel):
    """t0 kernel: a ramp of weights 2 (r - i) / (r (r + 1)), i = 0 .. r - 1 (they sum to 1), followed by the plateau -1 / fall.
    Evaluated in float64 and rounded once into the kernel (the reference's generator runs in object mode on Python floats,
    processors/kernels.py:12-61); the three parameter checks keep the reference's order and texts."""
    rise, fall = _scalar_for(kernel, rise), _scalar_for(kernel, fall)
    for value, what in ((rise, "rise"), (fall, "fall")):
        if value < 0:
            raise DSPFatal(f"The length of the {what} section must be positive")
    if len(kernel) != rise + fall:
        raise DSPFatal("The length of the output kernel must equal rise+fall")
    r = int(rise)
    if r > 0:
        ramp = 2.0 * np.arange(r, 0, -1, dtype=np.float64)  # 2 (r - i): exact integers
        kernel[:r] = ramp / (rise * (rise + 1))
    if len(kernel) > r:  # (an empty section divides by nothing)
        kernel[r:] = -1.0 / fall
    return kernel


def _moving_slope(g, kernel):
    """Least-squares slope of n equidistant samples as FIR weights, w_j = (n j - S1) / (n S2 - S1^2) for j = n .. 1 (the order a
    convolution wants), S1 = sum j, S2 = sum j^2.  Numerator and denominator are rounded to the kernel's type before the division,
    which is where the reference's in-place array arithmetic rounds (processors/kernels.py:69-100)."""
    n = len(kernel)
    s1 = n * (n + 1) / 2
    s2 = n * (n + 1) * (2 * n + 1) / 6
    numer = (n * np.arange(n, 0, -1, dtype=np.float64) - s1).astype(kernel.dtype)
    kernel[:] = numer / kernel.dtype.type(n * s2 - s1 * s1)
    return kernel


t0_filter = HipGUFunc("t0_filter", "(),(),(n)", ["fff", "ddd"], _t0_filter,
                      "t0 kernel generator (asymmetric trapezoid weights), host, once (reference processors/kernels.py:12-66)")
moving_slope = HipGUFunc("moving_slope", "(n)", ["f", "d"], _moving_slope,
                         "moving-slope kernel generator, host, once (reference processors/kernels.py:69-98)")
cusp_filter = HipGUFunc("cusp_filter", "(),(),(),(n)", ["ffff", "dddd"], _cusp_filter,
                        "CUSP kernel generator, evaluated once on the host at chain build (reference processors/energy_kernels.py:12-73)")
zac_filter = HipGUFunc("zac_filter", "(),(),(),(n)", ["ffff", "dddd"], _zac_filter,
                       "zero-area CUSP kernel generator, host, once (reference processors/energy_kernels.py:76-157)")

__all__ = ["bl_subtract", "pole_zero", "double_pole_zero", "trap_filter", "trap_norm", "asym_trap_filter", "fixed_time_pickoff",
           "time_point_thresh", "interpolated_time_point_thresh", "min_max", "min_max_norm", "linear_slope_fit", "mean_below_threshold", "windower", "avg_current", "upsampler", "moving_window_multi", "trap_pickoff", "discrete_wavelet_transform", "convolve_wf", "fft_convolve_wf", "cusp_filter", "zac_filter", "t0_filter", "moving_slope"]
