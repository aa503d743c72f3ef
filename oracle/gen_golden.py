#!/usr/bin/env python3
"""Golden-vector generator for the dspeed hot path (TEST INFRASTRUCTURE, never shipped).

Runs ONLY in the build container, where the read-only reference checkout lives at
/root/reference.  It executes the reference's *own* processor bodies (the pure-Python
functions underneath the numba decorators -- the same thing the reference's
``compare_numba_vs_python`` fixture treats as the semantic spec,
tests/conftest.py:62-180) on seeded inputs and stores inputs + outputs as small
``.npz`` fixtures under tests/golden/.  No reference source text is written anywhere:
fixtures are numeric data only.

numba itself is not importable here, so a pass-through stub of the ``numba`` module is
registered and the numba *typing rules* are emulated through the argument types we
feed each body:

* nopython kernels, float32 loop: arrays are ``np.float32``; a float scalar that the body
  mixes with an int literal (``-1 / t_tau``, ``t_in - i_in``) is passed as
  ``np.float64(np.float32(v))`` because numba promotes int64 (op) float32 -> float64,
  while NumPy-2 (NEP 50) would keep float32;  int scalars are ``np.int32``.
* object-mode (``forceobj=True``) kernels -- cusp_filter, zac_filter, convolve_wf,
  discrete_wavelet_transform: numba boxes float32 scalars into Python ``float`` and
  int8/int32 scalars into Python ``int`` before calling the body, arrays stay float32.
* ``fixed_time_pickoff`` mode 's' mixes every waveform read with int64/float64, so numba
  evaluates it entirely in float64 -> the waveform is fed as a float64 copy of the
  float32 samples for that mode only.

Usage:  python oracle/gen_golden.py            (writes tests/golden/*.npz)
        /opt/conda/bin/python3.9 oracle/gen_golden.py --dwt   (PyWavelets 1.1.1 lives there)
        python oracle/gen_golden.py --arithmetic   (only tests/golden/arithmetic.npz)
        python oracle/gen_golden.py --windows      (only tests/golden/windows.npz)
"""
from __future__ import annotations

import importlib
import json
import os
import sys
import types

import numpy as np

REF_SRC = "/root/reference/src/dspeed"
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


# ----------------------------------------------------------------------------- numba stub
def _install_stubs():
    nb = types.ModuleType("numba")

    def _passthrough(*_a, **_k):
        def deco(f):
            return f

        return deco

    nb.guvectorize = _passthrough
    nb.vectorize = _passthrough
    nb.jit = _passthrough
    nb.njit = _passthrough
    sys.modules["numba"] = nb
    for name in ("numba.np", "numba.np.ufunc", "numba.np.ufunc.sigparse"):
        sys.modules[name] = types.ModuleType(name)
    sp = sys.modules["numba.np.ufunc.sigparse"]

    def parse_signature(sig):  # minimal stand-in for the numba helper (dims only)
        ins, _, outs = sig.partition("->")

        def dims(s):
            return [tuple(x for x in p.strip("() ").split(",") if x) for p in s.split("),") if p.strip()]

        return dims(ins), dims(outs) if outs else []

    sp.parse_signature = parse_signature
    sys.modules["numba.np.ufunc"].sigparse = sp
    sys.modules["numba.np"].ufunc = sys.modules["numba.np.ufunc"]
    nb.np = sys.modules["numba.np"]
    pkg = types.ModuleType("dspeed")
    pkg.__path__ = [REF_SRC]  # namespace-style: skips dspeed/__init__.py (needs lgdo/lh5)
    sys.modules["dspeed"] = pkg
    pp = types.ModuleType("dspeed.processors")
    pp.__path__ = [os.path.join(REF_SRC, "processors")]
    sys.modules["dspeed.processors"] = pp


def _ref(modname):
    return importlib.import_module("dspeed.processors." + modname)


# ----------------------------------------------------------------------------- helpers
class Book:
    """Collects cases: arrays go to the npz, scalars/metadata into a JSON index."""

    def __init__(self, name):
        self.name = name
        self.arrays = {}
        self.index = []

    def add(self, case, kernel, dtype, arrays, params=None, fatal=False, note=""):
        entry = {"case": case, "kernel": kernel, "dtype": dtype, "params": params or {},
                 "fatal": bool(fatal), "note": note, "arrays": sorted(arrays)}
        for k, v in arrays.items():
            self.arrays[f"{case}/{k}"] = np.asarray(v)
        self.index.append(entry)

    def save(self):
        os.makedirs(GOLDEN, exist_ok=True)
        path = os.path.join(GOLDEN, self.name + ".npz")
        np.savez_compressed(path, __index__=np.array(json.dumps(self.index)), **self.arrays)
        print(f"wrote {path}: {len(self.index)} cases, {os.path.getsize(path)/1024:.1f} KiB")


def f64_of_f32(v):
    return np.float64(np.float32(v))


def synth_waveforms(rng, n_wf, wf_len, tau=1716.28, sigma=5.0, amp=(500, 15000), bl=(9000, 11000),
                    t0_frac=(0.45, 0.55), dtype=np.float32):
    """Synthetic HPGe-like pulses: baseline + step with exponential decay + noise."""
    i = np.arange(wf_len, dtype=np.float64)[None, :]
    B = rng.uniform(*bl, size=(n_wf, 1))
    A = rng.uniform(*amp, size=(n_wf, 1))
    t0 = np.floor(rng.uniform(*t0_frac, size=(n_wf, 1)) * wf_len)
    x = B + A * np.exp(-(i - t0) / tau) * (i >= t0) + sigma * rng.standard_normal((n_wf, wf_len))
    return x.astype(dtype), B[:, 0].astype(dtype), t0[:, 0]


def run_body(fn, *args):
    """Call a reference body; returns True if it raised DSPFatal."""
    from dspeed.errors import DSPFatal

    try:
        fn(*args)
    except DSPFatal:
        return True
    return False


# ----------------------------------------------------------------------------- per-kernel callers
def call_bl_subtract(w, bl, dt):
    m = _ref("bl_subtract")
    out = np.empty_like(w)
    fatal = run_body(m.bl_subtract, w, dt(bl), out)
    return out, fatal


def call_pole_zero(w, tau, dt):
    m = _ref("pole_zero")
    out = np.empty_like(w)
    t = f64_of_f32(tau) if dt is np.float32 else np.float64(tau)
    fatal = run_body(m.pole_zero, w, t, out)
    return out, fatal


def call_double_pole_zero(w, tau1, tau2, frac, dt):
    m = _ref("pole_zero")
    out = np.empty_like(w)
    if dt is np.float32:
        a = (f64_of_f32(tau1), f64_of_f32(tau2), np.float32(frac))
    else:
        a = (np.float64(tau1), np.float64(tau2), np.float64(frac))
    fatal = run_body(m.double_pole_zero, w, *a, out)
    return out, fatal


def call_trap(name, w, *ints):
    m = _ref("trap_filters")
    out = np.empty_like(w)
    fatal = run_body(getattr(m, name), w, *[np.int32(v) for v in ints], out)
    return out, fatal


def call_pickoff(w, t_in, mode, dt):
    m = _ref("fixed_time_pickoff")
    out = np.empty(1, dtype=w.dtype)
    if dt is np.float32:
        t = f64_of_f32(t_in)
        wv = w.astype(np.float64) if mode == "s" else w
    else:
        t, wv = np.float64(t_in), w
    fatal = run_body(m.fixed_time_pickoff, wv, t, ord(mode), out)
    return out[0], fatal


def call_tpt(w, thr, t_start, walk, dt):
    m = _ref("time_point_thresh")
    out = np.empty(1, dtype=w.dtype)
    fatal = run_body(m.time_point_thresh, w, dt(thr), dt(t_start), dt(walk), out)
    return out[0], fatal


def call_itpt(w, thr, t_start, walk, mode, dt):
    m = _ref("time_point_thresh")
    out = np.empty(1, dtype=w.dtype)
    fatal = run_body(m.interpolated_time_point_thresh, w, dt(thr), dt(t_start), int(walk), ord(mode), out)
    return out[0], fatal


def call_min_max(w):
    m = _ref("min_max")
    o = [np.empty(1, dtype=w.dtype) for _ in range(4)]
    fatal = run_body(m.min_max, w, *o)
    return np.array([x[0] for x in o], dtype=w.dtype), fatal


def call_mean_below(w, thr):
    """float32 loop: numba keeps `total` (0.0) in float64 and adds float32 samples to it; plain NumPy-2 would demote the
    Python float to float32.  Feeding the float32 samples as float64 copies reproduces numba (comparison unchanged)."""
    m = _ref("arithmetic")
    out = np.empty(1, dtype=w.dtype)
    fatal = run_body(m.mean_below_threshold, w.astype(np.float64), np.float64(w.dtype.type(thr)), out)
    return out[0], fatal


def call_kernel_gen(name, sigma, flat, decay, length, dt):
    m = _ref("energy_kernels")
    out = np.zeros(length, dtype=dt)
    # object mode: float32 scalars are boxed to Python floats holding the float32 value
    cv = (lambda v: float(np.float32(v))) if dt is np.float32 else float
    fatal = run_body(getattr(m, name), cv(sigma), cv(flat), cv(decay), out)
    return out, fatal


def call_convolve(w, k, mode, out_len):
    m = _ref("convolutions")
    out = np.empty(out_len, dtype=w.dtype)
    fatal = run_body(m.convolve_wf, w, k, ord(mode), out)
    return out, fatal


def call_fft_convolve(wblock, k, mode, out_len):
    m = _ref("convolutions")
    out = np.empty((wblock.shape[0], out_len), dtype=wblock.dtype)
    body = m.fft_convolve_wf.ufunc  # GUFuncWrapper(vectorized=True): body works on the block
    fatal = run_body(body, wblock.copy(), k, ord(mode), out)
    return out, fatal


# ----------------------------------------------------------------------------- case builders
def gen_elementwise(rng):
    b = Book("bl_subtract")
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        w, bl, _ = synth_waveforms(rng, 3, 256, dtype=dt)
        for r in range(3):
            out, fatal = call_bl_subtract(w[r], bl[r], dt)
            b.add(f"{tag}_synth{r}", "bl_subtract", tag, {"w_in": w[r], "baseline": bl[r], "w_out": out}, fatal=fatal)
        wn = w[0].copy()
        wn[17] = np.nan
        out, fatal = call_bl_subtract(wn, bl[0], dt)
        b.add(f"{tag}_nan_in", "bl_subtract", tag, {"w_in": wn, "baseline": bl[0], "w_out": out}, fatal=fatal)
        out, fatal = call_bl_subtract(w[0], np.nan, dt)
        b.add(f"{tag}_nan_bl", "bl_subtract", tag, {"w_in": w[0], "baseline": dt(np.nan), "w_out": out}, fatal=fatal)
    b.save()


def gen_pole_zero(rng):
    b = Book("pole_zero")
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        # the reference's own known-answer input (tests/processors/test_pole_zero.py:20-48)
        tau, amp = 30000, 17500
        ts = np.arange(0, 8192, dtype=np.float64)
        pulse = np.zeros(len(ts) + 20, dtype=dt)
        pulse[20:] = amp * np.exp(-ts / tau)
        out, fatal = call_pole_zero(pulse, tau, dt)
        b.add(f"{tag}_reftest_step", "pole_zero", tag, {"w_in": pulse, "w_out": out}, {"tau": tau}, fatal,
              note="expected ~ step of 17500 after 20 zeros (rtol 1e-6 f32 / 1e-7 f64)")
        w, bl, _ = synth_waveforms(rng, 3, 1024, dtype=dt)
        for r in range(3):
            x = (w[r] - bl[r]).astype(dt)
            out, fatal = call_pole_zero(x, 1716.28, dt)
            b.add(f"{tag}_synth{r}", "pole_zero", tag, {"w_in": x, "w_out": out}, {"tau": 1716.28}, fatal)
        w4, bl4, _ = synth_waveforms(rng, 1, 4096, dtype=dt)
        x = (w4[0] - bl4[0]).astype(dt)
        out, fatal = call_pole_zero(x, 1716.28, dt)
        b.add(f"{tag}_synth4096", "pole_zero", tag, {"w_in": x, "w_out": out}, {"tau": 1716.28}, fatal)
        short = np.array([3.0, -1.0, 2.5, 7.0, 7.5], dtype=dt)
        for tau_s in (1.0, 0.5, 1e9):
            out, fatal = call_pole_zero(short, tau_s, dt)
            b.add(f"{tag}_short_tau{tau_s:g}", "pole_zero", tag, {"w_in": short, "w_out": out}, {"tau": tau_s}, fatal)
        one = np.array([4.25], dtype=dt)
        out, fatal = call_pole_zero(one, 10.0, dt)
        b.add(f"{tag}_len1", "pole_zero", tag, {"w_in": one, "w_out": out}, {"tau": 10.0}, fatal)
        wn = np.ones(64, dtype=dt)
        wn[4] = np.nan
        out, fatal = call_pole_zero(wn, 30000, dt)
        b.add(f"{tag}_nan_in", "pole_zero", tag, {"w_in": wn, "w_out": out}, {"tau": 30000}, fatal)
        out, fatal = call_pole_zero(np.ones(64, dtype=dt), np.nan, dt)
        b.add(f"{tag}_nan_tau", "pole_zero", tag, {"w_in": np.ones(64, dtype=dt), "w_out": out}, {"tau": float("nan")}, fatal)
        winf = np.ones(16, dtype=dt)
        winf[3] = np.inf
        out, fatal = call_pole_zero(winf, 100.0, dt)
        b.add(f"{tag}_inf_in", "pole_zero", tag, {"w_in": winf, "w_out": out}, {"tau": 100.0}, fatal,
              note="inf - inf -> NaN in output -> DSPFatal")
    b.save()

    b = Book("double_pole_zero")
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        # tests/processors/test_pole_zero.py:51-96
        wf_len, tp0, amp, tau1, tau2, frac = 8192, 20, 17500, 1000, 30000, 0.98
        ts = np.arange(0, wf_len - tp0, dtype=np.float64)
        ys = amp * (1 - frac) * np.exp(-ts / tau1) + amp * frac * np.exp(-ts / tau2)
        pulse = np.zeros(wf_len, dtype=dt)
        pulse[tp0:] = ys
        out, fatal = call_double_pole_zero(pulse, tau1, tau2, frac, dt)
        b.add(f"{tag}_reftest_step", "double_pole_zero", tag, {"w_in": pulse, "w_out": out},
              {"tau1": tau1, "tau2": tau2, "frac": frac}, fatal)
        w, bl, _ = synth_waveforms(rng, 2, 1024, dtype=dt)
        for r in range(2):
            x = (w[r] - bl[r]).astype(dt)
            out, fatal = call_double_pole_zero(x, 1716.28, 62.5, 0.02, dt)
            b.add(f"{tag}_synth{r}", "double_pole_zero", tag, {"w_in": x, "w_out": out},
                  {"tau1": 1716.28, "tau2": 62.5, "frac": 0.02}, fatal)
        out, fatal = call_double_pole_zero(np.ones(3, dtype=dt), tau1, tau2, frac, dt)
        b.add(f"{tag}_len3_fatal", "double_pole_zero", tag, {"w_in": np.ones(3, dtype=dt), "w_out": out},
              {"tau1": tau1, "tau2": tau2, "frac": frac}, fatal)
        four = np.array([1.0, 2.0, 4.0, 8.0], dtype=dt)
        out, fatal = call_double_pole_zero(four, 10.0, 3.0, 0.25, dt)
        b.add(f"{tag}_len4", "double_pole_zero", tag, {"w_in": four, "w_out": out},
              {"tau1": 10.0, "tau2": 3.0, "frac": 0.25}, fatal)
        wn = np.ones(64, dtype=dt)
        wn[4] = np.nan
        out, fatal = call_double_pole_zero(wn, tau1, tau2, frac, dt)
        b.add(f"{tag}_nan_in", "double_pole_zero", tag, {"w_in": wn, "w_out": out},
              {"tau1": tau1, "tau2": tau2, "frac": frac}, fatal)
        out, fatal = call_double_pole_zero(np.ones(64, dtype=dt), tau1, np.nan, frac, dt)
        b.add(f"{tag}_nan_tau2", "double_pole_zero", tag, {"w_in": np.ones(64, dtype=dt), "w_out": out},
              {"tau1": tau1, "tau2": float("nan"), "frac": frac}, fatal)
    b.save()


def _pz_step(rng, wf_len, dt, amp=None):
    """A pole-zero corrected synthetic pulse (what the trap filters see in the energy chain)."""
    w, bl, t0 = synth_waveforms(rng, 1, wf_len, dtype=dt, amp=amp or (500, 15000))
    x = (w[0] - bl[0]).astype(dt)
    out, _ = call_pole_zero(x, 1716.28, dt)
    return out, t0[0]


def gen_traps(rng):
    b = Book("trap_filters")
    ramp16 = np.arange(1, 17)
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        geo = [(4096, 625, 188), (1024, 64, 16), (1024, 100, 0), (256, 128, 0), (256, 1, 0), (256, 1, 254), (100, 7, 13)]
        for k, (n, r, f) in enumerate(geo):
            w, _ = _pz_step(rng, n, dt)
            for name in ("trap_filter", "trap_norm"):
                out, fatal = call_trap(name, w, r, f)
                b.add(f"{tag}_{name}_geo{k}", name, tag, {"w_in": w, "w_out": out}, {"rise": r, "flat": f}, fatal)
        # big amplitude -> accumulator beyond 2^24 (fp32 rounding every step)
        w, _ = _pz_step(rng, 4096, dt, amp=(17000, 17500))
        out, fatal = call_trap("trap_filter", w, 1000, 300)
        b.add(f"{tag}_trap_filter_big", "trap_filter", tag, {"w_in": w, "w_out": out}, {"rise": 1000, "flat": 300}, fatal)
        # small exact cases (SURVEY 8a edge semantics)
        r16 = ramp16.astype(dt)
        for name, ints in (("trap_filter", (2, 0)), ("trap_filter", (1, 1)), ("trap_filter", (0, 3)), ("trap_filter", (0, 0)),
                           ("trap_filter", (8, 0)), ("trap_filter", (5, 6)), ("trap_norm", (2, 1)), ("trap_norm", (3, 0)),
                           ("trap_norm", (0, 2)), ("asym_trap_filter", (2, 1, 4)), ("asym_trap_filter", (1, 0, 1)),
                           ("asym_trap_filter", (4, 4, 8)), ("asym_trap_filter", (0, 1, 2)), ("asym_trap_filter", (2, 1, 0))):
            with np.errstate(all="ignore"):
                out, fatal = call_trap(name, r16, *ints)
            b.add(f"{tag}_{name}_ramp_{'_'.join(map(str, ints))}", name, tag, {"w_in": r16, "w_out": out},
                  dict(zip(("rise", "flat", "fall"), ints)), fatal)
        for name, ints in (("trap_filter", (-1, 2)), ("trap_filter", (2, -1)), ("trap_filter", (7, 3)), ("trap_norm", (-1, 2)),
                           ("trap_norm", (8, 1)), ("asym_trap_filter", (-1, 1, 1)), ("asym_trap_filter", (1, -1, 1)),
                           ("asym_trap_filter", (1, 1, -1)), ("asym_trap_filter", (8, 4, 5))):
            out, fatal = call_trap(name, r16, *ints)
            b.add(f"{tag}_{name}_fatal_{'_'.join(map(str, ints))}".replace("-", "m"), name, tag,
                  {"w_in": r16, "w_out": out}, dict(zip(("rise", "flat", "fall"), ints)), fatal)
        wn = r16.copy()
        wn[5] = np.nan
        for name, ints in (("trap_filter", (2, 1)), ("trap_norm", (2, 1)), ("asym_trap_filter", (2, 1, 4))):
            out, fatal = call_trap(name, wn, *ints)
            b.add(f"{tag}_{name}_nan_in", name, tag, {"w_in": wn, "w_out": out}, dict(zip(("rise", "flat", "fall"), ints)), fatal)
        # asym trap at the ICPC geometry (8/4/125 samples) on 8192 and 1024 samples
        for n in (8192, 1024):
            w, _ = _pz_step(rng, n, dt)
            out, fatal = call_trap("asym_trap_filter", w, 8, 4, 125)
            b.add(f"{tag}_asym_trap_filter_icpc{n}", "asym_trap_filter", tag, {"w_in": w, "w_out": out},
                  {"rise": 8, "flat": 4, "fall": 125}, fatal)
    b.save()


def gen_pickoff(rng):
    b = Book("fixed_time_pickoff")
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        n = 20
        ones = np.ones(n, dtype=dt)
        wn = ones.copy()
        wn[4] = np.nan
        k = 0
        # tests/processors/test_fixed_time_pickoff.py:15-90
        for w, t, mode in ((wn, 1, "i"), (ones, np.nan, "i"), (ones, -1, "i"), (ones, n, "i"), (ones, 1.5, "i"), (ones, 1.5, " "),
                           (ones, n - 1, "l"), (ones, n - 0.5, "l"), (ones, 0, "h"), (ones, 3, " ")):
            out, fatal = call_pickoff(w, t, mode, dt)
            b.add(f"{tag}_edge{k}", "fixed_time_pickoff", tag, {"w_in": w, "a_out": out}, {"t_in": float(t), "mode": mode}, fatal)
            k += 1
        ramp = np.arange(n, dtype=dt)
        sine = np.sin(np.arange(n)).astype(dt)
        noise = rng.standard_normal(64).astype(dt) * 100
        for wname, w in (("ramp", ramp), ("sine", sine), ("noise", noise)):
            for mode in "nfclhs":
                for t in (3.5, 3.25, 0.2, len(w) - 1.8, 7.75, 3.0, 0.5, len(w) - 1.0, len(w) - 1.25):
                    out, fatal = call_pickoff(w, t, mode, dt)
                    b.add(f"{tag}_{wname}_{mode}_{t:g}", "fixed_time_pickoff", tag, {"w_in": w, "a_out": out},
                          {"t_in": float(t), "mode": mode}, fatal)
        out, fatal = call_pickoff(ramp, 3, "i", dt)
        b.add(f"{tag}_ramp_i_3", "fixed_time_pickoff", tag, {"w_in": ramp, "a_out": out}, {"t_in": 3.0, "mode": "i"}, fatal)
    b.save()


def gen_tpt(rng):
    b = Book("time_point_thresh")
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        saw = np.concatenate([np.arange(-1, 5, 1), np.arange(-1, 5, 1)]).astype(dt)
        sn = saw.copy()
        sn[4] = np.nan
        cases = [(sn, 1, 11, 0), (saw, np.nan, 11, 0), (saw, 1, np.nan, 0), (saw, 1, 11, np.nan), (saw, 1, 10.5, 0),
                 (saw, 1, 11, 0.5), (saw, 1, 12, 0), (saw, 1, -1, 1), (saw, 1, 11, 0), (saw, 3, 0, 1),
                 (np.array([5, 4, 3, 2, 1, 0, -1.0], dtype=dt), 2.5, 0, 1), (np.array([0, 1, 2, 3, 4, 5.0], dtype=dt), 2.5, 0, 1),
                 (np.array([-5, -4, -3, -2, -1, 0.0], dtype=dt), -2.5, 0, 1), (np.array([0, -1, -2, -3, -4, -5.0], dtype=dt), -2.5, 0, 1),
                 (np.array([3, 1, 1, 5, 5, 2.0], dtype=dt), 4, 5, 0), (np.array([3, 1, 1, 5, 5, 2.0], dtype=dt), 4, 0, 1),
                 (np.array([3, 1, 1, 5, 5, 2.0], dtype=dt), 10, 0, 1), (np.array([3, 1, 1, 5, 5, 2.0], dtype=dt), 10, 5, 0),
                 (saw, 1, 0, 0), (saw, 4, 11, 1), (saw, 4, 5, 0), (saw, 4, 5, 2), (saw, -1, 6, 0), (saw, 2, 11, -1)]
        for k, (w, thr, ts, wf) in enumerate(cases):
            out, fatal = call_tpt(w, thr, ts, wf, dt)
            b.add(f"{tag}_case{k}", "time_point_thresh", tag, {"w_in": w, "t_out": out},
                  {"a_threshold": float(thr), "t_start": float(ts), "walk_forward": float(wf)}, fatal)
        for r in range(4):
            w, t0 = _pz_step(rng, 1024, dt)
            at, _ = call_trap("asym_trap_filter", w, 8, 4, 125)
            mm, _ = call_min_max(at)
            for thr_scale, walk in ((0.05, 0), (0.5, 0), (0.5, 1), (2.0, 0)):
                thr = dt(thr_scale * mm[3])
                ts = mm[1] if walk == 0 else 0
                out, fatal = call_tpt(at, thr, ts, walk, dt)
                b.add(f"{tag}_synth{r}_{thr_scale:g}_{walk}", "time_point_thresh", tag, {"w_in": at, "t_out": out},
                      {"a_threshold": float(thr), "t_start": float(ts), "walk_forward": float(walk)}, fatal)
    b.save()

    b = Book("min_max")
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        cases = [np.array([3, 1, 1, 5, 5, 2.0], dtype=dt), np.array([7.0], dtype=dt), np.zeros(70, dtype=dt),
                 np.arange(130, dtype=dt), -np.arange(130, dtype=dt), rng.standard_normal(1000).astype(dt),
                 np.array([0.0, -0.0, 0.0, -0.0], dtype=dt), np.array([np.inf, -np.inf, 1, np.inf, -np.inf], dtype=dt)]
        tie = np.zeros(4096, dtype=dt)
        tie[[100, 2000, 4095]] = 9
        tie[[63, 64, 3000]] = -9
        cases.append(tie)
        wn = rng.standard_normal(100).astype(dt)
        wn[99] = np.nan
        cases.append(wn)
        for k, w in enumerate(cases):
            out, fatal = call_min_max(w)
            b.add(f"{tag}_case{k}", "min_max", tag, {"w_in": w, "out": out}, {}, fatal, note="out = t_min,t_max,a_min,a_max")
    b.save()


def gen_windows():
    """windower, avg_current, trap_pickoff; own seed, own book"""
    rng = np.random.default_rng(0x51DE)
    b = Book("windows")
    mw, mv, mt = _ref("windower"), _ref("moving_windows"), _ref("trap_filters")
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        w = synth_waveforms(rng, 1, 1000, dtype=dt)[0][0]
        wn = w.copy()
        wn[500] = np.nan
        k = 0
        for src, t0, m in [(w, 0, 100), (w, 250, 300), (w, 250.9, 300), (w, 900, 300), (w, 999, 10), (w, 1000, 10), (w, 5000, 10), (w, -1, 50),
                           (w, -49, 50), (w, -50, 50), (w, -500.5, 50), (w, -0.5, 999), (w, np.nan, 50), (wn, 10, 50), (w, 0, 1000), (w, 0, 1200)]:
            out = np.empty(m, dtype=dt)
            fatal = run_body(mw.windower, src, dt(t0), out)
            b.add(f"{tag}_win{k}", "windower", tag, {"w_in": src, "w_out": out}, {"t0_in": float(dt(t0))}, fatal)
            k += 1
        k = 0
        for src, length in [(w, 1), (w, 2), (w, 7), (w, 500), (w, 999), (wn, 3), (w, 1000), (w, -1), (w, 2.0)]:
            L = int(length)
            m = max(len(src) - L, 1) if 0 < L < len(src) else 10
            out = np.empty(m, dtype=dt)
            fatal = run_body(mv.avg_current, src, dt(length), out)
            b.add(f"{tag}_cur{k}", "avg_current", tag, {"w_in": src, "w_out": out}, {"length": float(dt(length))}, fatal)
            k += 1
        k = 0
        # numba: i_1 / i_2 start as 0.0 (float64) and add T samples -> float64 sums; feed float64 copies for the float32 loop
        for src, rise, flat, tpo in [(w, 10, 5, 600), (w, 100, 0, 999), (w, 100, 30, 229), (w, 100, 30, 228), (w, 1, 0, 1), (w, 1, 0, 0),
                                    (w, 400, 200, 999), (w, 400, 201, 999), (w, -1, 5, 600), (w, 5, -1, 600), (w, 10, 5, 600.5), (wn, 10, 5, 600),
                                    (w, 10, 5, np.nan), (w, 10, 5, 1000), (w, 10, 5, 5000), (w, 10, 5, -3)]:
            out = np.empty(1, dtype=dt)
            fatal = run_body(mt.trap_pickoff, src.astype(np.float64), np.int32(rise), np.int32(flat), np.float64(dt(tpo)), out)
            b.add(f"{tag}_tpo{k}", "trap_pickoff", tag, {"w_in": src, "a_out": out[0]}, {"rise": rise, "flat": flat, "t_pickoff": float(dt(tpo))}, fatal)
            k += 1
    b.save()


def gen_current():
    """upsampler, moving_window_multi (the A/E branch, icpc-dsp-config.json:323-334)"""
    rng = np.random.default_rng(0xC0FFEE)
    b = Book("current")
    mu, mv = _ref("upsampler"), _ref("moving_windows")
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        w = (100 * rng.standard_normal(300) + 50 * np.sin(np.arange(300) / 9.0)).astype(dt)
        wn = w.copy()
        wn[17] = np.nan
        k = 0
        # numba: t_in * upsample is int64 * T -> float64, upsample / 2 is T / int -> float64: feed the T value as a float64 scalar
        for src, up, m in [(w, 16, 4800), (w, 16, 4784), (w, 16, 5000), (w, 1, 300), (w, 2, 600), (w, 3, 900), (w, 2.5, 750), (w, 0.5, 150),
                           (w, 16, 100), (wn, 4, 1200), (w, 0, 10), (w, -2, 10)]:
            out = np.empty(m, dtype=dt)
            fatal = run_body(mu.upsampler, src, np.float64(dt(up)), out)
            b.add(f"{tag}_up{k}", "upsampler", tag, {"w_in": src, "w_out": out}, {"upsample": float(dt(up))}, fatal)
            k += 1
        k = 0
        wl = (100 * rng.standard_normal(4800) + 3000 * np.exp(-((np.arange(4800) - 2400) / 300.0) ** 2)).astype(dt)
        for src, L, num, typ in [(w, 5, 1, 1), (w, 5, 1, 2), (w, 5, 2, 0), (w, 5, 3, 0), (w, 1, 3, 0), (w, 2, 4, 0), (w, 299, 1, 1), (wl, 48, 3, 0),
                                 (wl, 48, 3, 1), (wl, 48, 3, 2), (wl, 100, 5, 0), (w, 5, 0, 0), (wn, 5, 3, 0), (w, 5.5, 3, 0), (w, 5, 2.5, 0),
                                 (w, 300, 3, 0), (w, -1, 3, 0), (w, 5, -1, 0)]:
            out = np.empty(len(src), dtype=dt)
            with np.errstate(all="ignore"):
                fatal = run_body(mv.moving_window_multi, src, dt(L), dt(num), np.int32(typ), out)
            b.add(f"{tag}_mw{k}", "moving_window_multi", tag, {"w_in": src, "w_out": out}, {"length": float(L), "num_mw": float(num), "mw_type": typ}, fatal)
            k += 1
    b.save()


def gen_linear_slope_fit():
    """PARITY UNPINNED.  linear_slope_fit's nopython typing (in-place array expressions mixing float32 arrays with int64 scalars)
    cannot be reproduced by running the reference body under NumPy 2 -- `temp / (i + 1)` is float32 there, float64 in numba -- so
    these fixtures come from an explicit emulation of numba's rules (the same rules the C oracle restates), and the result of the
    plain NumPy-2 execution of the reference body is stored beside them (`numpy2_*`) to show the size of the difference."""
    rng = np.random.default_rng(0x5107E)
    b = Book("linear_slope_fit")
    ref = _ref("linear_slope_fit")

    def emulate(w):
        T = w.dtype.type
        m, s = T(0), T(0)
        sxy, sy, sx, sx2 = np.float64(0), np.float64(0), 0, 0
        n = len(w)
        for i in range(n):
            temp = T(w[i] - m)
            m = T(np.float64(m) + np.float64(temp) / np.float64(i + 1))
            s = T(s + T(temp * T(w[i] - m)))
            sx += i
            sx2 += i * i
            sxy += np.float64(w[i]) * np.float64(i)
            sy += np.float64(w[i])
        with np.errstate(all="ignore"):
            s = np.sqrt(T(np.float64(s) / np.float64(n - 1)))
            slope = T((np.float64(n) * sxy - np.float64(sx) * sy) / np.float64(n * sx2 - sx * sx))
            icpt = T((sy - np.float64(sx) * np.float64(slope)) / np.float64(n))
        return np.array([m, s, slope, icpt], dtype=w.dtype)

    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        cases = [np.array([1, 2, 3, 4, 5.0], dtype=dt), np.array([7.0, 7.0], dtype=dt), np.arange(100, dtype=dt) * 3 - 20]
        for n in (750, 1650, 6692):
            cases.append(synth_waveforms(rng, 1, n, dtype=dt)[0][0])
        cases.append((10000 + 5 * rng.standard_normal(750) + 0.02 * np.arange(750)).astype(dt))
        wn = cases[3].copy()
        wn[5] = np.nan
        cases.append(wn)
        for k, w in enumerate(cases):
            if np.isnan(w).any():
                out = np.full(4, np.nan, dtype=dt)
            else:
                out = emulate(w)
            o = [np.empty(1, dtype=dt) for _ in range(4)]
            with np.errstate(all="ignore"):
                run_body(ref.linear_slope_fit, w, *o)
            b.add(f"{tag}_lsf{k}", "linear_slope_fit", tag, {"w_in": w, "out": out, "numpy2_out": np.array([x[0] for x in o], dtype=dt)}, {},
                  False, note="out = mean, stdev, slope, intercept under numba typing (emulated); numpy2_out = the body run under NumPy 2")
    b.save()


def gen_kernels():
    """t0_filter, moving_slope (object-mode generators, kernels.py): float32 scalars arrive as Python floats"""
    b = Book("kernels")
    m = _ref("kernels")
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        for k, (rise, fall) in enumerate([(8, 125), (1, 1), (0, 5), (5, 0), (3.0, 7.0), (-1, 5), (4, -2), (8, 100)]):
            n = int(rise + fall) if (rise >= 0 and fall >= 0) else 6
            if k == 7:
                n = 50  # wrong length -> DSPFatal
            out = np.zeros(max(n, 1), dtype=dt)
            with np.errstate(divide="ignore", invalid="ignore"):
                try:
                    fatal = run_body(m.t0_filter, float(dt(rise)), float(dt(fall)), out)
                except ZeroDivisionError:
                    continue  # (fall == 0: Python raises; not a DSPFatal)
            b.add(f"{tag}_t0{k}", "t0_filter", tag, {"kernel": out}, {"rise": float(rise), "fall": float(fall)}, fatal)
        for k, n in enumerate([2, 3, 12, 133]):
            out = np.zeros(n, dtype=dt)
            fatal = run_body(m.moving_slope, out)
            b.add(f"{tag}_slope{k}", "moving_slope", tag, {"kernel": out}, {"length": n}, fatal)
    b.save()


def gen_arithmetic():
    """own seed: added after the other books, which must not change"""
    rng = np.random.default_rng(0xA717)
    b = Book("arithmetic")
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        cases = [(np.array([1, 2, 3, 4, 5.0], dtype=dt), 4.0), (np.array([10, 20, 30, 40, 50.0], dtype=dt), 10.0),
                 (np.array([1, 2, 3, 4, 5.0], dtype=dt), 100.0), (np.array([1, 2, np.nan, 4, 5.0], dtype=dt), 4.0),
                 (np.array([1, 2, 3, 4, 5.0], dtype=dt), np.nan), (np.array([7.5], dtype=dt), 8.0),
                 (np.array([-np.inf, 1.0, 2.0], dtype=dt), 1.5), (np.array([0.0, -0.0, 1e-30], dtype=dt), 1e-20)]
        for n in (63, 64, 65, 1000, 4096, 8192):
            w = synth_waveforms(rng, 1, n, dtype=dt)[0][0]
            cases.append((w, float(np.median(w))))
            cases.append((w, float(w.min())))
            cases.append((w, float(w.max()) + 1.0))
        wide = (rng.standard_normal(3000) * 10.0 ** rng.uniform(-6, 6, 3000)).astype(dt)
        cases.append((wide, 0.0))
        for k, (w, thr) in enumerate(cases):
            out, fatal = call_mean_below(w, thr)
            b.add(f"{tag}_case{k}", "mean_below_threshold", tag, {"w_in": w, "result": out}, {"threshold": float(dt(thr))}, fatal)
    b.save()


def gen_fir(rng):
    b = Book("energy_kernels")
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        geos = [(30.0, 10, 400.0, 200), (1250.0, 188, 28125.0, 5792), (8.0, 0, 50.0, 65), (12.5, 3, 1716.28, 128)]
        for k, (s, f, d, n) in enumerate(geos):
            for name in ("cusp_filter", "zac_filter"):
                out, fatal = call_kernel_gen(name, s, f, d, n, dt)
                b.add(f"{tag}_{name}_geo{k}", name, tag, {"kernel": out}, {"sigma": s, "flat": f, "decay": d, "length": n}, fatal)
        for name in ("cusp_filter", "zac_filter"):
            for j, (s, f, d) in enumerate(((-1.0, 3, 10.0), (5.0, -1, 10.0), (5.0, 2.5, 10.0), (5.0, 3, -2.0))):
                out, fatal = call_kernel_gen(name, s, f, d, 64, dt)
                b.add(f"{tag}_{name}_fatal{j}", name, tag, {"kernel": out}, {"sigma": s, "flat": f, "decay": d, "length": 64}, fatal)
    b.save()

    b = Book("convolutions")
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        w, bl, _ = synth_waveforms(rng, 3, 512, dtype=dt)
        x = (w - bl[:, None]).astype(dt)
        kern, _ = call_kernel_gen("zac_filter", 30.0, 10, 400.0, 129, dt)
        kshort = rng.standard_normal(31).astype(dt)
        for kname, k in (("zac129", kern), ("rand31", kshort)):
            for mode, olen in (("v", 512 - len(k) + 1), ("s", 512), ("f", 512 + len(k) - 1)):
                out = np.stack([call_convolve(x[r], k, mode, olen)[0] for r in range(3)])
                b.add(f"{tag}_convolve_{kname}_{mode}", "convolve_wf", tag, {"w_in": x, "kernel": k, "w_out": out}, {"mode": mode})
                out, fatal = call_fft_convolve(x, k, mode, olen)
                b.add(f"{tag}_fft_{kname}_{mode}", "fft_convolve_wf", tag, {"w_in": x, "kernel": k, "w_out": out}, {"mode": mode}, fatal)
        # ICPC geometry: 5792 taps over wf[:6092] of an 8192-sample waveform -> 301 outputs
        if dt is np.float32:
            w8, bl8, _ = synth_waveforms(rng, 2, 8192, dtype=dt)
            x8 = (w8 - bl8[:, None]).astype(dt)
            for name in ("cusp_filter", "zac_filter"):
                k, _ = call_kernel_gen(name, 1250.0, 188, 28125.0, 5792, dt)
                out = np.stack([call_convolve(np.ascontiguousarray(x8[r, :6092]), k, "v", 301)[0] for r in range(2)])
                b.add(f"{tag}_icpc_{name}", "convolve_wf", tag, {"w_in": x8, "kernel": k, "w_out": out},
                      {"mode": "v", "slice_stop": 6092}, note="input is w_in[:, :6092]")
        # NaN + fatal behaviour
        xn = x[0].copy()
        xn[100] = np.nan
        out, fatal = call_convolve(xn, kshort, "v", 512 - 31 + 1)
        b.add(f"{tag}_convolve_nan_in", "convolve_wf", tag, {"w_in": xn[None], "kernel": kshort, "w_out": out[None]}, {"mode": "v"}, fatal)
        kn = kshort.copy()
        kn[3] = np.nan
        out, fatal = call_convolve(x[0], kn, "s", 512)
        b.add(f"{tag}_convolve_nan_kernel", "convolve_wf", tag, {"w_in": x[:1], "kernel": kn, "w_out": out[None]}, {"mode": "s"}, fatal)
        xb = x.copy()
        xb[1, 7] = np.nan
        out, fatal = call_fft_convolve(xb, kshort, "s", 512)
        b.add(f"{tag}_fft_nan_row", "fft_convolve_wf", tag, {"w_in": xb, "kernel": kshort, "w_out": out}, {"mode": "s"}, fatal)
        for j, (wl, kl, mode, olen) in enumerate(((16, 31, "v", 16), (64, 31, "v", 33), (64, 31, "s", 63), (64, 31, "f", 64), (64, 31, "x", 64))):
            out, fatal = call_convolve(np.ones(wl, dtype=dt), np.ones(kl, dtype=dt), mode, olen)
            b.add(f"{tag}_convolve_fatal{j}", "convolve_wf", tag, {"w_in": np.ones((1, wl), dtype=dt), "kernel": np.ones(kl, dtype=dt),
                                                                  "w_out": out[None]}, {"mode": mode}, fatal)
    b.save()


def gen_chains(rng):
    """Chain-level fixtures = the processor sequence ProcessingChain would run (pc.py:1144-1163)."""
    b = Book("chains")
    dt = np.float32
    # C1: pole_zero -> trap_filter, 8 x 1024
    w, bl, _ = synth_waveforms(rng, 8, 1024)
    pz = np.stack([call_pole_zero(w[r], 1716.28, dt)[0] for r in range(8)])
    tr = np.stack([call_trap("trap_filter", pz[r], 64, 16)[0] for r in range(8)])
    b.add("c1_pz_trap", "chain_c1", "f32", {"waveform": w, "wf_pz": pz, "wf_trap": tr}, {"tau": 1716.28, "rise": 64, "flat": 16})
    # C2: bl_subtract -> pole_zero -> trap_filter -> fixed_time_pickoff('l'), 12 x 4096 (+1 NaN row)
    n = 12
    w, bl, t0 = synth_waveforms(rng, n, 4096)
    w[5, 4000] = np.nan
    t_pick = (t0 + 625 + 0.8 * 188).astype(np.float32)
    t_pick[7] = np.float32(4096.5)  # out of range -> NaN
    t_pick[8] = np.float32(np.floor(t_pick[8]))  # exact integer hit
    blsub = np.stack([call_bl_subtract(w[r], bl[r], dt)[0] for r in range(n)])
    pz = np.empty_like(blsub)
    tr = np.empty_like(blsub)
    e = np.empty(n, dtype=dt)
    for r in range(n):
        pz[r], _ = call_pole_zero(blsub[r], 1716.28, dt)
        tr[r], _ = call_trap("trap_filter", pz[r], 625, 188)
        e[r], _ = call_pickoff(tr[r], t_pick[r], "l", dt)
    b.add("c2_energy", "chain_c2", "f32", {"waveform": w, "baseline": bl, "t_pick": t_pick, "wf_pz": pz[:2], "wf_trap": tr[:2],
                                           "trapEftp": e}, {"tau": 1716.28, "rise": 625, "flat": 188, "mode": "l"})
    # C5: int16 input -> double_pole_zero -> asym_trap -> min_max -> time_point_thresh (DWT fixture is separate)
    n = 4
    wf, blf, _ = synth_waveforms(rng, n, 8192, bl=(-15000, -14000), amp=(2000, 20000))
    wi = np.rint(wf).astype(np.int16)
    wfl = wi.astype(np.float32)  # ProcessorManager: int16 selects the float32 loop, NumPy casts (pc.py:1565-1572)
    dpz = np.stack([call_double_pole_zero(wfl[r] , 1716.28, 62.5, 0.02, dt)[0] for r in range(n)])
    at = np.stack([call_trap("asym_trap_filter", dpz[r], 8, 4, 125)[0] for r in range(n)])
    mm = np.stack([call_min_max(at[r])[0] for r in range(n)])
    thr = (0.1 * mm[:, 3]).astype(np.float32)
    tp0 = np.array([call_tpt(at[r], thr[r], mm[r, 1], 0, dt)[0] for r in range(n)], dtype=np.float32)
    b.add("c5_int16", "chain_c5", "f32", {"waveform": wi, "thr": thr, "wf_pz": dpz[:1], "wf_atrap": at[:1], "min_max": mm, "tp_0": tp0},
          {"tau1": 1716.28, "tau2": 62.5, "frac": 0.02, "rise": 8, "flat": 4, "fall": 125})
    b.save()


def gen_dwt():
    """Runs under /opt/conda/bin/python3.9 (PyWavelets 1.1.1, numpy 1.26): calls pywt.downcoef exactly as
    the reference's discrete_wavelet_transform does (processors/dwt.py:81)."""
    import pywt

    rng = np.random.default_rng(0xD5BEED + 7)
    b = Book("dwt")
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        cases = [(np.ones(16, dtype=dt), 2), (rng.standard_normal(8192).astype(dt) * 50 + 1000, 5),
                 (rng.standard_normal(256).astype(dt), 1), (rng.standard_normal(256).astype(dt), 3),
                 (rng.standard_normal(100).astype(dt), 2), (rng.standard_normal(37).astype(dt), 3),
                 (rng.standard_normal(1024).astype(dt) * 1e3, 10)]
        for k, (w, level) in enumerate(cases):
            for wt, wname in (("h", "haar"), ("d", "db1")):
                for part in "ad":
                    out = pywt.downcoef(part, w, wname, level=level)
                    assert out.dtype == dt
                    b.add(f"{tag}_case{k}_{wt}_{part}", "discrete_wavelet_transform", tag, {"w_in": w, "w_out": out},
                          {"level": level, "wave_type": wt, "coeff": part})
    b.index.append({"case": "__meta__", "pywt": pywt.__version__, "numpy": np.__version__})
    b.save()


def gen_min_max_norm():
    """min_max_norm (min_max.py:85-140): the reference's test cases (tests/processors/test_min_max_norm.py), zero / NaN / negative
    bounds, synthetic waveforms normalised by their own extremes.  a_min / a_max are 1-element arrays in the gufunc ("float32[:]")."""
    rng = np.random.default_rng(0x3A3)
    b = Book("min_max_norm")
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        ones = np.ones(10, dtype=dt)
        wn = ones.copy()
        wn[4] = np.nan
        ramp = np.linspace(-3, 7, 64).astype(dt)
        cases = [(wn, 1, 1), (ones, 0, 0), (ones, -1, 2), (ones, -2, 1), (ramp, -3, 7), (ramp, -7, 3), (ramp, 0, 5), (ramp, -5, 0),
                 (ramp, np.nan, 2), (ramp, -2, np.nan), (ramp, np.nan, 0), (ramp, 0, np.nan), (ramp, -0.0, 3), (ramp, 2, 2), (ramp, -2, 2),
                 (ramp, np.inf, 1), (ramp, 1e-30, 1e-30)]
        for k, (w, lo, hi) in enumerate(cases):
            out = np.empty_like(w)
            m = _ref("min_max")
            fatal = run_body(m.min_max_norm, w, np.array([lo], dtype=dt), np.array([hi], dtype=dt), out)
            b.add(f"{tag}_case{k}", "min_max_norm", tag, {"w_in": w, "w_out": out}, {"a_min": float(lo), "a_max": float(hi)}, fatal)
        for r in range(3):
            w, _t0 = _pz_step(rng, 1024, dt)
            w = w - dt(np.median(w))
            mm, _ = call_min_max(w)
            out = np.empty_like(w)
            fatal = run_body(_ref("min_max").min_max_norm, w, np.array([mm[2]], dtype=dt), np.array([mm[3]], dtype=dt), out)
            b.add(f"{tag}_synth{r}", "min_max_norm", tag, {"w_in": w, "w_out": out}, {"a_min": float(mm[2]), "a_max": float(mm[3])}, fatal)
    b.save()


def gen_itpt():
    """interpolated_time_point_thresh (time_point_thresh.py:95-222): the reference's own test cases (tests/processors/
    test_time_point_thresh.py:118-218), every mode on both walks, starts outside the waveform and between samples, no crossing,
    an unknown mode with and without a crossing, synthetic trapezoid edges."""
    rng = np.random.default_rng(0x17A7)
    b = Book("interpolated_time_point_thresh")
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        saw = np.concatenate([np.arange(-1, 5, 1), np.arange(-1, 5, 1)]).astype(dt)
        sn = saw.copy()
        sn[4] = np.nan
        cases = [(sn, 1, 11, 0, "i"), (saw, np.nan, 11, 0, "i"), (saw, 1, np.nan, 0, "i"), (saw, 1, 12, 0, "i"), (saw, 1, -1, 1, "i"),
                 (saw, 1, 11.7, 0, "l"), (saw, 3, 0.9, 1, "l"), (saw, 10, 11, 0, "l"), (saw, 10, 0, 1, "l"), (saw, 1, 11, 0, "x"),
                 (saw, 10, 11, 0, "x"), (saw, 1, 2, 0, "i"), (saw, 0, 1, 0, "i"), (saw, 4, 10, 1, "i"), (saw, 4, 11, 1, "i")]
        for mode in "ibcafrnl":
            cases += [(saw, 1, 11, 0, mode), (saw, 3, 0, 1, mode), (saw, 1.5, 11, 0, mode), (saw, 3.5, 0, 1, mode), (saw, 2.25, 11, 0, mode),
                      (saw, 0.75, 3, 1, mode), (-saw, -1.5, 11, 0, mode), (-saw, -3.25, 0, 1, mode)]
        for k, (w, thr, ts, wf, mode) in enumerate(cases):
            out, fatal = call_itpt(w, thr, ts, wf, mode, dt)
            b.add(f"{tag}_case{k}_{mode}", "interpolated_time_point_thresh", tag, {"w_in": w, "t_out": out},
                  {"a_threshold": float(thr), "t_start": float(ts), "walk_forward": int(wf), "mode": mode}, fatal)
        for r in range(3):
            w, t0 = _pz_step(rng, 1024, dt)
            at, _ = call_trap("asym_trap_filter", w, 8, 4, 125)
            mm, _ = call_min_max(at)
            for thr_scale, walk in ((0.05, 0), (0.5, 0), (0.5, 1), (0.93, 0)):
                for mode in "lrn":
                    thr = dt(thr_scale * mm[3])
                    ts = mm[1] if walk == 0 else 0
                    out, fatal = call_itpt(at, thr, ts, walk, mode, dt)
                    b.add(f"{tag}_synth{r}_{thr_scale:g}_{walk}_{mode}", "interpolated_time_point_thresh", tag, {"w_in": at, "t_out": out},
                          {"a_threshold": float(thr), "t_start": float(ts), "walk_forward": int(walk), "mode": mode}, fatal)
    b.save()


def main():
    if "--dwt" in sys.argv:
        gen_dwt()
        return
    _install_stubs()
    if "--arithmetic" in sys.argv:
        gen_arithmetic()
        return
    if "--windows" in sys.argv:
        gen_windows()
        return
    if "--kernels" in sys.argv:
        gen_kernels()
        return
    if "--current" in sys.argv:
        gen_current()
        return
    if "--lsf" in sys.argv:
        gen_linear_slope_fit()
        return
    if "--itpt" in sys.argv:
        gen_itpt()
        return
    if "--norm" in sys.argv:
        gen_min_max_norm()
        return
    rng = np.random.default_rng(0xD5BEED)
    gen_elementwise(rng)
    gen_pole_zero(rng)
    gen_traps(rng)
    gen_pickoff(rng)
    gen_tpt(rng)
    gen_fir(rng)
    gen_chains(rng)
    gen_arithmetic()
    gen_windows()
    gen_kernels()
    gen_current()
    gen_linear_slope_fit()
    gen_itpt()
    gen_min_max_norm()


if __name__ == "__main__":
    main()
