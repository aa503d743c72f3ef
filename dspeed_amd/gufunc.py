"""``HipGUFunc``: the processor-object protocol of the reference, backed by the HIP library.

``ProcessorManager`` in the reference only needs ``signature``, ``types``, ``nin``, ``nout``, ``__name__`` and
``__call__(*inputs, *outputs)`` with outputs written in place (processing_chain.py:1527-1543, 1778-1781); the
reference's own non-numba implementation of that protocol is ``GUFuncWrapper`` (utils.py:12-163).  This class is
the device-side counterpart: same attributes, same calling convention, NumPy arrays *or* ``DeviceArray``s.

NumPy arguments take the host path (H2D -> kernel -> D2H); ``DeviceArray`` arguments stay on the device.
There is no CPU implementation behind these objects: without the library or a GPU the call raises.
"""
from __future__ import annotations

import ctypes as C
import re

import numpy as np

from . import _lib
from .device import DeviceArray, dtype_code


def _parse_signature(sig: str):
    """'(n),()->(n)' -> ([('n',), ()], [('n',)]);  a signature without '->' has only in-place arguments."""
    def dims(s):
        return [tuple(x.strip() for x in m.split(",") if x.strip()) for m in re.findall(r"\(([^)]*)\)", s)]

    if "->" in sig:
        a, b = sig.split("->")
        return dims(a), dims(b)
    return dims(sig), []


class HipGUFunc:
    def __init__(self, name: str, signature: str, types: list[str], impl, doc: str = ""):
        self.__name__ = name
        self.signature = signature
        self.types = list(types)
        self.in_dims, self.out_dims = _parse_signature(signature)
        # like numba gufuncs built from a "(n),(m),(),(p)" layout: trailing args are outputs written in place
        self.nin = len(self.in_dims)
        self.nout = len(self.out_dims)
        self.nargs = self.nin + self.nout
        self.ntypes = len(self.types)
        self._impl = impl
        self.__doc__ = doc

    def __repr__(self):
        return f"<HipGUFunc {self.__name__} {self.signature}>"

    def __call__(self, *args, **kwargs):
        per_row = self._per_row_integers(args)
        if not per_row:
            return self._impl(self, *args, **kwargs)
        return self._by_parameter_value(per_row, args, kwargs)

    def _per_row_integers(self, args):
        """positions of integer parameters ('i' in the type string, dimension ()) given as one value per waveform -- the gufunc broadcasts them
        like any other argument (reference processors: ``trap_filter(w_in, rise, flat, w_out)`` with rise / flat arrays)"""
        sig = self.types[0].replace("->", "")
        return [k for k in range(min(self.nin, len(args))) if sig[k] == "i" and self.in_dims[k] == () and not isinstance(args[k], DeviceArray)
                and np.ndim(args[k]) >= 1 and np.size(args[k]) > 1]

    def _by_parameter_value(self, per_row, args, kwargs):
        """The kernels take an integer parameter (a filter length, a level) as a constant of the launch: rows are grouped by the values they
        ask for, each group runs with its constants, and the results go back to the rows' places."""
        cols = [np.asarray(args[k]).reshape(-1) for k in per_row]
        n_rows = cols[0].size
        if any(c.size != n_rows for c in cols):
            raise ValueError(f"{self.__name__}: per-waveform parameters of different lengths")
        if any(isinstance(a, DeviceArray) for a in args):
            raise NotImplementedError(f"{self.__name__}: per-waveform integer parameters need the rows in host memory (they are grouped by value)")
        if any(np.isnan(c.astype(np.float64)).any() for c in cols):
            raise NotImplementedError(f"{self.__name__}: NaN integer parameter")
        table = np.stack([c.astype(np.int64) for c in cols], axis=1)
        values, which = np.unique(table, axis=0, return_inverse=True)
        which = which.reshape(-1)
        args = [np.asarray(a) if a is not None and k < self.nin else a for k, a in enumerate(args)]
        rowwise = [isinstance(a, np.ndarray) and a.ndim >= 1 and a.shape[0] == n_rows for a in args]
        if not rowwise[0]:
            raise ValueError(f"{self.__name__}: {n_rows} parameter values for a waveform argument of shape {np.shape(args[0])}")
        given = list(args[self.nin:]) + [None] * (self.nargs - len(args))
        results = [o if isinstance(o, np.ndarray) else None for o in given]
        for g, vals in enumerate(values):
            rows = np.flatnonzero(which == g)
            part = [a[rows] if rw and k < self.nin else a for k, (a, rw) in enumerate(zip(args[: self.nin], rowwise))]
            for k, v in zip(per_row, vals):
                part[k] = int(v)
            got = self._impl(self, *part, **kwargs)
            got = got if isinstance(got, tuple) else (got,)
            for j, r in enumerate(got):
                r = np.asarray(r)
                if results[j] is None:
                    results[j] = np.empty((n_rows, *r.shape[1:]), dtype=r.dtype)
                results[j][rows] = r
        return results[0] if len(results) == 1 else tuple(results)


# ---------------------------------------------------------------------------------------------------------------
# helpers used by the processor implementations
# ---------------------------------------------------------------------------------------------------------------
def loop_suffix(wf_dtype) -> str:
    """Which gufunc loop an input waveform dtype selects (first castable signature wins, processing_chain.py:1565-1572,
    1654-1664): (u)int16 and float32 -> the float32 loop; (u)int32, float64 -> the float64 loop."""
    dt = np.dtype(wf_dtype)
    if dt in (np.dtype(np.float32), np.dtype(np.int16), np.dtype(np.uint16)):
        return "f32"
    if dt in (np.dtype(np.float64), np.dtype(np.int32), np.dtype(np.uint32)):
        return "f64"
    raise TypeError(f"waveform dtype {dt} matches no loop of this processor")


def entry(name: str, sfx: str):
    fn = getattr(_lib.lib(), f"dsp_{name}_{sfx}", None)
    if fn is None:
        raise NotImplementedError(f"{name}: the float64 loop is not implemented on the device (float32 / int16 / uint16 "
                                  "waveforms run the float32 loop); dspeed_amd has no CPU fallback")
    return fn


class Staging:
    """Moves gufunc arguments to the device for one call and brings outputs back."""

    def __init__(self):
        self.keep = []
        self.copy_back = []

    def wf_in(self, a, min_ndim=2):
        """Waveform input -> (device ptr, dtype code, n_wf, wf_len, row stride, was_1d)."""
        if isinstance(a, DeviceArray):
            shape = a.shape
            d = a
        else:
            a = np.asarray(a)
            if a.dtype == np.bool_ or a.dtype.kind not in "fiu":
                raise TypeError(f"unsupported waveform dtype {a.dtype}")
            shape = a.shape
            d = DeviceArray.from_numpy(a if a.ndim >= 1 else a.reshape(1))
            self.keep.append(d)
        if len(shape) == 1:
            n_wf, n = 1, shape[0]
        elif len(shape) == 2:
            n_wf, n = shape
        else:
            raise ValueError("waveform blocks must be 1-D or 2-D")
        return d.ptr, dtype_code(d.dtype), int(n_wf), int(n), int(n), len(shape) == 1

    def scalar_in(self, v, n_wf, dt):
        """Scalar argument '()' -> (device column pointer or None, broadcast value)."""
        if isinstance(v, DeviceArray):
            if v.dtype != np.dtype(dt):
                raise TypeError(f"per-waveform scalar column must be {np.dtype(dt)}")
            return v.ptr, 0.0
        a = np.asarray(v)
        if a.ndim == 0 or a.size == 1 and n_wf != 1:
            return None, float(a.reshape(-1)[0])
        if a.shape != (n_wf,):
            if a.size == 1:
                return None, float(a.reshape(-1)[0])
            raise ValueError(f"scalar argument has shape {a.shape}, expected () or ({n_wf},)")
        d = DeviceArray.from_numpy(a.astype(dt))
        self.keep.append(d)
        return d.ptr, 0.0

    def out(self, given, shape, dt):
        """Output argument: use the caller's array (in place) or allocate.  Returns (device ptr, python result object)."""
        if isinstance(given, DeviceArray):
            if tuple(given.shape) != tuple(shape) and int(np.prod(given.shape)) != int(np.prod(shape)):
                raise ValueError("Outputs are not the right shape")
            return given.ptr, given
        if given is None:
            host = np.empty(shape, dtype=dt)
        else:
            host = given
            if not isinstance(host, np.ndarray):
                raise TypeError("output arguments must be NumPy arrays or DeviceArrays")
            if int(np.prod(host.shape)) != int(np.prod(shape)):
                raise ValueError("Outputs are not the right shape")
        d = DeviceArray(shape, dt)
        self.keep.append(d)
        self.copy_back.append((d, host))
        return d.ptr, host

    def finish(self):
        """Copy outputs back to the caller's arrays, then release the staging buffers (idempotent)."""
        pending, self.copy_back = self.copy_back, []
        try:
            for d, host in pending:
                tmp = d.to_numpy()
                host[...] = tmp.reshape(host.shape).astype(host.dtype, copy=False)
        finally:
            self.release()

    def release(self):
        self.copy_back = []
        for d in self.keep:
            d.free()
        self.keep.clear()


def run(fn, what, *cargs):
    row = C.c_int64(-1)
    rc = fn(*cargs, None, C.byref(row))
    _lib.check(rc, row=row.value if row.value >= 0 else None, what=what)
