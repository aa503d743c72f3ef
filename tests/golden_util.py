"""Loader for the committed fixtures in tests/golden (produced by oracle/gen_golden.py)."""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Case:
    def __init__(self, entry, npz):
        self.name = entry["case"]
        self.kernel = entry["kernel"]
        self.dtype = np.float32 if entry["dtype"] == "f32" else np.float64
        self.tag = entry["dtype"]
        self.params = entry["params"]
        self.fatal = entry["fatal"]
        self.note = entry.get("note", "")
        self._npz = npz
        self._arrays = entry["arrays"]

    def __getitem__(self, key):
        return self._npz[f"{self.name}/{key}"]

    def __contains__(self, key):
        return key in self._arrays

    def __repr__(self):
        return f"<golden {self.kernel}:{self.name}>"


def load(book):
    npz = np.load(os.path.join(GOLDEN_DIR, book + ".npz"))
    index = json.loads(str(npz["__index__"]))
    return [Case(e, npz) for e in index if e["case"] != "__meta__"]


def cases(book, kernel=None, tag=None):
    out = load(book)
    if kernel:
        out = [c for c in out if c.kernel == kernel]
    if tag:
        out = [c for c in out if c.tag == tag]
    return out


def zerodiv(c):
    """Cases where numba (error_model='python') raises ZeroDivisionError while the plain-NumPy body returns inf/nan."""
    p = c.params
    if c.kernel == "trap_norm":
        return p.get("rise") == 0
    if c.kernel == "asym_trap_filter":
        return p.get("rise") == 0 or p.get("fall") == 0
    return False


def assert_rel_to_peak(got, want, tol, what=""):
    """|got - want| <= tol * max|want| per row, NaN positions identical."""
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, f"{what}: shape {got.shape} != {want.shape}"
    nan_w = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan_w), f"{what}: NaN pattern differs"
    if nan_w.all():
        return 0.0
    g = np.where(nan_w, 0, got).astype(np.float64)
    w = np.where(nan_w, 0, want).astype(np.float64)
    if w.ndim == 1:
        g, w = g[None], w[None]
    peak = np.max(np.abs(w), axis=-1, keepdims=True)
    peak = np.where(peak == 0, 1.0, peak)
    err = np.max(np.abs(g - w) / peak)
    assert err <= tol, f"{what}: max |diff|/peak = {err:.3e} > {tol:g}"
    return err


def recipe_kernel(which):
    """The kernels the Ge recipes generate once (t0 filter 8 + 125 samples; cusp / zero-area cusp with sigma 1250, flat top 188, decay 28 125
    samples, 5792 taps), as the REFERENCE's own generator bodies made them (oracle/gen_golden.py: kernels.npz f32_t00, energy_kernels.npz
    f32_*_geo1) -- so that a whole-recipe test does not build its expectation with the product's generators."""
    book, case = {"t0": ("kernels", "f32_t00"), "cusp": ("energy_kernels", "f32_cusp_filter_geo1"), "zac": ("energy_kernels", "f32_zac_filter_geo1")}[which]
    c = next(c for c in load(book) if c.name == case)
    assert not c.fatal
    return np.ascontiguousarray(c["kernel"], dtype=np.float32)
