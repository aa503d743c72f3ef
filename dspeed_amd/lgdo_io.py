"""LGDO / LH5 containers on either side of the chain (reference src/dspeed/processing_chain.py:1984-2360 ``LGDO*IOManager``,
src/dspeed/build_dsp.py:256-266 and :399-432 ``LH5Iterator`` -> chain -> ``LH5Store.write``).

``lgdo`` and ``h5py`` are optional here: everything below works on the *protocol* of the LGDO types -- ``Array`` /
``ArrayOfEqualSizedArrays`` expose ``.nda`` (+ ``.attrs``), ``VectorOfVectors`` ``.flattened_data.nda`` and ``.cumulative_length.nda``,
``WaveformTable`` ``.values`` / ``.dt`` / ``.t0`` (+ ``dt_units`` / ``t0_units``), ``Table`` is a mapping of those, ``LH5Iterator`` an
iterable of ``Table`` chunks with ``current_i_entry`` and ``reset_field_mask`` -- so the same code serves the real classes where the
packages are installed and duck-typed stand-ins where they are not (tests/test_lgdo_io_cpu.py).  Opening an LH5 *file* needs the real
``lgdo.lh5``; asking for that without the package raises ImportError with the way out (NumPy tables, ``.npz``).
"""
from __future__ import annotations

import queue
import threading
from collections.abc import Mapping

import numpy as np

from .errors import DSPFatal, ProcessingChainError

#: time units the waveform grids understand -> nanoseconds (what pint resolves for the reference, processing_chain.py:2281-2303)
TIME_NS = {"ns": 1.0, "nanosecond": 1.0, "nanoseconds": 1.0, "us": 1e3, "µs": 1e3, "microsecond": 1e3, "microseconds": 1e3, "ms": 1e6,
           "millisecond": 1e6, "milliseconds": 1e6, "s": 1e9, "second": 1e9, "seconds": 1e9, "ps": 1e-3, "picosecond": 1e-3}


def lgdo_or_none():
    try:
        import lgdo  # noqa: F401
        from lgdo import lh5  # noqa: F401
    except Exception:
        return None
    return lgdo


def require_lh5(what: str):
    lg = lgdo_or_none()
    if lg is None:
        raise ImportError(f"{what} needs the 'lgdo' package (legend-pydataobj, with h5py), which is not installed here. "
                          "dspeed_amd.build_dsp also takes NumPy tables, LGDO-like tables in memory and '.npz' files laid out like an LH5 file.")
    from lgdo import lh5

    return lg, lh5


def _units(obj, name=None):
    if name is not None and getattr(obj, name, None) is not None:
        return getattr(obj, name)
    attrs = getattr(obj, "attrs", None)
    return attrs.get("units") if isinstance(attrs, Mapping) else None


def is_waveform_table(obj) -> bool:
    """values / dt / t0 that are themselves LGDO arrays (this package's own WaveformInput has plain numbers and ndarrays there)"""
    if isinstance(obj, (np.ndarray, Mapping)) or not all(hasattr(obj, a) for a in ("values", "dt", "t0")):
        return False
    return hasattr(obj.dt, "nda") and (hasattr(obj.values, "nda") or is_vov(obj.values))


def is_vov(obj) -> bool:
    return hasattr(obj, "flattened_data") and hasattr(obj, "cumulative_length")


def is_lgdo_column(obj) -> bool:
    return is_waveform_table(obj) or is_vov(obj) or (hasattr(obj, "nda") and not isinstance(obj, np.ndarray))


def is_lgdo_table(obj) -> bool:
    """a Table / Struct: a mapping (or something with keys()) whose members are LGDO columns"""
    if isinstance(obj, np.ndarray) or not hasattr(obj, "keys"):
        return False
    try:
        keys = list(obj.keys())
    except Exception:
        return False
    return bool(keys) and any(is_lgdo_column(obj[k]) for k in keys)


def is_chunk_iterator(obj) -> bool:
    """an LH5Iterator: iterable over Table chunks that knows where it is in the file"""
    return hasattr(obj, "__iter__") and hasattr(obj, "current_i_entry") and not isinstance(obj, Mapping)


class RaggedColumn:
    """A VectorOfVectors as the chain sees it (processing_chain.py:2198-2232): rows padded to a common length -- NaN for floats, 0 for
    integers -- and the number of valid samples of every row, the variable the reference calls ``len(<name>)``."""

    def __init__(self, padded: np.ndarray, lengths: np.ndarray, unit=None):
        self.padded, self.lengths, self.unit = padded, lengths, unit

    def __len__(self):
        return len(self.padded)

    @staticmethod
    def from_vov(vov, max_len: int | None = None) -> "RaggedColumn":
        flat = np.asarray(vov.flattened_data.nda)
        cl = np.asarray(vov.cumulative_length.nda).astype(np.int64)
        n = len(cl)
        starts = np.concatenate([[0], cl[:-1]]) if n else np.zeros(0, dtype=np.int64)
        lengths = (cl - starts).astype(np.uint32)
        if max_len is None:  # the reference's fall-back: twice the longest vector of the first batch (:2216-2226)
            max_len = int(2 * lengths.max()) if n else 0
        if n and int(lengths.max()) > max_len:
            raise DSPFatal("VectorOfVectors entry has length larger than array variable length")
        fill = 0 if np.issubdtype(flat.dtype, np.integer) else np.nan
        padded = np.full((n, max_len), fill, dtype=flat.dtype)
        if n:
            col = np.arange(max_len)[None, :]
            mask = col < lengths[:, None]
            padded[mask] = flat[: int(cl[-1])]  # (row-major order of the mask = the order of the flattened data)
        return RaggedColumn(padded, lengths, _units(vov))

    def to_flat(self):
        """-> (flattened_data, cumulative_length) the way ``VectorOfVectors._set_vector_unsafe`` lays them out (:2250-2255)"""
        lengths = np.minimum(self.lengths.astype(np.int64), self.padded.shape[1])
        mask = np.arange(self.padded.shape[1])[None, :] < lengths[:, None]
        return self.padded[mask], np.cumsum(lengths).astype(np.uint32)


def column(obj, max_len: int | None = None):
    """LGDO column -> what the chain takes: ndarray, WaveformInput (values + sampling period + time of sample 0, in ns) or RaggedColumn"""
    from .processing_chain import WaveformInput

    if is_waveform_table(obj):
        dt_u, t0_u = _units(obj.dt) if _units(obj, "dt_units") is None else obj.dt_units, _units(obj.t0) if _units(obj, "t0_units") is None else obj.t0_units
        if dt_u is None:
            dt_u = t0_u
        if t0_u is None:
            t0_u = dt_u
        values = obj.values
        vals = RaggedColumn.from_vov(values, max_len) if is_vov(values) else np.asarray(values.nda)
        if isinstance(dt_u, str) and dt_u in TIME_NS and isinstance(t0_u, str) and t0_u in TIME_NS:
            dt = np.asarray(obj.dt.nda).reshape(-1)
            t0 = np.asarray(obj.t0.nda)
            period = float(dt[0]) * TIME_NS[dt_u] if len(dt) else 1.0  # one sampling period per table (reference: wf_table.dt[0])
            t0_ns = t0.astype(np.float64 if t0.dtype.itemsize > 4 or t0.dtype.kind in "iu" else np.float32) * TIME_NS[t0_u]
            t0_arg = t0_ns if t0_ns.size != 1 else float(t0_ns.reshape(-1)[0])
            if isinstance(vals, RaggedColumn):  # variable-length waveforms (:2327-2328): padded rows on the grid + the variable len(<name>)
                wf = WaveformInput(vals.padded, period, t0_arg)
                wf.lengths = vals.lengths
                return wf
            return WaveformInput(vals, period, t0_arg)
        return vals  # no usable time units: a plain array without a coordinate grid (:2304-2308)
    if is_vov(obj):
        return RaggedColumn.from_vov(obj, max_len)
    if hasattr(obj, "nda"):
        return np.asarray(obj.nda)
    return obj


def table_columns(tb, fields=None) -> dict:
    """LGDO Table -> {name: column}; a VectorOfVectors ``v`` also yields its length column ``len(v)``"""
    out = {}
    for k in tb.keys():
        if fields is not None and k not in fields:
            continue
        c = column(tb[k])
        if isinstance(c, RaggedColumn):
            out[k] = c.padded
            out[f"len({k})"] = c.lengths
        else:
            out[k] = c
            if getattr(c, "lengths", None) is not None:  # a WaveformTable of variable-length waveforms
                out[f"len({k})"] = c.lengths
    return out


#: callables ``(file, group, iterator=False, n_rows=None, **row selection) -> table | chunk iterator | None`` tried in turn before the
#: built-in ways of opening an auxiliary input (other file formats, tests)
FRIEND_OPENERS: list = []


def open_friend(file, group, iterator: bool = False, n_rows: int | None = None, **selection):
    """The table ``group`` of ``file`` that a recipe's ``inputs`` joins to the table being processed (reference build_dsp.py:304-330): a
    chunk iterator over it (``iterator=True``: what ``LH5Iterator.add_friend`` takes) or its first ``n_rows`` rows in memory.  ``.npz``
    files laid out like an LH5 file are read here; LH5 files through ``lgdo.lh5``."""
    for opener in FRIEND_OPENERS:
        got = opener(file, group, iterator=iterator, n_rows=n_rows, **selection)
        if got is not None:
            return got
    if isinstance(file, str) and file.endswith(".npz"):
        if iterator:
            raise NotImplementedError("an '.npz' friend of a chunk iterator: give the friend as an LH5 file, or the main table as arrays")
        from .build_dsp import _read_npz, _select

        tables = _read_npz(file)
        key = group if group in tables else group.strip("/")
        if key not in tables:
            raise ProcessingChainError(f"auxiliary input: no table '{group}' in {file}")
        return {k: _select(v, slice(0, n_rows)) if n_rows is not None else v for k, v in tables[key].items()}
    _, lh5 = require_lh5(f"reading the auxiliary input '{file}'")
    if iterator:
        return lh5.LH5Iterator(file, group, **selection)
    return lh5.LH5Store(keep_open=True).read(group, file, n_rows=n_rows)


#: callables ``(file, path) -> array | number | None`` tried in turn before the built-in ways of loading a constant (other formats, tests)
CONSTANT_LOADERS: list = []


def load_constant(file: str, path: str):
    """``loadlh5(file, path)`` of the recipe language: the object ``path`` of ``file`` as a constant of the chain -- an array (the taps of a
    filter) or a number (reference processing_chain.py:1444-1467).  ``.npz`` files are read here (``path`` names the array), LH5 files
    through ``lgdo.lh5``."""
    for loader in CONSTANT_LOADERS:
        got = loader(file, path)
        if got is not None:
            return got
    try:
        if isinstance(file, str) and file.endswith(".npz"):
            with np.load(file) as z:
                key = path if path in z.files else path.strip("/")
                if key not in z.files:
                    raise ValueError(f"no array '{path}'")
                got = z[key]
            return got[()] if got.ndim == 0 else got
        _, lh5 = require_lh5(f"loadlh5('{file}', '{path}')")
        obj = lh5.read(path, file)
        return obj.value if hasattr(obj, "value") else np.asarray(obj.nda)
    except (ValueError, OSError, KeyError) as exc:
        raise ProcessingChainError(f"LH5 file not found: {file}") from exc


def snapshot(cols: dict) -> dict:
    """own copies of a chunk's columns: an LH5Iterator refills the same buffers on its next read"""
    from .processing_chain import WaveformInput

    out = {}
    for k, c in cols.items():
        if isinstance(c, WaveformInput):
            out[k] = WaveformInput(np.array(c.values, copy=True), c.dt, c.t0 if isinstance(c.t0, float) else np.array(c.t0, copy=True))
            if getattr(c, "lengths", None) is not None:
                out[k].lengths = np.array(c.lengths, copy=True)
        else:
            out[k] = np.array(c, copy=True)
    return out


class Column(np.ndarray):
    """an output column with the attributes an LGDO array would carry (``.attrs``: units, lh5_attrs, description) -- what
    ``results_table`` hands out where the lgdo package is absent; ``.nda`` is the plain array, as on ``lgdo.Array``"""

    def __new__(cls, values, attrs=None):
        obj = np.asarray(values).view(cls)
        obj.attrs = dict(attrs or {})
        return obj

    def __array_finalize__(self, obj):
        self.attrs = dict(getattr(obj, "attrs", None) or {})

    @property
    def nda(self):
        return np.asarray(self)


def results_table(cols: dict, units: dict | None = None, lengths: dict | None = None, attrs: dict | None = None):
    """{name: ndarray} -> an ``lgdo.Table`` (Array / ArrayOfEqualSizedArrays / VectorOfVectors columns with their units) when lgdo is
    installed, else a dict of ``Column``s.  ``lengths``: name -> per-row lengths of a variable-length output; ``attrs``: name -> the
    attributes of the column (units, lh5_attrs, description)."""
    lg = lgdo_or_none()
    lengths, attrs = lengths or {}, {k: dict(v) for k, v in (attrs or {}).items()}
    for k, u in (units or {}).items():
        if u:
            attrs.setdefault(k, {}).setdefault("units", u)
    units = {k: v.get("units") for k, v in attrs.items()}
    if lg is None:
        out = {}
        for k, v in cols.items():
            out[k] = RaggedColumn(np.asarray(v), np.asarray(lengths[k]), units.get(k)) if k in lengths else Column(v, attrs.get(k))
        return out
    tb = lg.Table(size=len(next(iter(cols.values()))) if cols else 0)
    for k, v in cols.items():
        a = np.asarray(v)
        attrs_k = attrs.get(k, {})
        if k in lengths:
            flat, cl = RaggedColumn(a, np.asarray(lengths[k])).to_flat()
            tb.add_field(k, lg.VectorOfVectors(flattened_data=flat, cumulative_length=cl, attrs=attrs_k))
        elif a.ndim == 1:
            tb.add_field(k, lg.Array(a, attrs=attrs_k))
        else:
            tb.add_field(k, lg.ArrayOfEqualSizedArrays(nda=a, attrs=attrs_k))
    return tb


def write_back(out_tb, cols: dict, start: int = 0) -> None:
    """results into an existing LGDO-like output table (``proc_chain(tb_in, tb_out)`` of the reference, processing_chain.py:675-716):
    every column of ``cols`` that the table holds receives rows ``start ..``"""
    for k, v in cols.items():
        if k not in out_tb.keys():
            continue
        dst = out_tb[k]
        a = np.asarray(v)
        if is_waveform_table(dst):
            dst = dst.values
        if hasattr(dst, "resize") and len(dst) < start + len(a):
            dst.resize(start + len(a))
        nda = dst.nda if hasattr(dst, "nda") else dst
        if nda.shape[1:] != a.shape[1:]:
            raise ProcessingChainError(f"output column '{k}': shape {a.shape[1:]} does not fit the table's {nda.shape[1:]}")
        nda[start:start + len(a)] = a


class ChunkReader:
    """Reads the chunks of an LH5Iterator one ahead on a thread of its own (HDF5 decompression releases the GIL), so the file read of
    chunk k+1 overlaps the transfers and kernels of chunk k -- the overlap the reference's loop (build_dsp.py:399-432: read, process,
    write in turn) does not have.  Yields ``(i_entry, n_rows, columns)`` with the columns copied out of the iterator's buffers."""

    def __init__(self, iterator, fields=None, depth: int = 2):
        self._it, self._fields = iterator, fields
        self._q: queue.Queue = queue.Queue(maxsize=max(1, depth))
        self._thread = threading.Thread(target=self._run, name="dspeed-lh5-read", daemon=True)
        self._stop = False
        self._thread.start()

    def _run(self):
        try:
            for chunk in self._it:
                if self._stop:
                    break
                n = len(chunk)
                i_entry = int(getattr(self._it, "current_i_entry", 0))
                cols = snapshot(table_columns(chunk, self._fields))
                self._q.put((i_entry, n, cols))
            self._q.put(None)
        except BaseException as e:  # (the consumer re-raises it)
            self._q.put(e)

    def __iter__(self):
        while True:
            item = self._q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            yield item

    def close(self):
        self._stop = True
        while self._thread.is_alive():
            try:
                self._q.get_nowait()
            except queue.Empty:
                self._thread.join(0.05)
