"""Stand-ins with the protocol of the LGDO containers and of lh5.LH5Iterator (legend-pydataobj is not installed here): just the
attributes and methods the reference's IO managers and build_dsp touch (processing_chain.py:1984-2360, build_dsp.py:256-432)."""
import numpy as np


class Array:
    def __init__(self, nda, attrs=None):
        self.nda = np.asarray(nda)
        self.attrs = dict(attrs or {})
        self.dtype = self.nda.dtype

    def __len__(self):
        return len(self.nda)

    def resize(self, n):
        if n != len(self.nda):
            new = np.zeros((n,) + self.nda.shape[1:], dtype=self.nda.dtype)
            m = min(n, len(self.nda))
            new[:m] = self.nda[:m]
            self.nda = new


class ArrayOfEqualSizedArrays(Array):
    pass


class VectorOfVectors:
    def __init__(self, flattened_data, cumulative_length, attrs=None):
        self.flattened_data = Array(flattened_data)
        self.cumulative_length = Array(np.asarray(cumulative_length, dtype=np.uint32))
        self.attrs = dict(attrs or {})
        self.dtype = self.flattened_data.nda.dtype

    def __len__(self):
        return len(self.cumulative_length)


class WaveformTable:
    def __init__(self, values, dt, t0, dt_units="ns", t0_units="ns"):
        self.values = values if hasattr(values, "nda") or hasattr(values, "flattened_data") else ArrayOfEqualSizedArrays(values)
        n = len(self.values)
        self.dt = Array(np.full(n, dt, dtype=np.float64) if np.ndim(dt) == 0 else dt, {"units": dt_units} if dt_units else {})
        self.t0 = Array(np.full(n, t0, dtype=np.float64) if np.ndim(t0) == 0 else t0, {"units": t0_units} if t0_units else {})
        self.dt_units, self.t0_units = dt_units, t0_units
        self.attrs = {}

    def __len__(self):
        return len(self.values)


class Table(dict):
    def __len__(self):  # rows, like lgdo.Table (``if self`` would ask for the length again: count the keys through dict)
        return len(next(iter(self.values()))) if dict.__len__(self) else 0

    def join(self, other, prefix="", suffix=""):  # lgdo.Table.join: the other table's columns under prefixed / suffixed names
        for k in other.keys():
            self[f"{prefix}{k}{suffix}"] = other[k]


def _slice(col, a, b):
    if isinstance(col, WaveformTable):
        return WaveformTable(ArrayOfEqualSizedArrays(col.values.nda[a:b]), col.dt.nda[a:b], col.t0.nda[a:b], col.dt_units, col.t0_units)
    if isinstance(col, VectorOfVectors):
        cl = col.cumulative_length.nda.astype(np.int64)
        s = int(cl[a - 1]) if a > 0 else 0
        return VectorOfVectors(col.flattened_data.nda[s:int(cl[b - 1])] if b > a else col.flattened_data.nda[:0], cl[a:b] - s, col.attrs)
    return type(col)(col.nda[a:b], col.attrs)


class LH5Iterator:
    """chunks of ``buffer_len`` rows of an in-memory table, handed out in ONE buffer that the next read overwrites (like the real one)"""

    def __init__(self, table: Table, buffer_len=3200, i_start=0, n_entries=None):
        self._table, self.buffer_len, self.i_start = table, buffer_len, i_start
        self.n_entries = (len(table) - i_start) if n_entries is None else min(n_entries, len(table) - i_start)
        self.current_i_entry = 0
        self.field_mask = None
        self.reads = []
        self._buf = {}
        self.friends = []

    def add_friend(self, other, prefix="", suffix=""):
        """lh5.LH5Iterator.add_friend: the friend is read in step and its columns appear in every chunk under prefixed / suffixed names"""
        assert other.buffer_len == self.buffer_len and len(other) >= len(self)
        self.friends.append((other, prefix, suffix))

    def __len__(self):
        return self.n_entries

    def reset_field_mask(self, mask):
        self.field_mask = list(mask)

    def __iter__(self):
        if self.friends:
            own = LH5Iterator.__iter__
            streams = [iter(f) for f, _p, _s in self.friends]
            friends, self.friends = self.friends, []
            try:
                for chunk in own(self):
                    for (f, prefix, suffix), st in zip(friends, streams):
                        part = next(st)
                        for k in part.keys():
                            name = f"{prefix}{k}{suffix}"
                            if self.field_mask is None or name in self.field_mask:
                                chunk[name] = part[k]
                    yield chunk
            finally:
                self.friends = friends
            return
        self.current_i_entry = 0
        pos = 0
        while pos < self.n_entries:
            n = min(self.buffer_len, self.n_entries - pos)
            keys = [k for k in self._table if self.field_mask is None or k in self.field_mask]
            self.reads.append((pos, n, tuple(keys)))
            chunk = Table()
            for k in keys:
                part = _slice(self._table[k], self.i_start + pos, self.i_start + pos + n)
                if isinstance(part, (Array, WaveformTable)) and not isinstance(part, VectorOfVectors):
                    # the shared buffer: same arrays every time, refilled
                    tgt = part.values if isinstance(part, WaveformTable) else part
                    key = (k, tgt.nda.shape[1:], tgt.nda.dtype)
                    if key not in self._buf or len(self._buf[key]) < n:
                        self._buf[key] = np.empty((self.buffer_len,) + tgt.nda.shape[1:], dtype=tgt.nda.dtype)
                    self._buf[key][:n] = tgt.nda
                    tgt.nda = self._buf[key][:n]
                chunk[k] = part
            self.current_i_entry = pos
            yield chunk
            pos += n
