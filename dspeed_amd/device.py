"""Device memory and streams on top of the C ABI -- the counterpart of the reference's pre-allocated
``ProcChainVar`` buffers (processing_chain.py:259-269), but resident in HBM.

``DeviceArray`` is deliberately tiny: shape, dtype, pointer, explicit copies.  Anything that exposes
``data_ptr()`` (e.g. a torch tensor on the same device) can be wrapped with ``DeviceArray.from_ptr``.
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np

from . import _lib

_DTYPES = {np.dtype(np.float32): _lib.F32, np.dtype(np.float64): _lib.F64, np.dtype(np.int16): _lib.I16,
           np.dtype(np.uint16): _lib.U16, np.dtype(np.int32): _lib.I32, np.dtype(np.uint32): _lib.U32,
           np.dtype(np.bool_): _lib.BOOL, np.dtype(np.int64): _lib.I64, np.dtype(np.uint64): _lib.U64}


def dtype_code(dt) -> int:
    try:
        return _DTYPES[np.dtype(dt)]
    except KeyError:
        raise TypeError(f"dtype {dt} is not supported by the device path") from None


def device_count() -> int:
    n = C.c_int(0)
    _lib.lib().dsp_device_count(C.byref(n))
    return n.value


def set_device(i: int) -> None:
    _lib.check(_lib.lib().dsp_set_device(i), what="set_device")


def device_info(i: int = 0) -> dict:
    name = C.create_string_buffer(256)
    cus, hbm, lds = C.c_int(), C.c_int64(), C.c_int()
    _lib.check(_lib.lib().dsp_device_info(i, name, 256, C.byref(cus), C.byref(hbm), C.byref(lds)), what="device_info")
    return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": hbm.value, "lds_bytes_per_cu": lds.value}


def sync() -> None:
    _lib.check(_lib.lib().dsp_sync(), what="sync")


class Stream:
    """A HIP stream (``hipStreamNonBlocking``); ``Stream.ptr`` is what the C ABI takes."""

    def __init__(self):
        p = C.c_void_p()
        _lib.check(_lib.lib().dsp_stream_create(C.byref(p)), what="stream_create")
        self.ptr = p

    def sync(self):
        _lib.check(_lib.lib().dsp_stream_sync(self.ptr), what="stream_sync")

    def wait_event(self, event: "Event"):
        """work queued on this stream from now on starts after ``event``"""
        _lib.check(_lib.lib().dsp_stream_wait_event(self.ptr, event.ptr), what="stream_wait_event")

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().dsp_stream_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


class Event:
    def __init__(self):
        p = C.c_void_p()
        _lib.check(_lib.lib().dsp_event_create(C.byref(p)), what="event_create")
        self.ptr = p

    def record(self, stream: Stream | None = None):
        _lib.check(_lib.lib().dsp_event_record(self.ptr, stream.ptr if stream else None), what="event_record")

    def sync(self):
        _lib.check(_lib.lib().dsp_event_sync(self.ptr), what="event_sync")

    def elapsed_ms(self, later: "Event") -> float:
        ms = C.c_float()
        _lib.check(_lib.lib().dsp_event_elapsed_ms(self.ptr, later.ptr, C.byref(ms)), what="event_elapsed")
        return ms.value

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().dsp_event_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


class PinnedArray:
    """Page-locked host staging buffer viewed as a NumPy array (for H2D/D2H that overlaps kernels)."""

    def __init__(self, shape, dtype):
        self.shape = tuple(int(s) for s in np.atleast_1d(shape))
        self.dtype = np.dtype(dtype)
        nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        _lib.check(_lib.lib().dsp_host_alloc(C.byref(p), nbytes), what="host_alloc")
        self.ptr = p
        buf = (C.c_char * max(nbytes, 1)).from_address(p.value)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(self.shape))).reshape(self.shape)

    def __del__(self):
        try:
            if self.ptr:
                self.array = None
                _lib.lib().dsp_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


class _Bounce:
    """One page-locked staging buffer for the synchronous copies (``DeviceArray.copy_from`` / ``to_numpy`` without a stream: the
    processor-by-processor path).  Copying between device memory and arbitrary NumPy memory makes the runtime page-lock those pages
    on the fly, per call; NumPy hands megabyte arrays back to the OS when they die and gets the same addresses again, and the
    runtime's records of such ranges have been seen to end a long test session in abort() inside hipMemcpy.  Through this buffer
    the device only ever exchanges data with memory the runtime allocated itself; the extra host copy is noise next to PCIe."""

    CHUNK = 32 << 20
    _local = threading.local()  # one buffer per thread: worker threads of a multi-device build_dsp copy at the same time

    @classmethod
    def get(cls) -> np.ndarray:
        buf = getattr(cls._local, "buf", None)
        if buf is None:
            buf = cls._local.buf = PinnedArray((cls.CHUNK,), np.uint8)
        return buf.array

    @classmethod
    def h2d(cls, dev_ptr: int, a: np.ndarray) -> None:
        src = a.reshape(-1).view(np.uint8)
        buf = cls.get()
        for o in range(0, src.nbytes, cls.CHUNK):
            n = min(cls.CHUNK, src.nbytes - o)
            buf[:n] = src[o:o + n]
            _lib.check(_lib.lib().dsp_h2d(dev_ptr + o, buf.ctypes.data, n), what="h2d")

    @classmethod
    def d2h(cls, out: np.ndarray, dev_ptr: int) -> None:
        dst = out.reshape(-1).view(np.uint8)
        buf = cls.get()
        for o in range(0, dst.nbytes, cls.CHUNK):
            n = min(cls.CHUNK, dst.nbytes - o)
            _lib.check(_lib.lib().dsp_d2h(buf.ctypes.data, dev_ptr + o, n), what="d2h")
            dst[o:o + n] = buf[:n]


_NEVER_FREED = []
_PINNED = {}  # (address, bytes) -> number of HostPin objects holding that registration (hipHostRegister itself does not count)


class HostPin:
    """Page-locks a caller-owned C-contiguous NumPy array in place for as long as this object lives, so that ``h2d_async`` /
    ``d2h_async`` on it overlap kernels without a staging copy.  The reference's ``build_dsp`` fills the same input buffer for
    every file chunk (build_dsp.py:399-432), so a linked buffer is pinned once and reused.  Several holders of the same buffer
    (two chains linked to one table) share one registration: the runtime accepts a second ``hipHostRegister`` of a range but a
    single unregister then drops it for everybody, so the count is kept here."""

    def __init__(self, array: np.ndarray):
        if not (isinstance(array, np.ndarray) and array.flags.c_contiguous and array.nbytes > 0):
            raise ValueError("HostPin needs a non-empty C-contiguous NumPy array")
        self.array = array
        self.ptr = array.ctypes.data
        self.nbytes = array.nbytes
        self._key = (self.ptr, self.nbytes)
        if _PINNED.get(self._key, 0) == 0:
            # The runtime locks whole pages and keeps one record per range: a second registration that shares pages with a live one
            # (a row range of a pinned array, the same start with another length) must not reach it -- its unregister would drop
            # or orphan the other's record, and an orphaned record outlives the memory it describes.  Such a range stays unpinned.
            lo, hi = self.ptr & ~0xFFF, (self.ptr + self.nbytes + 0xFFF) & ~0xFFF
            for (p, n), held in _PINNED.items():
                if held > 0 and (p & ~0xFFF) < hi and lo < ((p + n + 0xFFF) & ~0xFFF):
                    raise RuntimeError("HostPin: the range shares pages with one that is already page-locked")
            _lib.check(_lib.lib().dsp_host_register(self.ptr, self.nbytes), what="host_register")
        _PINNED[self._key] = _PINNED.get(self._key, 0) + 1
        self._held = True

    def close(self):
        if getattr(self, "_held", False):
            self._held = False
            left = _PINNED.get(self._key, 1) - 1
            if left <= 0:
                _PINNED.pop(self._key, None)
                if _lib.lib().dsp_host_unregister(self.ptr) != 0:
                    _NEVER_FREED.append(self.array)  # the runtime kept its record of the range: the addresses must not come back
            else:
                _PINNED[self._key] = left

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceArray:
    """C-contiguous array in device memory."""

    def __init__(self, shape, dtype, ptr=None, owner=True):
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self._owner = owner and ptr is None
        if ptr is None:
            p = C.c_void_p()
            _lib.check(_lib.lib().dsp_malloc(C.byref(p), self.nbytes), what="malloc")
            ptr = p.value
        self.ptr = int(ptr) if ptr else 0

    # -- construction
    @classmethod
    def from_numpy(cls, a) -> "DeviceArray":
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        d.copy_from(a)
        return d

    @classmethod
    def from_ptr(cls, ptr: int, shape, dtype) -> "DeviceArray":
        return cls(shape, dtype, ptr=ptr, owner=False)

    @classmethod
    def zeros(cls, shape, dtype) -> "DeviceArray":
        d = cls(shape, dtype)
        _lib.check(_lib.lib().dsp_memset(d.ptr, 0, d.nbytes, None), what="memset")
        return d

    # -- copies
    def copy_from(self, a, stream: Stream | None = None):
        """Host -> device.  With a stream the copy is asynchronous and reads ``a``'s memory until the stream has passed it: ``a`` must
        then already be a C-contiguous array of this array's dtype that the caller keeps alive (a converted temporary would be freed
        under the DMA)."""
        if stream is not None:
            if not (isinstance(a, np.ndarray) and a.flags.c_contiguous and a.dtype == self.dtype):
                raise ValueError("asynchronous copy_from needs a C-contiguous ndarray of the device array's dtype (no temporary copies)")
            assert a.nbytes == self.nbytes, f"size mismatch {a.shape} vs {self.shape}"
            self._h2d_src = a  # (kept until the next copy: the stream may still be reading it when the caller drops its reference)
            _lib.check(_lib.lib().dsp_h2d_async(self.ptr, a.ctypes.data, self.nbytes, stream.ptr), what="h2d_async")
            return
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.nbytes == self.nbytes, f"size mismatch {a.shape} vs {self.shape}"
        if self.nbytes:
            _Bounce.h2d(self.ptr, a)

    def to_numpy(self, out=None, stream: Stream | None = None):
        if out is None:
            out = np.empty(self.shape, dtype=self.dtype)
        assert out.flags.c_contiguous and out.nbytes == self.nbytes
        if stream is None:
            if self.nbytes:
                _Bounce.d2h(out, self.ptr)
        else:
            _lib.check(_lib.lib().dsp_d2h_async(out.ctypes.data, self.ptr, self.nbytes, stream.ptr), what="d2h_async")
        return out

    def view_rows(self, start: int, stop: int) -> "DeviceArray":
        """Rows [start, stop) of a 2-D (or 1-D) array, sharing memory."""
        row_bytes = self.nbytes // self.shape[0] if self.shape[0] else 0
        v = DeviceArray((stop - start, *self.shape[1:]), self.dtype, ptr=self.ptr + start * row_bytes, owner=False)
        v._base = self  # (the view keeps the allocation alive)
        return v

    def __len__(self):
        return self.shape[0]

    @property
    def ndim(self):
        return len(self.shape)

    def __repr__(self):
        return f"DeviceArray(shape={self.shape}, dtype={self.dtype}, ptr=0x{self.ptr:x})"

    def free(self):
        if self._owner and self.ptr:
            _lib.lib().dsp_free(self.ptr)
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
