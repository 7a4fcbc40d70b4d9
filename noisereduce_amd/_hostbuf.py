"""Host-side result buffers for numpy-in / numpy-out calls.

The reference returns a freshly allocated ndarray (base.py:217-226).  A device-to-host copy into
fresh pageable memory costs ~3x the PCIe time on ROCm (the driver has to lock and map the new pages
first: 6.4 ms instead of 2.0 ms for 115 MB, measured with tools/try_hugepage.py; huge pages do not
help).  Results are therefore handed out as ndarrays backed by page-locked buffers from a small
pool; a buffer returns to the pool when the last array (or view) that refers to it is garbage
collected.  The pool is bounded: when too many results are alive at once, or the result is small,
plain ``np.empty`` memory is used instead -- callers never see the difference.
"""
import os
import threading
import weakref

import numpy as np
import torch

_MIN_BYTES = 1 << 20          # below this a pinned buffer buys nothing
_MAX_LIVE = 4                 # page-locked result buffers alive (handed out + idle) per size class
_MAX_TOTAL = 2 << 30          # bytes of page-locked memory the pool may hold in total
_ENABLED = os.environ.get("NOISEREDUCE_AMD_PINNED_RESULTS", "1") != "0"

_lock = threading.Lock()
_idle = {}                    # size class -> [uint8 pinned tensors]
_live = {}                    # size class -> number of buffers in existence
_total = 0


class _Owner:
    """Keeps a pooled buffer alive for as long as any ndarray (or view of it) refers to it."""

    def __init__(self, buf, shape, dtype):
        self.buf = buf
        self.__array_interface__ = dict(shape=tuple(shape), typestr=np.dtype(dtype).str,
                                        data=(buf.data_ptr(), False), version=3)


def _release(cls, buf):
    with _lock:
        _idle.setdefault(cls, []).append(buf)


def _take(nbytes):
    """A page-locked uint8 tensor of at least nbytes from the pool, or None (pool exhausted)."""
    global _total
    cls = (nbytes + _MIN_BYTES - 1) // _MIN_BYTES * _MIN_BYTES
    with _lock:
        free = _idle.get(cls)
        if free:
            return cls, free.pop()
        if _live.get(cls, 0) >= _MAX_LIVE or _total + cls > _MAX_TOTAL:
            return cls, None
        _live[cls] = _live.get(cls, 0) + 1
        _total += cls
    try:
        return cls, torch.empty(cls, dtype=torch.uint8, pin_memory=True)
    except RuntimeError:
        with _lock:
            _live[cls] -= 1
            _total -= cls
        return cls, None


def result_array(shape, dtype):
    """(ndarray, tensor): a writable ndarray of `shape`/`dtype` for a result and a torch tensor that
    shares its memory (the target of the device-to-host copy)."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) if len(shape) else 1
    nbytes = n * dtype.itemsize
    if _ENABLED and nbytes >= _MIN_BYTES and torch.cuda.is_available():
        cls, buf = _take(nbytes)
        if buf is not None:
            owner = _Owner(buf, shape, dtype)
            weakref.finalize(owner, _release, cls, buf)
            arr = np.asarray(owner)
            return arr, torch.from_numpy(arr)
    arr = np.empty(tuple(shape), dtype=dtype)
    return arr, torch.from_numpy(arr)


def pool_stats():
    with _lock:
        return dict(live=dict(_live), idle={k: len(v) for k, v in _idle.items()}, total_bytes=_total)
