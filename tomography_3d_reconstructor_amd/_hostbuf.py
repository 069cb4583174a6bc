"""Host arrays for the volume-sized downloads of the drop-in classes.

Host arrays in, host arrays out: create_voxel_data and smooth_voxel_data each hand back a 1 B/voxel ndarray
(voxel_processor.py:46, :84 -- np.stack / .copy() in the reference), 1 GiB at 1024^3.  Measured on the MI355X host
(tools/pintime.py): a FRESH gibibyte costs ~65 ms before a byte has moved -- the page faults of whoever touches it first, be
it page-locking (hipHostMalloc 68 ms, not parallel: 4 x 1 GiB on four threads 223 ms), np.ones (64 ms) or the copy itself --
while the transfer takes 19 ms into page-locked and 21 ms into plain pageable memory whose pages are already there (ROCm
pins a large pageable destination on the fly).  Rounds 1-2 returned page-locked arrays: fast once torch's pinned cache was
warm, but the FIRST create -> smooth -> extract of a process -- the only one the reference's orchestrator ever makes
(tomography_3d_reconstruction.py:271-323) -- took 230-260 ms against 68 ms warm.  Now the arrays are ordinary NumPy
arrays whose pages are brought in by many threads at once (tomo_host_touch: the faults of different pages run in parallel),
and reserve() starts that for the arrays a call chain WILL hand back while the upload of the mask stack is still under way.
"""
import os
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib

_pool = None
_lock = threading.Lock()
_reserved = {}       # nbytes -> list of (time, future of a paged-in uint8 array)
STALE_S = 20.0       # a reservation nobody came for is dropped after this long (its memory goes back to the system)


def _threads():
    return max(1, min(16, os.cpu_count() or 1))


def _fresh(nbytes):
    a = np.empty(int(nbytes), dtype=np.uint8)
    if nbytes >= (1 << 22):
        _lib.check(_lib.lib().tomo_host_touch(a.ctypes.data, a.nbytes, _threads()), "tomo_host_touch")
    return a


def _purge(now):
    for n in list(_reserved):
        _reserved[n] = [(t, f) for t, f in _reserved[n] if now - t < STALE_S]
        if not _reserved[n]:
            del _reserved[n]


_timer = [None]


def _purge_later():
    """Stale reservations are dropped by a timer, not by whoever calls next: a process that reserved and never came back (a
    create_voxel_data without the smooth_voxel_data that usually follows) must not keep volume-sized arrays resident."""
    with _lock:
        _timer[0] = None
        _purge(time.monotonic())
        again = bool(_reserved)
    if again:
        _arm_timer()


def _arm_timer():
    with _lock:
        if _timer[0] is None and _reserved:
            t = _timer[0] = threading.Timer(STALE_S + 0.5, _purge_later)
            t.daemon = True
            t.start()


MAX_RESERVE = int(os.environ.get("TOMO_HOSTBUF_MAX", str(16 << 30)))   # bytes of reservations in flight at most (beyond: take() allocates)


def reserve(nbytes, count=1):
    """Start paging `count` arrays of `nbytes` in, in the background, for take() calls that will follow shortly."""
    global _pool
    now = time.monotonic()
    with _lock:
        _purge(now)
        held = sum(n * len(v) for n, v in _reserved.items())
        count = min(count, len(_reserved.get(int(nbytes), [])) + max(0, (MAX_RESERVE - held) // max(int(nbytes), 1)))
        if _pool is None:
            _pool = ThreadPoolExecutor(2, thread_name_prefix="tomo-hostbuf")
        have = len(_reserved.get(int(nbytes), []))
        for _ in range(max(0, count - have)):
            _reserved.setdefault(int(nbytes), []).append((now, _pool.submit(_fresh, nbytes)))
    _arm_timer()


def take(shape, dtype):
    """A C-contiguous ndarray of this shape and dtype with its pages present: a reserved one (waits for it) or a fresh one."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    fut = None
    with _lock:
        _purge(time.monotonic())
        lst = _reserved.get(n)
        if lst:
            fut = lst.pop(0)[1]
            if not lst:
                del _reserved[n]
    a = None
    if fut is not None:
        try:
            a = fut.result()
        except Exception:           # noqa: BLE001 -- the plain allocation below reports whatever is wrong
            a = None
    if a is None:
        a = _fresh(n)
    return a.view(dtype).reshape(shape)
