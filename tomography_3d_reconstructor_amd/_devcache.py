"""Device-side cache of volumes that crossed the NumPy boundary.

The reference's orchestrator hands the arrays returned by VoxelProcessor straight back into
smooth_voxel_data / extract_manifold_surface (tomography_3d_reconstruction.py:108-134).  To avoid a
host->device round trip per call, the bit-packed device copy is remembered against the returned
ndarray: object identity + data pointer + shape + a sampled checksum (so an in-place edit by the caller
is noticed and the volume is simply uploaded again).
"""
import weakref
from collections import OrderedDict

import numpy as np

_MAX = 4
_cache = OrderedDict()


def _fingerprint(arr):
    a = np.ascontiguousarray(arr) if not arr.flags.c_contiguous else arr
    flat = a.reshape(-1).view(np.uint8)
    n = flat.size
    step = max(1, n // 65536)
    sample = np.ascontiguousarray(flat[::step])      # contiguous: it is re-viewed as uint64 below
    return (arr.__array_interface__["data"][0], arr.shape, arr.strides, str(arr.dtype),
            int(sample.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(sample[: (sample.size // 8) * 8].view(np.uint64)))
            if sample.size >= 8 else 0)


def put(arr, vol):
    key = id(arr)
    try:
        ref = weakref.ref(arr, lambda _r, k=key: _cache.pop(k, None))
    except TypeError:
        return
    _cache[key] = (ref, _fingerprint(arr), vol)
    _cache.move_to_end(key)
    while len(_cache) > _MAX:
        _cache.popitem(last=False)


def get(arr):
    ent = _cache.get(id(arr))
    if ent is None:
        return None
    ref, fp, vol = ent
    if ref() is not arr or fp != _fingerprint(arr):
        _cache.pop(id(arr), None)
        return None
    _cache.move_to_end(id(arr))
    return vol


def clear():
    _cache.clear()
