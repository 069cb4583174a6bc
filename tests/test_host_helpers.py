"""CPU tier: the host-side helpers of the drop-in classes' transfers (tomo_host_gather = np.stack on several threads,
tomo_host_touch = page a fresh buffer in) and the host-array reservations built on them (_hostbuf.py)."""
import numpy as np
import pytest

from tomography_3d_reconstructor_amd import _hostbuf, _lib


@pytest.mark.parametrize("n,shape,threads", [(11, (37, 53), 4), (1, (5, 7), 8), (40, (64, 64), 1), (3, (1200, 1000), 16)])
def test_host_gather_is_np_stack(n, shape, threads):
    L = _lib.lib()
    rng = np.random.default_rng(n)
    ms = [rng.random(shape) < 0.5 for _ in range(n)]
    ptrs = np.array([m.__array_interface__["data"][0] for m in ms], dtype=np.uintp)
    out = np.full((n,) + shape, True, np.bool_)
    assert L.tomo_host_gather(ptrs.ctypes.data, n, ms[0].size, out.ctypes.data, threads) == 0
    assert np.array_equal(out, np.stack(ms))
    assert L.tomo_host_gather(None, 0, 10, None, 4) == 0                       # nothing to do
    assert L.tomo_host_gather(None, 2, 10, out.ctypes.data, 4) == -1           # TOMO_E_ARG
    ptrs[0] = 0
    assert L.tomo_host_gather(ptrs.ctypes.data, n, ms[0].size, out.ctypes.data, threads) == -1


def test_host_touch_and_reservations():
    L = _lib.lib()
    a = np.empty(5 * 4096 + 17, np.uint8)
    assert L.tomo_host_touch(a.ctypes.data, a.nbytes, 3) == 0 and L.tomo_host_touch(None, 0, 3) == 0
    assert L.tomo_host_touch(None, 8, 3) == -1
    n = (1 << 22) + 4096
    _hostbuf.reserve(n, count=2)
    x = _hostbuf.take((n,), np.bool_)
    y = _hostbuf.take((n // 8,), np.int64)
    z = _hostbuf.take((3, 5), np.float32)                                      # no reservation of that size: a fresh array
    assert x.shape == (n,) and x.dtype == np.bool_ and y.dtype == np.int64 and z.shape == (3, 5)
    assert all(v.flags.c_contiguous and v.flags.writeable for v in (x, y, z))
    assert not np.shares_memory(x, y)
    x[:] = True
    y[:] = 7
    assert x.all() and (y == 7).all()


def test_protected_results_cannot_be_written_through_their_base():
    """_devcache.put(protect=True) write-protects the array AND every ndarray it is a view of: the volume-sized results are typed
    views of a byte buffer (_hostbuf.take), and a cached device copy is trusted for as long as the array is read-only."""
    from tomography_3d_reconstructor_amd import _devcache
    if _devcache.WRITEABLE_RESULTS:
        pytest.skip("TOMO_WRITEABLE_RESULTS: results stay writeable and are verified by checksum instead")
    a = _hostbuf.take((4, 5, 6), np.bool_)
    a[:] = True
    assert isinstance(a.base, np.ndarray) and a.base.flags.writeable
    _devcache.put(a, object())
    assert not a.flags.writeable and not a.base.flags.writeable
    with pytest.raises(ValueError):
        a.base[0] = 0
    assert _devcache.get(a) is not None                       # read-only all the way down: the remembered copy is exact
    a.base.setflags(write=True)                               # the owner of the bytes CAN be re-enabled (NumPy allows it) ...
    a.base[0] = 0
    assert _devcache.get(a) is None                           # ... and the remembered copy is then no longer trusted
