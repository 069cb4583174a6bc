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
