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


def test_protected_results_cannot_be_written_through_their_base(monkeypatch):
    """_devcache.put(protect=True) write-protects the array AND every ndarray it is a view of: the volume-sized results are typed
    views of a byte buffer (_hostbuf.take), and a cached device copy is trusted for as long as the array is read-only."""
    from tomography_3d_reconstructor_amd import _devcache
    monkeypatch.setattr(_devcache, "WRITEABLE_RESULTS", False)         # the opt-in fast path (TOMO_READONLY_RESULTS=1)
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


@pytest.mark.parametrize("shape", [(0, 5), (5, 0), (0, 0)])
def test_zero_size_masks_never_reach_the_native_gather(shape, capsys):
    """masks without a pixel.  The reference (voxel_processor.py:46-54) returns np.stack of them and prints
    `active: 0` (probed); the staging path must not hand tomo_host_gather a byte count it made up (max(size, 1) wrote
    nz bytes into a 0-byte staging array).  Needs no GPU: nothing is uploaded.  Runs under ASan in tools/asan_cpu_tier.sh."""
    from tomography_3d_reconstructor_amd import voxel_processor as VP
    masks = [np.zeros(shape, bool) for _ in range(4)]
    vp = VP.VoxelProcessor()
    out = vp.create_voxel_data(masks, True, 0, 4, 0)
    assert out.shape == (4,) + shape and out.dtype == np.bool_ and out.flags.writeable
    assert "Voxels: %s, active: 0" % ((4,) + shape,) in capsys.readouterr().out
    with pytest.raises(ValueError):
        vp.create_voxel_data([np.zeros(shape, bool), np.zeros((1, 1), bool)], True, 0, 2, 0)      # ragged: np.stack's error
    with pytest.raises(ValueError):
        VP._stage_masks(masks)                                                                    # the staging path refuses
    L = _lib.lib()
    dst = np.empty(0, np.uint8)
    ptrs = np.array([m.__array_interface__["data"][0] or 1 for m in masks], dtype=np.uintp)
    assert L.tomo_host_gather(ptrs.ctypes.data, 4, 0, dst.ctypes.data, 4) == 0                    # bytes_each = 0: returns at once


def test_idle_reservations_and_staging_are_released_by_a_timer(monkeypatch):
    """a volume-sized reservation / staging array nobody comes back for goes away by itself (a timer), not only when
    the next call happens to purge it."""
    import time
    from tomography_3d_reconstructor_amd import voxel_processor as VP
    monkeypatch.setattr(_hostbuf, "STALE_S", 0.2)
    with _hostbuf._lock:                                               # (a timer an earlier test armed with the real period)
        if _hostbuf._timer[0] is not None:
            _hostbuf._timer[0].cancel()
            _hostbuf._timer[0] = None
    n = (1 << 22) + 8192
    _hostbuf.reserve(n, count=2)
    assert len(_hostbuf._reserved.get(n, [])) == 2
    t_end = time.monotonic() + 10
    while _hostbuf._reserved and time.monotonic() < t_end:
        time.sleep(0.05)
    assert not _hostbuf._reserved
    monkeypatch.setattr(_hostbuf, "MAX_RESERVE", n)                    # the cap: one array of this size at most
    _hostbuf.reserve(n, count=3)
    assert len(_hostbuf._reserved.get(n, [])) == 1
    assert _hostbuf.take((n,), np.uint8).shape == (n,) and not _hostbuf._reserved
    monkeypatch.setattr(VP, "STAGE_IDLE_S", 0.2)
    a = VP._staging((3, 4, 5))
    assert VP._staging((3, 4, 5)) is a                                 # kept for the next call of the same shape ...
    t_end = time.monotonic() + 10
    while VP._STAGE and time.monotonic() < t_end:
        time.sleep(0.05)
    assert not VP._STAGE                                               # ... and released when none comes
    monkeypatch.setattr(VP, "STAGE_KEEP_MAX", 10)
    assert VP._staging((3, 4, 5)) is not VP._staging((3, 4, 5)) and not VP._STAGE      # too large to keep at all
