"""Drop-in for the reference's voxel_processor.py: same class, same methods, same argument meaning
(/root/reference/voxel_processor.py:27-164), computed by the HIP kernels of libtomo_hip.so.

Host arrays in, host arrays out (the reference's consumers need real np.ndarray); the bit-packed
device copy of every returned volume is cached (see _devcache: returned volumes are write-protected, a
writeable array is verified byte for byte before its cached copy is used) so the orchestrator's next call does
not upload it again.  There is no CPU fallback: without a GPU / the built library these methods raise.
"""
import os
import sys
import threading

import numpy as np
import torch

from . import _devcache, _hostbuf, _memo, pipeline


def _device():
    if not torch.cuda.is_available():
        raise pipeline._lib.TomoUnavailable("no MI355X visible: the HIP path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def to_device_volume(voxel_data):
    """ndarray (nz,ny,nx) -> BitVolume, via the cache when the array came from this package."""
    vol = _devcache.get(voxel_data) if isinstance(voxel_data, np.ndarray) else None
    if vol is not None:
        return vol
    return upload_volume(voxel_data)


def with_device_volume(voxel_data, fn):
    """fn(vol) with the device copy of `voxel_data`.  A remembered copy of a WRITEABLE array (the default kind of result) is
    used speculatively: its byte-for-byte verification (a checksum of the whole array, 5-6 ms per GiB) runs on a helper thread
    NEXT TO fn -- the device work and the download of fn's result -- and fn runs again on a fresh upload if the array turns out
    to have been edited.  fn must have no effect but its return value."""
    if isinstance(voxel_data, np.ndarray):
        vol, check = _devcache.get_deferred(voxel_data)
        if vol is not None:
            res = fn(vol)
            if check is None or check():
                return res
    return fn(upload_volume(voxel_data))


def upload_volume(voxel_data):
    """ndarray (nz,ny,nx) -> a FRESH BitVolume (never the cached one: the caller may overwrite it)."""
    a = np.ascontiguousarray(voxel_data)
    if a.ndim != 3:
        raise ValueError("voxel data must be 3-D")
    if a.dtype != np.bool_:
        a = a != 0
    t = torch.from_numpy(a.view(np.uint8)).to(_device(), non_blocking=True)
    return pipeline.pack(t)


_NP_OF = {torch.bool: np.bool_, torch.uint8: np.uint8, torch.float32: np.float32, torch.int64: np.int64, torch.int32: np.int32}


BIG = 1 << 28        # bytes from which a transfer goes straight from / into ordinary (pageable) memory: ROCm page-locks a large
#                      pageable buffer on the fly (1 GiB: 21 ms against 19 ms page-locked), smaller ones pass through its staging
#                      buffers at a fraction of the bus rate -- those use torch's page-locked cache (a few ms to lock, once)


def to_host_array(t):
    """Device tensor -> fresh host ndarray.  Volume-sized results are ordinary NumPy arrays whose pages many threads have
    brought in (_hostbuf: page-locking a fresh gibibyte costs ~65 ms, the first run of a process would pay it three times);
    mesh-sized ones are NumPy views of page-locked torch tensors."""
    if t.numel() * t.element_size() >= BIG:
        out = _hostbuf.take(tuple(t.shape), _NP_OF[t.dtype])
        torch.from_numpy(out).copy_(t)          # ordered after the current stream's work; returns when the bytes are there
        return out
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return host.numpy()


PIECE = 128 << 20   # bytes per piece of a download that is checksummed on the way (a whole number of checksum chunks)
_DIGEST_POOL = []


def _download_with_digest(t):
    """to_host_array for a volume-sized tensor whose checksum is wanted as well (writeable results: _devcache verifies them
    at the next lookup): the download runs in pieces of PIECE bytes and piece k is digested by a helper thread (native, threaded;
    the foreign call releases the interpreter lock) while piece k + 1 is on the bus -- the 5-6 ms a checksum of 1 GiB costs
    disappear behind the 20 ms of its download.  -> (host ndarray, digest as _devcache.checksum gives it)."""
    import ctypes
    from concurrent.futures import ThreadPoolExecutor
    L = pipeline._lib.lib()
    out = _hostbuf.take(tuple(t.shape), _NP_OF[t.dtype])
    flat_h = out.reshape(-1).view(np.uint8)
    flat_d = t.reshape(-1).view(torch.uint8)
    n = flat_h.shape[0]
    chunk = int(L.tomo_host_checksum_chunk_bytes())
    piece = max(chunk, PIECE // chunk * chunk)
    nchunks = -(-n // chunk)
    dig = np.zeros(2 * nchunks + 2, dtype=np.uint64)
    nt = max(1, min(16, os.cpu_count() or 1))
    if not _DIGEST_POOL:
        _DIGEST_POOL.append(ThreadPoolExecutor(1, thread_name_prefix="tomo-digest"))

    def digest(a, b):
        pipeline._lib.check(L.tomo_host_checksum_part(flat_h[a:b].ctypes.data, b - a, a // chunk, nt, 0,
                                                      dig[2 * (a // chunk):].ctypes.data), "tomo_host_checksum_part")
    futs = []
    for a in range(0, n, piece):
        b = min(n, a + piece)
        torch.from_numpy(flat_h[a:b]).copy_(flat_d[a:b])      # ordered after the current stream's work; returns when the bytes are there
        futs.append(_DIGEST_POOL[0].submit(digest, a, b))
    for f in futs:
        f.result()
    res = (ctypes.c_uint64 * 2)()
    pipeline._lib.check(L.tomo_host_checksum_fold(dig.ctypes.data, nchunks, n, res), "tomo_host_checksum_fold")
    return out, (int(res[0]), int(res[1]))


def to_host_volume(vol):
    """BitVolume -> fresh host bool ndarray, remembered against its device copy (writeable and checksummed by default;
    write-protected under TOMO_READONLY_RESULTS)."""
    t = pipeline.unpack(vol)
    if _devcache.WRITEABLE_RESULTS and t.numel() * t.element_size() >= BIG:
        out, digest = _download_with_digest(t)
        _devcache.put(out, vol, digest=digest)
        return out
    out = to_host_array(t)
    _devcache.put(out, vol)
    return out


_STAGE = {}          # the staging array of the last upload, kept for the next one of the same size: a fresh gibibyte costs its
#                      page faults on the way in AND ~25 ms of munmap on the way out (measured: tools/h2hprof.py)
_STAGE_LOCK = threading.Lock()
_STAGE_TIMER = [None]
STAGE_IDLE_S = 20.0  # ... but not for ever: an idle staging array goes back to the system after this long (a timer, not the
#                      next call: a process that made its one create_voxel_data must not sit on a volume-sized array)
STAGE_KEEP_MAX = int(os.environ.get("TOMO_STAGE_KEEP_MAX", str(8 << 30)))   # larger staging arrays are never kept between calls


def _release_staging():
    with _STAGE_LOCK:
        _STAGE.clear()
        _STAGE_TIMER[0] = None


def _staging(shape):
    key = (threading.get_ident(), tuple(shape))
    with _STAGE_LOCK:
        a = _STAGE.get(key)
        if a is None:
            _STAGE.clear()                      # one staging array per process: another shape replaces it
            a = np.empty(shape, dtype=np.bool_)
            if a.nbytes <= STAGE_KEEP_MAX:
                _STAGE[key] = a
        if _STAGE_TIMER[0] is not None:
            _STAGE_TIMER[0].cancel()
        if _STAGE:
            t = _STAGE_TIMER[0] = threading.Timer(STAGE_IDLE_S, _release_staging)
            t.daemon = True
            t.start()
    return a


def _stage_masks(mask_images):
    """np.stack(mask_images) (voxel_processor.py:46) straight into a page-locked staging tensor, copied by a few host
    threads (NumPy releases the GIL while copying) and uploaded chunk by chunk behind them: (nz, ny, nx) uint8 0/1 on
    the device."""
    from concurrent.futures import ThreadPoolExecutor
    first = np.asarray(mask_images[0])
    if first.ndim != 2:
        raise ValueError("all input arrays must have the same shape")          # what np.stack reports for ragged input
    nz = len(mask_images)
    # np.stack into ONE staging array (ordinary memory: its pages are brought in by the copying workers themselves, in
    # parallel), cut into chunks of slices; every chunk is uploaded as soon as its host copy is done, so the PCIe transfer
    # runs under the host copies of the later chunks.  (Rounds 1-2 staged through page-locked memory: 68 ms of page-locking
    # in front of the first call of a process, for a transfer that is as fast from pageable memory on this platform.)
    stage = _staging((nz,) + first.shape)
    # uploads in pieces of >= 128 MiB (below that ROCm stages a pageable source through its own buffers at a fraction of the
    # bus rate; above it the source is page-locked on the fly: 1 GiB in 8 pieces 19 ms, in 32 pieces 90-120 ms)
    per_slice = int(first.size)             # bytes per mask (bool); create_voxel_data keeps zero-size stacks away from here
    if per_slice == 0:
        raise ValueError("zero-size masks have nothing to stage")
    up = max(1, min(nz, -(-(128 << 20) // per_slice)))                                           # slices per upload piece
    dev = torch.empty((nz,) + first.shape, dtype=torch.bool, device=_device())
    # Fast path (what an image loader produces: separate C-contiguous bool / uint8-of-0/1... arrays of one shape): the
    # stacking is a native multi-threaded gather (tomo_host_gather) -- piece k + 1 is gathered on a helper thread while
    # piece k is on the bus.  (Python worker threads copying slice by slice took 17-60 ms for the same gibibyte, at the
    # mercy of the interpreter lock; np.stack itself 85 ms.)
    L = pipeline._lib.lib()
    ptrs = np.empty(nz, dtype=np.uintp)
    plain = True
    for i, m in enumerate(mask_images):
        if not (isinstance(m, np.ndarray) and m.dtype == np.bool_ and m.shape == first.shape and m.flags.c_contiguous):
            plain = False
            break
        ptrs[i] = m.__array_interface__["data"][0]
    nthreads = max(1, min(16, os.cpu_count() or 1))
    if plain:
        def gather(lo, hi):
            pipeline._lib.check(L.tomo_host_gather(ptrs[lo:hi].ctypes.data, hi - lo, per_slice, stage[lo:hi].ctypes.data, nthreads),
                                "tomo_host_gather")
        with ThreadPoolExecutor(1) as ex:          # measured (tools/h2hprof.py): gather 1.2-1.5 ms, upload 2.4-2.9 ms per 128 MiB
            spans = [(lo, min(nz, lo + up)) for lo in range(0, nz, up)]
            futs = [ex.submit(gather, lo, hi) for lo, hi in spans]
            for (lo, hi), fut in zip(spans, futs):
                fut.result()
                dev[lo:hi].copy_(torch.from_numpy(stage[lo:hi]))
        return dev.view(torch.uint8)
    workers = max(1, min(8, os.cpu_count() or 1, nz))
    fine = max(1, -(-up // workers))                                                           # slices per worker task

    def fill(lo, hi):
        for i in range(lo, hi):
            m = np.asarray(mask_images[i])
            if m.shape != first.shape:
                raise ValueError("all input arrays must have the same shape")
            stage[i] = m if m.dtype == np.bool_ else (m != 0)
    with ThreadPoolExecutor(workers) as ex:
        pieces = []
        for lo in range(0, nz, up):
            hi = min(nz, lo + up)
            pieces.append((lo, hi, [ex.submit(fill, a, min(hi, a + fine)) for a in range(lo, hi, fine)]))
        for lo, hi, futs in pieces:
            for fut in futs:
                fut.result()
            dev[lo:hi].copy_(torch.from_numpy(stage[lo:hi]))
    return dev.view(torch.uint8)


def _common_base(mask_images):
    """If the masks are the consecutive slices (views) of ONE contiguous (nz, ny, nx) array -- what this package's
    ImageLoader hands out -- return that array: np.stack would only copy it."""
    first = mask_images[0]
    base = getattr(first, "base", None)
    if not (isinstance(first, np.ndarray) and isinstance(base, np.ndarray) and base.ndim == 3 and
            base.shape[0] == len(mask_images) and base.flags.c_contiguous and base.dtype == first.dtype):
        return None
    ptr0, step = base.__array_interface__["data"][0], base.strides[0]
    for i, m in enumerate(mask_images):
        if not (isinstance(m, np.ndarray) and m.base is base and m.shape == base.shape[1:] and
                m.strides == base.strides[1:] and m.__array_interface__["data"][0] == ptr0 + i * step):
            return None
    return base


class VoxelProcessor:
    """Handles voxel data creation and processing operations (reference: voxel_processor.py:27)."""

    def __init__(self):
        self.voxel_data = None
        self.side_0_count = 0
        self.side_1_count = 0
        self.side_2_count = 0

    def create_voxel_data(self, mask_images: list, close_ends: bool = True,
                          side_0_count: int = 0, side_1_count: int = 0, side_2_count: int = 0) -> np.ndarray:
        """voxel_processor.py:36-54."""
        if not mask_images:
            raise ValueError("Load masks first, hmm.")
        self.side_0_count = side_0_count
        self.side_1_count = side_1_count
        self.side_2_count = side_2_count
        if np.asarray(mask_images[0]).size == 0:
            # masks without a single pixel (shape (0, k) / (k, 0)): nothing to upload, close or count -- np.stack as the reference
            # does (voxel_processor.py:46; a ragged list raises its ValueError), closing the ends of an empty volume changes nothing
            self.voxel_data = np.stack(mask_images, axis=0)
            print(f"Voxels: {self.voxel_data.shape}, active: {int(np.sum(self.voxel_data)):,}")
            return self.voxel_data
        base = _common_base(mask_images)
        cached = _devcache.get(base) if base is not None else None
        if close_ends and torch.cuda.is_available():
            # the 1 B/voxel host arrays this call and the smooth_voxel_data that follows it hand back: their pages are brought in
            # NOW, on helper threads, next to the upload
            _hostbuf.reserve(len(mask_images) * np.asarray(mask_images[0]).size, count=2)
        if cached is not None and not base.flags.writeable and any(m.flags.writeable for m in mask_images):
            # writeable views of a protected stack: its content may have changed behind the flag -- the remembered device
            # copy is dropped for good (a later lookup must not find it either) and the stack is uploaded again
            _devcache.invalidate(base)
            cached = None
        if close_ends:
            if cached is not None:                           # uploaded (and thresholded) by ImageLoader already: closed into a COPY
                vol = pipeline.close_ends(cached)
            elif base is not None:
                vol = pipeline.close_ends(upload_volume(base), inplace=True)           # a fresh upload, never a cached volume: ours to overwrite
            else:
                vol = pipeline.pack_closed(_stage_masks(mask_images))
            active = int(pipeline.popcount_async(vol).item())
            self.voxel_data = to_host_volume(vol)
        else:
            stacked = np.stack(mask_images, axis=0)          # the reference returns a new array here
            vol = cached if cached is not None else upload_volume(stacked)
            active = int(pipeline.popcount_async(vol).item())
            self.voxel_data = stacked
            _devcache.put(stacked, vol)
        print(f"Voxels: {self.voxel_data.shape}, active: {active:,}")
        return self.voxel_data

    def smooth_voxel_data(self, voxel_data: np.ndarray, iterations: int = 3, create_manifold: bool = True) -> np.ndarray:
        """voxel_processor.py:79-97 (binary_opening + `iterations` x binary_closing, 3-D cross).  Like the reference
        (:93-95) a failure of the full sequence falls back to the closings alone; only a missing GPU / library raises
        on the first attempt (there is no CPU path to fall back to)."""
        def work(vol):
            try:
                # the orchestrator repeats this very call several times: the device result is remembered (see _memo), the
                # host array is a fresh download every time
                key = (int(iterations), bool(create_manifold))
                sm = _memo.smoothed.get(vol, key)
                if sm is None:
                    sm = pipeline.smooth(vol, iterations, create_manifold)
                    _memo.smoothed.put(vol, key, sm)
                return to_host_volume(sm)
            except pipeline._lib.TomoUnavailable:
                raise
            except Exception as e:                                       # noqa: BLE001 -- the reference catches Exception here
                print(f"tomography_3d_reconstructor_amd: smoothing failed ({e}); closings only", file=sys.stderr)
                return to_host_volume(pipeline.smooth(vol, iterations, False))
        return with_device_volume(voxel_data, work)

    def generate_point_cloud(self, voxel_data: np.ndarray, mm_per_pixel_x: float, mm_per_pixel_y: float,
                             slice_depths: np.ndarray, subsample_factor: int = 1) -> np.ndarray:
        """voxel_processor.py:99-127 (fallback path of the orchestrator; host NumPy, not on the hot path)."""
        z, y, x = np.where(voxel_data)
        if subsample_factor > 1:
            idx = np.arange(0, len(z), subsample_factor)
            z, y, x = z[idx], y[idx], x[idx]
        d = np.asarray(slice_depths, dtype=np.float64)
        cum = np.cumsum(np.concatenate([[0], d]))
        if len(d):
            zc = np.minimum(z, len(d) - 1)
            z_mm = np.where(z < len(d), cum[zc] + d[zc] / 2, cum[-1])
        else:
            z_mm = np.full(len(z), cum[-1])
        return np.column_stack([z_mm, y * mm_per_pixel_y, x * mm_per_pixel_x])

    def calculate_slice_depths(self, total_depth_mm: float) -> np.ndarray:
        """voxel_processor.py:129-164."""
        s0, s1, s2 = self.side_0_count, self.side_1_count, self.side_2_count
        total = s0 + s1 + s2
        if s1 == 0 or total == 0:
            if total == 0:
                return np.array([])
            return np.full(total, total_depth_mm / total)
        d1 = total_depth_mm / s1
        d02 = 2 * d1
        d0 = d02 / s0 if s0 > 0 else 0
        d2 = d02 / s2 if s2 > 0 else 0
        depths = [d0] * s0 + [d1] * s1 + [d2] * s2
        print(f"Slice depth sequence: Side_0[0-{s0-1}], Side_1[{s0}-{s0+s1-1}], Side_2[{s0+s1}-{len(depths)-1}]")
        return np.array(depths)
