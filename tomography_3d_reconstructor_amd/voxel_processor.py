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


def to_host_volume(vol):
    """BitVolume -> fresh host bool ndarray, remembered (write-protected) against its device copy."""
    out = to_host_array(pipeline.unpack(vol))
    _devcache.put(out, vol)
    return out


_STAGE = {}          # the staging array of the last upload, kept for the next one of the same size: a fresh gibibyte costs its
#                      page faults on the way in AND ~25 ms of munmap on the way out (measured: tools/h2hprof.py)


def _staging(shape):
    key = (threading.get_ident(), tuple(shape))
    a = _STAGE.get(key)
    if a is None:
        _STAGE.clear()                      # one staging array per process: another shape replaces it
        a = _STAGE[key] = np.empty(shape, dtype=np.bool_)
    return a


def _stage_masks(mask_images, on_piece=None, piece_bytes=128 << 20):
    """np.stack(mask_images) (voxel_processor.py:46) into a host staging array and, piece by piece behind it, onto the device:
    -> (nz, ny, nx) uint8 0/1 on the device.  on_piece(dev_u8, lo, hi), if given, is called on this thread right after the
    slices [lo, hi) have been uploaded (the pipelined create_voxel_data closes and downloads what it can meanwhile)."""
    from concurrent.futures import ThreadPoolExecutor
    import os
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
    per_slice = max(first.size, 1)
    up = max(1, min(nz, -(-int(piece_bytes) // per_slice)))                                      # slices per upload piece
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
                if on_piece is not None:
                    on_piece(dev.view(torch.uint8), lo, hi)
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
            if on_piece is not None:
                on_piece(dev.view(torch.uint8), lo, hi)
    return dev.view(torch.uint8)


_SIDE = {}           # device -> the stream the pipelined create_voxel_data downloads on
PIPELINED_CREATE = os.environ.get("TOMO_CREATE_PIPELINED", "1") not in ("", "0")
PIPE_PIECE_BYTES = 160 << 20   # > 128 MiB after the one slice a piece waits for: ROCm's fast path for pageable memory (tests shrink it)


def _create_closed_pipelined(mask_images):
    """create_voxel_data(close_ends=True) for a list of separate masks with the upload, the close-ends pass and the download
    of the result running AT THE SAME TIME, piece by piece: the bus is full duplex, and the three steps one after the other
    (19 + 1 + 21 ms at 1024^3) are most of a call that hands a gibibyte in and a gibibyte back.  The closed slice z needs the
    ORIGINAL slices z - 1 and z + 1, so once piece k is up, everything below its last slice is closed
    (tomo_pack_close_range), unpacked and handed to a helper thread that copies it into the result array on a second stream
    while this thread uploads piece k + 1.  The two end slices are packed and filled as soon as they are on the device.
    -> (BitVolume, host bool ndarray) or None when the layout needs the plain sequence (tomo_pack_close_range wants nx % 16 == 0,
    16-byte aligned slices; stacks of fewer than four pieces gain nothing)."""
    from concurrent.futures import ThreadPoolExecutor
    from . import _lib
    first = np.asarray(mask_images[0])
    nz = len(mask_images)
    if first.ndim != 2 or not PIPELINED_CREATE or not pipeline.PACK_CLOSE_FUSED:
        return None
    ny, nx = first.shape
    piece_bytes = PIPE_PIECE_BYTES
    if nx % 16 != 0 or (ny * nx) % 16 != 0 or nz * ny * nx < 4 * piece_bytes or nz < 8:
        return None
    L = _lib.lib()
    dev = _device()
    wx = L.tomo_words_per_row(nx)
    bits = torch.empty((nz, ny, wx), dtype=torch.int64, device=dev)
    scratch = torch.empty(ny * wx + 8, dtype=torch.int64, device=dev)
    out = _hostbuf.take((nz, ny, nx), np.bool_)
    out_t = torch.from_numpy(out)
    side = _SIDE.get(str(dev))
    if side is None:
        side = _SIDE[str(dev)] = torch.cuda.Stream(device=dev)
    st = torch.cuda.current_stream()
    state = {"closed": 0}
    jobs = []

    def download(piece, a, b, ev):
        with torch.cuda.stream(side):
            side.wait_event(ev)
            out_t[a:b].copy_(piece.view(torch.bool))          # returns when the bytes are in `out`

    with ThreadPoolExecutor(1) as pool:
        def on_piece(mask_u8, lo, hi):
            if mask_u8.data_ptr() % 16 != 0:
                raise _lib.TomoError("device mask is not 16-byte aligned")
            s = st.cuda_stream
            for z in ([0] if lo == 0 else []) + ([nz - 1] if hi == nz else []):
                _lib.check(L.tomo_pack_bits(mask_u8[z].data_ptr(), bits[z].data_ptr(), 1, ny, nx, s), "tomo_pack_bits")
                _lib.check(L.tomo_fill_holes_slice(bits.data_ptr(), nz, ny, nx, z, scratch.data_ptr(), s), "tomo_fill_holes_slice")
            a, b = state["closed"], (nz if hi == nz else hi - 1)
            if b <= a:
                return
            _lib.check(L.tomo_pack_close_range(mask_u8.data_ptr(), bits.data_ptr(), nz, ny, nx, a, b, None, None, 1, 1, s),
                       "tomo_pack_close_range")
            piece = pipeline.unpack(pipeline.BitVolume(bits[a:b], (b - a, ny, nx)))
            piece.record_stream(side)
            ev = torch.cuda.Event()
            ev.record(st)
            jobs.append(pool.submit(download, piece, a, b, ev))
            state["closed"] = b
        _stage_masks(mask_images, on_piece, piece_bytes)
        for j in jobs:
            j.result()
    return pipeline.BitVolume(bits, (nz, ny, nx)), out


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


def nz_ok(mask_images):
    return len(mask_images) >= 3


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
        piped = None
        if close_ends:
            if cached is not None:                           # uploaded (and thresholded) by ImageLoader already: closed into a COPY
                vol = pipeline.close_ends(cached)
            elif base is not None:
                vol = pipeline.close_ends(upload_volume(base), inplace=True)           # a fresh upload, never a cached volume: ours to overwrite
            else:
                piped = _create_closed_pipelined(mask_images) if nz_ok(mask_images) else None
                vol = piped[0] if piped is not None else pipeline.pack_closed(_stage_masks(mask_images))
            active = int(pipeline.popcount_async(vol).item())
            if piped is not None:
                self.voxel_data = piped[1]                       # downloaded piece by piece while the stack was going up
                _devcache.put(self.voxel_data, vol)
            else:
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
        vol = to_device_volume(voxel_data)
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
        except Exception as e:                                           # noqa: BLE001 -- the reference catches Exception here
            print(f"tomography_3d_reconstructor_amd: smoothing failed ({e}); closings only", file=sys.stderr)
            return to_host_volume(pipeline.smooth(vol, iterations, False))

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
