"""RCCL through its C API, on the job's own HIP stream.

torch.distributed runs every collective on an internal stream of the process group: each call costs the compute stream
two cross-stream joins (15-25 us each on MI355X, measured with a 1-rank group: tools/slab_selfloop_bench.py) around a
5-15 us kernel, and ~0.1 ms of host time in Python.  The Z-slab job (slab.py) makes four small exchanges per pass with
nothing to overlap them with, so here they are enqueued straight into the stream the kernels run on:
ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd and ncclAllGather of librccl.so (the copy PyTorch-ROCm ships and has
loaded already), bound with ctypes.  Stream order replaces every event: a received buffer is ready for the next kernel
in the stream, a sent one may be freed right after the call (the caching allocator is stream ordered).

The communicator is created next to an initialised torch.distributed process group, which is used once, to hand rank 0's
ncclUniqueId to the other ranks.  Same call surface as slab.TorchDistComm (rank, world, exchange, exchange_async,
all_gather, stats), so SlabJob does not know which one it talks to.
"""
import ctypes
import os
import time

import torch

_UINT8 = 1          # ncclUint8 (rccl.h: ncclDataType_t)
_LIB = None


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]          # NCCL_UNIQUE_ID_BYTES


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("TOMO_RCCL_LIB") or os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        L = ctypes.CDLL(path)          # the handle of the copy torch has loaded already (same path)
        vp, sz, i = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        L.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        L.ncclCommInitRank.argtypes = [ctypes.POINTER(vp), i, _UniqueId, i]
        L.ncclCommDestroy.argtypes = [vp]
        L.ncclSend.argtypes = [vp, sz, i, i, vp, vp]
        L.ncclRecv.argtypes = [vp, sz, i, i, vp, vp]
        L.ncclAllGather.argtypes = [vp, vp, sz, i, vp, vp]
        L.ncclGroupStart.argtypes = []
        L.ncclGroupEnd.argtypes = []
        L.ncclGetErrorString.argtypes = [i]
        L.ncclGetErrorString.restype = ctypes.c_char_p
        for f in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclSend", "ncclRecv", "ncclAllGather",
                  "ncclGroupStart", "ncclGroupEnd"):
            getattr(L, f).restype = i
        _LIB = L
    return _LIB


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, lib().ncclGetErrorString(rc).decode()))


class RcclComm:
    """Neighbour exchange + all-gather of a Z-slab job over one ncclComm, enqueued into the CURRENT torch stream of `device`."""

    def __init__(self, device):
        import torch.distributed as td
        self.device = torch.device(device)
        self.rank, self.world = td.get_rank(), td.get_world_size()
        L = lib()
        uid = _UniqueId()
        # rank 0's id travels together with its verdict: a failure there becomes the SAME exception on every rank instead of
        # rank 0 leaving the broadcast the others wait in
        box = [None, None]
        if self.rank == 0:
            try:
                _check(L.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
                box = [ctypes.string_at(ctypes.addressof(uid), 128), None]                   # the raw 128 bytes (NULs included)
            except Exception as e:                                                         # noqa: BLE001
                box = [None, repr(e)]
        if self.world > 1:
            td.broadcast_object_list(box, src=0)
        if box[0] is None:
            raise RuntimeError("rank 0 could not create an ncclUniqueId: %s" % box[1])
        ctypes.memmove(ctypes.addressof(uid), box[0], 128)
        self._comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _check(L.ncclCommInitRank(ctypes.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")
        self.backend = "rccl-direct"
        self.reset_stats()

    # logical rank of the job -> rank in the communicator (tools/slab_selfloop_bench.py plays a middle rank on a 1-rank comm)
    def _peer(self, r):
        return r

    def reset_stats(self):
        self.stats = {"bytes_sent": 0, "calls": 0, "seconds": 0.0}

    def close(self):
        if self._comm:
            torch.cuda.synchronize(self.device)
            lib().ncclCommDestroy(self._comm)
            self._comm = ctypes.c_void_p()

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def exchange(self, to_prev, to_next, dtype, recv_shape_prev=None, recv_shape_next=None):
        """slab.TorchDistComm.exchange on the current stream: one RCCL group, no events, nothing to wait for."""
        t_in = time.perf_counter()
        L = lib()
        r, w = self.rank, self.world
        sends, recvs = [], []
        from_prev = from_next = None
        if to_next is not None:
            if r + 1 < w and to_next.numel():
                sends.append((to_next.contiguous(), r + 1))
            if r - 1 >= 0:
                from_prev = torch.empty(tuple(recv_shape_prev) if recv_shape_prev is not None else tuple(to_next.shape),
                                        dtype=dtype, device=self.device)
                if from_prev.numel():
                    recvs.append((from_prev, r - 1))
        if to_prev is not None:
            if r - 1 >= 0 and to_prev.numel():
                sends.append((to_prev.contiguous(), r - 1))
            if r + 1 < w:
                from_next = torch.empty(tuple(recv_shape_next) if recv_shape_next is not None else tuple(to_prev.shape),
                                        dtype=dtype, device=self.device)
                if from_next.numel():
                    recvs.append((from_next, r + 1))
        if sends or recvs:
            st = self._stream()
            _check(L.ncclGroupStart(), "ncclGroupStart")
            for t, peer in sends:
                _check(L.ncclSend(t.data_ptr(), t.numel() * t.element_size(), _UINT8, self._peer(peer), self._comm, st), "ncclSend")
            for t, peer in recvs:
                _check(L.ncclRecv(t.data_ptr(), t.numel() * t.element_size(), _UINT8, self._peer(peer), self._comm, st), "ncclRecv")
            _check(L.ncclGroupEnd(), "ncclGroupEnd")
        self.stats["bytes_sent"] += sum(t.numel() * t.element_size() for t, _ in sends)
        self.stats["calls"] += 1
        self.stats["seconds"] += time.perf_counter() - t_in
        return from_prev, from_next

    def exchange_async(self, to_prev, to_next, dtype, recv_shape_prev=None, recv_shape_next=None):
        """Same stream, so there is nothing to join later: what is enqueued next simply runs after the transfer."""
        return self.exchange(to_prev, to_next, dtype, recv_shape_prev, recv_shape_next) + ((lambda: None),)

    def all_gather(self, t):
        t_in = time.perf_counter()
        src = t.contiguous()
        out = torch.empty((self.world,) + tuple(src.shape), dtype=src.dtype, device=self.device)
        _check(lib().ncclAllGather(src.data_ptr(), out.data_ptr(), src.numel() * src.element_size(), _UINT8, self._comm,
                                   self._stream()), "ncclAllGather")
        self.stats["bytes_sent"] += src.numel() * src.element_size()
        self.stats["calls"] += 1
        self.stats["seconds"] += time.perf_counter() - t_in
        return list(out.unbind(0))

    def all_gather_into(self, src, out):
        """out (world, n), contiguous <- every rank's src (n,), on the current stream, no intermediate list / stack."""
        t_in = time.perf_counter()
        src = src.contiguous()
        assert out.is_contiguous() and out.numel() == self.world * src.numel() and out.dtype == src.dtype
        _check(lib().ncclAllGather(src.data_ptr(), out.data_ptr(), src.numel() * src.element_size(), _UINT8, self._comm,
                                   self._stream()), "ncclAllGather")
        self.stats["bytes_sent"] += src.numel() * src.element_size()
        self.stats["calls"] += 1
        self.stats["seconds"] += time.perf_counter() - t_in
        return out
