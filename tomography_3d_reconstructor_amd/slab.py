"""Z-slab sharding of the hot path across the GPUs of one node (one process per GPU).

The reference is single-process; this is the scale-out of the SAME path: the volume is
cut along z, rank r owns slices [z0, z1).  Only small, fixed exchanges happen, all with the z-neighbours
(RCCL send/recv over xGMI) plus one tiny all-gather (the vertex counts):

  close ends   c'[z] = c[z] | (c[z+1] & c'[z-1]) is the local stencil c[z] | (c[z-1] & c[z+1]): a slab needs ORIGINAL slices
  + smoothing  of its neighbours only.  ONE exchange of bit-packed original edge slices (11 up, 12 down) while the middle
               of the slab is packed + closed; every rank closes its neighbours' halo slices itself: 10 below / 11 above
               (8 passes of a radius-1 stencil + 2 for the Gaussian + 1 for the field slice beyond the slab), after which
               every smoothing pass runs locally (the contaminated rim shrinks into the halo).  Slabs thinner than 128
               slices: two exchanges (the stencil's neighbour slices, then the closed halos).
  field        computed locally on the halo-extended slab; the owned field slices are exact.
  marching     needs field slice z1 of rank r+1: computed locally from one more bit-packed halo slice from above
  cubes        (11 instead of 10; no float data travels).
  mesh         vertices on the shared plane are owned by the upper rank: rank r sends the coordinates of its
               top-plane vertices up, gets their indices back, and an all-gather of the per-rank unique counts
               gives the global numbering.  Faces stay on their rank, in reference order.  From a job's second pass
               on none of this waits for the host (SlabJob._numbering_deferred: fixed-capacity messages, counts
               taken from device memory, ONE download per pass; the exact pass is the fallback).

The result is bit-identical to the single-GPU path (tests/test_slab_cpu.py with the CPU oracle as engine over
gloo, tests/test_gpu_parity.py::test_slab_ranks_on_one_gpu_match_single_gpu with the HIP engine).

The class is written against two small interfaces so that the same orchestration code runs on RCCL, on gloo
(CPU tests) and inside one process (two threads sharing one GPU):
  comm   : rank, world, send(t, dst), recv(src), all_gather(t)
  engine : the device operations (HIP kernels in production; tests may inject the CPU oracle)
"""
import os

import numpy as np
import torch

from . import pipeline

HALO_BITS = 10   # 8 morphology passes + 2 slices for the 5-tap Gaussian along z
PC_EDGE = 4      # slices at either end of a slab that wait for the neighbours' slices (the rest is packed + closed first).  Only the
#                  first and the last slice of a slab read a neighbour's slice (the stencil has radius 1 on ORIGINAL slices); 4 = one
#                  group of the streaming kernel's march.  (Rounds 2-3a: 32 -- 64 slices per rank through the short-run kernel at
#                  2/3 of the middle's rate: 78 us of a BASELINE configs[4] rank's pass.)
MERGED_MIN = 128 # slices per rank from which the one-exchange front is used (thinner slabs: the two-exchange front)
SPLIT_PACK = os.environ.get("TOMO_SLAB_SPLIT", "1") not in ("", "0")              # middle of the slab before the neighbour chain (A/B switch)
DEFERRED_NUMBERING = os.environ.get("TOMO_SLAB_DEFERRED", "1") not in ("", "0")   # the pass with ONE download (A/B switch)
# Every host wait of a multi-rank pass has this deadline (seconds; 0 = none).  RCCL's C API has no timeout of its own and the
# collectives sit in the compute stream: a rank that left the job (an error only IT saw -- e.g. the triangle kernel's
# missing-vertex count, which exists only after the all-gather of the summaries and so cannot be decided jointly) would leave
# its neighbours waiting for ever in the next exchange.  With the deadline they raise TomoError instead.
TIMEOUT_S = float(os.environ.get("TOMO_SLAB_TIMEOUT_S", "300"))


# ----------------------------------------------------------------------------- communication
class TorchDistComm:
    """torch.distributed point-to-point + all_gather (backend "nccl" == RCCL on ROCm, or gloo on CPU)."""

    def __init__(self, device):
        import torch.distributed as td
        self.td = td
        self.rank = td.get_rank()
        self.world = td.get_world_size()
        self.device = device
        # gloo moves host memory only: device tensors are staged through the host then (a rehearsal path -- several ranks
        # sharing one GPU, where RCCL refuses to start; the product path is "nccl" = RCCL with device buffers end to end)
        self.stage_host = td.get_backend() == "gloo" and torch.device(device).type != "cpu"
        self.reset_stats()

    def reset_stats(self):
        """bytes_sent: payload this rank handed to send / all_gather; seconds: host wall time inside the calls (it
        includes waiting for the kernels that produce what is sent -- the streams are only joined here)."""
        self.stats = {"bytes_sent": 0, "calls": 0, "seconds": 0.0}

    @staticmethod
    def _bytes(t):
        return t.reshape(-1) if t.dtype == torch.uint8 else t.view(torch.uint8).reshape(-1)

    def exchange(self, to_prev, to_next, dtype, recv_shape_prev=None, recv_shape_next=None, defer=False):
        """Neighbour exchange along z.  `to_prev` goes to rank-1, `to_next` to rank+1; a direction is either used
        by EVERY rank (all pass a tensor; the end ranks' tensor simply has no destination) or by none (all pass
        None).  Returns (from_prev, from_next), None where there is no such neighbour / direction.
        No headers travel: what arrives from rank-1 has the shape of what this rank sends to rank+1 (and vice versa)
        unless recv_shape_prev / recv_shape_next say otherwise (variable-size lists: exchange the counts first).
        All sends and receives of one call go out as ONE batch (a grouped RCCL call: both directions at once, no
        send/recv ordering to deadlock on); empty tensors are skipped on both sides."""
        import time
        t_in = time.perf_counter()
        td = self.td
        r, w = self.rank, self.world
        ops, keep = [], []
        from_prev = from_next = None
        if to_next is not None:
            if r + 1 < w and to_next.numel():
                keep.append(to_next.contiguous())
                ops.append(td.P2POp(td.isend, self._bytes(keep[-1]), r + 1))
            if r - 1 >= 0:
                shape = tuple(recv_shape_prev) if recv_shape_prev is not None else tuple(to_next.shape)
                from_prev = torch.empty(shape, dtype=dtype, device=self.device)
                if from_prev.numel():
                    ops.append(td.P2POp(td.irecv, self._bytes(from_prev), r - 1))
        if to_prev is not None:
            if r - 1 >= 0 and to_prev.numel():
                keep.append(to_prev.contiguous())
                ops.append(td.P2POp(td.isend, self._bytes(keep[-1]), r - 1))
            if r + 1 < w:
                shape = tuple(recv_shape_next) if recv_shape_next is not None else tuple(to_prev.shape)
                from_next = torch.empty(shape, dtype=dtype, device=self.device)
                if from_next.numel():
                    ops.append(td.P2POp(td.irecv, self._bytes(from_next), r + 1))
        works = []
        if ops and self.stage_host:
            staged = []
            hops = []
            for op in ops:
                h = op.tensor.cpu() if op.op is td.isend else torch.empty(op.tensor.shape, dtype=op.tensor.dtype)
                staged.append((op, h))
                hops.append(td.P2POp(op.op, h, op.peer))
            for q in td.batch_isend_irecv(hops):
                q.wait()
            for op, h in staged:
                if op.op is td.irecv:
                    op.tensor.copy_(h)
        elif ops:
            works = td.batch_isend_irecv(ops)
            if not defer:
                for q in works:
                    q.wait()
        self.stats["bytes_sent"] += sum(k.numel() * k.element_size() for k in keep)
        self.stats["calls"] += 1
        self.stats["seconds"] += time.perf_counter() - t_in
        if not defer:
            return from_prev, from_next

        def finish(works=works, keep=keep):
            t0 = time.perf_counter()
            for q in works:
                q.wait()
            self.stats["seconds"] += time.perf_counter() - t0
        return from_prev, from_next, finish

    def exchange_async(self, to_prev, to_next, dtype, recv_shape_prev=None, recv_shape_next=None):
        """exchange() that does not make the current stream wait: -> (from_prev, from_next, finish).  What is enqueued
        between this call and finish() runs next to the transfer; the received tensors may be used only after finish()."""
        return self.exchange(to_prev, to_next, dtype, recv_shape_prev, recv_shape_next, defer=True)

    def all_gather(self, t):
        import time
        t_in = time.perf_counter()
        if self.stage_host:
            h = t.contiguous().cpu()
            out = [torch.empty_like(h) for _ in range(self.world)]
            self.td.all_gather(out, h)
            out = [o.to(self.device) for o in out]
        else:
            out = [torch.empty_like(t) for _ in range(self.world)]
            self.td.all_gather(out, t.contiguous())
        self.stats["bytes_sent"] += t.numel() * t.element_size()
        self.stats["calls"] += 1
        self.stats["seconds"] += time.perf_counter() - t_in
        return out

    def all_gather_into(self, src, out):
        """out (world, n), contiguous <- every rank's src (n,): the gathered table lands where the caller wants it."""
        return _gather_into_via_list(self, src, out)


def _gather_into_via_list(comm, src, out):
    """all_gather_into for a communicator that only has all_gather: out (world, n) <- the ranks' src rows."""
    out.copy_(torch.stack([g.reshape(-1) for g in comm.all_gather(src)]).view_as(out))
    return out


def preflight(comm):
    """First contact of the ranks, before any volume-sized work: ONE neighbour exchange (both directions, the byte-view
    send/recv batch the job uses) and ONE all-gather, checked for content; raises on any mismatch (a transport that
    cannot do these must fail here, inside the process group's timeout, not in the middle of a pass).
    -> {"ranks", "distinct_devices", "devices"} as the process group sees them."""
    dev = comm.device
    r, w = comm.rank, comm.world
    probe = torch.arange(8, dtype=torch.int64, device=dev) + 1000 * r
    lo, hi = comm.exchange(probe, probe + 1, torch.int64)
    if r > 0 and (lo is None or not torch.equal(lo.cpu(), torch.arange(8, dtype=torch.int64) + 1000 * (r - 1) + 1)):
        raise RuntimeError("preflight: rank %d received a wrong halo from rank %d" % (r, r - 1))
    if r + 1 < w and (hi is None or not torch.equal(hi.cpu(), torch.arange(8, dtype=torch.int64) + 1000 * (r + 1))):
        raise RuntimeError("preflight: rank %d received a wrong halo from rank %d" % (r, r + 1))
    ident = [r, -1, 0, 0]
    if torch.device(dev).type == "cuda":
        idx = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
        props = torch.cuda.get_device_properties(idx)
        u = getattr(props, "uuid", None)
        ub = getattr(u, "bytes", None) if u is not None else None
        if ub is not None and len(ub) == 16:
            ident = [r, idx, int.from_bytes(ub[:7], "little"), int.from_bytes(ub[8:15], "little")]
        else:
            ident = [r, idx, int(getattr(props, "pci_bus_id", idx)), int(getattr(props, "pci_domain_id", 0))]
    got = comm.all_gather(torch.tensor(ident, dtype=torch.int64, device=dev))
    rows = [[int(x) for x in g.cpu()] for g in got]
    if [row[0] for row in rows] != list(range(w)):
        raise RuntimeError("preflight: all_gather returned ranks %r" % ([row[0] for row in rows],))
    devices = [tuple(row[1:]) for row in rows]
    return {"ranks": w, "distinct_devices": len(set(devices)), "devices": [d[0] for d in devices],
            "backend": getattr(comm, "backend", None) or (comm.td.get_backend() if hasattr(comm, "td") else "threads")}


class ThreadComm:
    """In-process communicator for tests: `world` threads, queues between them (same call surface)."""

    def __init__(self, rank, world, queues, barrier, gather_slots):
        self.rank, self.world = rank, world
        self.q, self.barrier, self.slots = queues, barrier, gather_slots

    @staticmethod
    def _publish(t):
        """A copy of t that is COMPLETE when another thread (working on its own stream) picks it up."""
        c = t.clone()
        if c.is_cuda:
            torch.cuda.current_stream(c.device).synchronize()
        return c

    def send(self, t, dst):
        self.q[(self.rank, dst)].put(self._publish(t))

    @staticmethod
    def _adopt(t):
        """A tensor another thread allocated (on ITS stream) is about to be read by kernels of this thread's stream: tell the
        caching allocator, or the block goes back to the other stream's pool -- and is handed out and overwritten there --
        as soon as this thread drops the reference, which in a pass without host round trips is long before its kernels ran."""
        if t is not None and t.is_cuda:
            t.record_stream(torch.cuda.current_stream(t.device))
        return t

    def recv(self, src, dtype):
        return self._adopt(self.q[(src, self.rank)].get(timeout=120))

    def exchange(self, to_prev, to_next, dtype, recv_shape_prev=None, recv_shape_next=None):
        r, w = self.rank, self.world
        if r + 1 < w and to_next is not None:
            self.send(to_next, r + 1)
        if r - 1 >= 0 and to_prev is not None:
            self.send(to_prev, r - 1)
        from_prev = self.recv(r - 1, dtype) if (r - 1 >= 0 and to_next is not None) else None
        from_next = self.recv(r + 1, dtype) if (r + 1 < w and to_prev is not None) else None
        return from_prev, from_next

    def exchange_async(self, to_prev, to_next, dtype, recv_shape_prev=None, recv_shape_next=None):
        return self.exchange(to_prev, to_next, dtype, recv_shape_prev, recv_shape_next) + ((lambda: None),)

    def all_gather(self, t):
        self.slots[self.rank] = self._publish(t)
        self.barrier.wait()
        out = [self._adopt(self.slots[i]) for i in range(self.world)]
        self.barrier.wait()
        return out

    def all_gather_into(self, src, out):
        return _gather_into_via_list(self, src, out)

    @staticmethod
    def make(world):
        import queue
        import threading
        qs = {(a, b): queue.Queue() for a in range(world) for b in range(world) if a != b}
        bar = threading.Barrier(world)
        slots = [None] * world
        return [ThreadComm(r, world, qs, bar, slots) for r in range(world)]


# ----------------------------------------------------------------------------- the HIP engine
class HipEngine:
    """Device operations of the slab job, on BitVolume / torch tensors (libtomo_hip.so kernels)."""

    def pack(self, mask):
        return pipeline.pack(mask)

    def pack_into(self, mask, room):
        """Pack the rank's slices into the middle of ONE buffer with `room` spare slices on either side, so that the
        neighbour slices of the close-ends chain and the morphology halo are written next to them instead of being
        concatenated (two copies of the whole slab less).  -> (buffer bits (nzl + 2 room, ny, words), BitVolume of the middle)."""
        from . import _lib
        nzl, ny, nx = mask.shape
        wx = _lib.lib().tomo_words_per_row(nx)
        buf = torch.empty((nzl + 2 * room, ny, wx), dtype=torch.int64, device=mask.device)
        return buf, pipeline.pack(mask, out=buf[room:room + nzl])

    def pack_closed_slab(self, mask, room, comm, first, last, halo=None, merged=False):
        """np.stack + _close_volume_ends for this rank's slices in ONE pass over its mask (pipeline.pack_closed's slab form):
        the first and last slice are packed alone (and filled where they are global end slices), their ORIGINAL content goes
        to the neighbours, and the fused kernel packs everything else with the neighbours' slices closing the stencil.
        The middle of the slab depends on the mask alone and is enqueued FIRST: the GPU streams through it while the host
        is busy with the first RCCL call of the pass (after the download that ended the previous pass the host has no lead:
        the neighbour chain in front cost ~0.2 ms of idle GPU per pass); the PC_EDGE slices at either end follow once the
        neighbours' slices are there, then (halo = (H, Hu)) the closed halo slices are exchanged into the buffer.
        -> (halo buffer (nzl + 2 room, ny, words), bits of the closed slab = its middle), or None if the layout needs the
        separate kernels (nx % 16 != 0, fewer than 2 slices)."""
        from . import _lib
        L = _lib.lib()
        nzl, ny, nx = mask.shape
        mask = mask.contiguous()
        if merged and mask.data_ptr() % 16 != 0:
            mask = mask.clone()     # the one-exchange front is a JOINT decision: a rank must not drop out of it over the address
            #                         of its own mask (the other paths exchange differently)
        if nzl < 2 or nx % 16 != 0 or mask.data_ptr() % 16 != 0 or not pipeline.PACK_CLOSE_FUSED:
            return None
        E = PC_EDGE
        split = nzl >= 4 * E and SPLIT_PACK
        st = torch.cuda.current_stream().cuda_stream
        wx = L.tomo_words_per_row(nx)
        buf = torch.empty((nzl + 2 * room, ny, wx), dtype=torch.int64, device=mask.device)
        own = buf[room:room + nzl]
        scratch = torch.empty(ny * wx + 8, dtype=torch.int64, device=mask.device)
        lo_f, hi_f = (1 if first else 0), (1 if last else 0)

        def close_range(z_from, z_to, pb=None, pa=None):
            _lib.check(L.tomo_pack_close_range(mask.data_ptr(), own.data_ptr(), nzl, ny, nx, z_from, z_to, pb, pa, lo_f, hi_f, st),
                       "tomo_pack_close_range")

        if merged and halo is not None:
            # ONE exchange per pass in front of the smoothing: the ORIGINAL edge slices go out (Hu + 1 down, H + 1 up) while
            # the middle of the slab streams -- its first part covers the host's time in the RCCL call, the rest the
            # transfer -- and every rank closes its neighbours' halo slices itself (the stencil needs one slice more on
            # either side: the reason for the + 1) instead of receiving them closed in a second exchange.
            H, Hu = halo
            assert room >= Hu >= H and nzl >= MERGED_MIN and nzl >= 4 * E and nzl > 2 * (Hu + 1)
            zc = (E + (nzl - 2 * E) * 3 // 4) // 4 * 4        # 3/4 before the RCCL call (its host side takes ~0.15 ms), 1/4 after
            close_range(E, zc)
            edge_lo = torch.empty((Hu + 1, ny, wx), dtype=torch.int64, device=mask.device)
            edge_hi = torch.empty((H + 1, ny, wx), dtype=torch.int64, device=mask.device)
            _lib.check(L.tomo_pack_bits_pair(mask.data_ptr(), edge_lo.data_ptr(), Hu + 1, mask[nzl - (H + 1):].data_ptr(), edge_hi.data_ptr(),
                                             H + 1, ny, nx, st), "tomo_pack_bits_pair")
            if first:
                own[0].copy_(edge_lo[0])
                _lib.check(L.tomo_fill_holes_slice(own.data_ptr(), nzl, ny, nx, 0, scratch.data_ptr(), st), "tomo_fill_holes_slice")
            if last:
                own[nzl - 1].copy_(edge_hi[H])
                _lib.check(L.tomo_fill_holes_slice(own.data_ptr(), nzl, ny, nx, nzl - 1, scratch.data_ptr(), st), "tomo_fill_holes_slice")
            from_prev, from_next, finish = comm.exchange_async(edge_lo, edge_hi, torch.int64)
            close_range(zc, nzl - E)
            finish()
            # ONE launch for what is left: the E slices at either end of the slab (they needed the neighbours' slices) and the
            # stencil on the neighbours' halo slices -- closed slices z0 - H .. z0 - 1 from the originals z0 - H - 1 .. z0 - 1 and
            # this rank's own z0; closed slices z1 .. z1 + Hu - 1 from this rank's own z1 - 1 and the originals z1 .. z1 + Hu
            pb, pa = (None if first else from_prev[H].data_ptr()), (None if last else from_next[0].data_ptr())
            lo = (None, None, None, 0, None) if first else (from_prev[0].data_ptr(), from_prev[1:].data_ptr(), edge_lo[0].data_ptr(), H,
                                                            buf[room - H:room].data_ptr())
            hi = (None, None, None, 0, None) if last else (edge_hi[H].data_ptr(), from_next[:Hu].data_ptr(), from_next[Hu].data_ptr(), Hu,
                                                           buf[room + nzl:room + nzl + Hu].data_ptr())
            _lib.check(L.tomo_slab_edges(mask.data_ptr(), own.data_ptr(), nzl, ny, nx, E, pb, pa, lo_f, hi_f, *lo, *hi, st), "tomo_slab_edges")
            return buf, own
        if split:
            close_range(E, nzl - E)
        for z in (0, nzl - 1):
            _lib.check(L.tomo_pack_bits(mask[z].data_ptr(), own[z].data_ptr(), 1, ny, nx, st), "tomo_pack_bits")
        if first:
            _lib.check(L.tomo_fill_holes_slice(own.data_ptr(), nzl, ny, nx, 0, scratch.data_ptr(), st), "tomo_fill_holes_slice")
        if last:
            _lib.check(L.tomo_fill_holes_slice(own.data_ptr(), nzl, ny, nx, nzl - 1, scratch.data_ptr(), st), "tomo_fill_holes_slice")
        below, above = comm.exchange(own[:1], own[nzl - 1:], torch.int64)
        pb = None if below is None else below.data_ptr()
        pa = None if above is None else above.data_ptr()
        for z_from, z_to in (((0, E), (nzl - E, nzl)) if split else ((0, nzl),)):
            _lib.check(L.tomo_pack_close_range(mask.data_ptr(), own.data_ptr(), nzl, ny, nx, z_from, z_to, pb, pa, lo_f, hi_f, st),
                       "tomo_pack_close_range")
        if halo is not None:
            H, Hu = halo
            assert room >= Hu >= H
            lo, hi = comm.exchange(own[:Hu], own[nzl - H:], torch.int64)
            if not first:
                buf[room - H:room].copy_(lo)
            if not last:
                buf[room + nzl:room + nzl + Hu].copy_(hi)
        return buf, own

    def bits(self, vol):
        return vol.bits

    def from_bits(self, bits, shape):
        return pipeline.BitVolume(bits.contiguous(), tuple(shape))

    def fill_holes(self, vol, z):
        from . import _lib
        nz, ny, nx = vol.shape
        scratch = torch.empty(ny * vol.bits.shape[2] + 8, dtype=torch.int64, device=vol.bits.device)
        _lib.check(_lib.lib().tomo_fill_holes_slice(vol.bits.data_ptr(), nz, ny, nx, z, scratch.data_ptr(),
                                                    torch.cuda.current_stream().cuda_stream), "tomo_fill_holes_slice")

    def close_gp(self, vol):
        """(G, P) bit planes of the carry chain over slices 1 .. nz-2 of `vol` (nz >= 3)."""
        from . import _lib
        L = _lib.lib()
        nz, ny, nx = vol.shape
        wx = vol.bits.shape[2]
        ws = torch.empty(L.tomo_close_ends_workspace_words(nz, ny, nx), dtype=torch.int64, device=vol.bits.device)
        gp = torch.empty((2, ny, wx), dtype=torch.int64, device=vol.bits.device)
        _lib.check(L.tomo_close_ends_gp(vol.bits.data_ptr(), nz, ny, nx, ws.data_ptr(), gp.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream), "tomo_close_ends_gp")
        return gp[0], gp[1]

    def close_scan(self, vol):
        from . import _lib
        L = _lib.lib()
        nz, ny, nx = vol.shape
        if nz > 2:
            ws = torch.empty(L.tomo_close_ends_workspace_words(nz, ny, nx), dtype=torch.int64, device=vol.bits.device)
            _lib.check(L.tomo_close_ends_scan(vol.bits.data_ptr(), nz, ny, nx, ws.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream), "tomo_close_ends_scan")

    def popcount(self, vol):
        return int(pipeline.popcount_async(vol).item())

    def smooth(self, vol, iterations, create_manifold):
        return pipeline.smooth(vol, iterations, create_manifold)

    def field(self, vol):
        return pipeline.make_field(vol, True, True)

    def field_slices(self, f, a, b):
        return pipeline.Field(f.data[a:b], b - a, f.Ny, f.Nx, f.pitch, f.xorg,
                              f.signs[a:b] if f.signs is not None else None, f.signs_level,
                              f.gcls[a:b] if f.gcls is not None else None)

    def marching_cubes(self, f, z_offset):
        return pipeline.marching_cubes(f, 0.5, z_offset)

    def finalize_vertices(self, vpos, depths, mm_y, mm_x):
        return pipeline.finalize_vertices(vpos, depths, mm_y, mm_x, True, True)

    def unique(self, vpos):
        """-> (uniq (U,3), rank (V,) int32 final index of every input row)."""
        from . import _lib
        L = _lib.lib()
        dev = vpos.device
        nv = vpos.shape[0]
        totals = torch.zeros(4, dtype=torch.int64, device=dev)
        uniq = torch.empty((nv, 3), dtype=torch.float32, device=dev)
        rank = torch.empty(nv, dtype=torch.int32, device=dev)
        wsb = L.tomo_mesh_unique_workspace_bytes(nv)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        _lib.check(L.tomo_mesh_unique(vpos.contiguous().data_ptr(), nv, uniq.data_ptr(), rank.data_ptr(), totals.data_ptr(),
                                      ws.data_ptr(), wsb, torch.cuda.current_stream().cuda_stream), "tomo_mesh_unique")
        return uniq[: int(totals[0].item())], rank


    # optional fast paths (engines without them -- the CPU oracle engine of the tests -- use unique() and torch ops)
    def unique_mc(self, vpos, vkey, ny, nz=0):
        """unique() for rows in marching-cubes order with their keys: one-sort path with automatic fallback."""
        return pipeline.unique_rows(vpos, vkey, ny, nz)

    def lookup(self, uniq, query):
        return pipeline.lookup_rows(uniq, query)

    def lookup_async(self, uniq, query):
        """lookup() without the host round trip: the miss count stays on the device (a 1-element int64 tensor)."""
        return pipeline.lookup_rows(uniq, query, sync=False)

    def remap_faces(self, faces32, gid32):
        return pipeline.remap_faces(faces32, gid32)

    def mc3_vertices(self, f, z_offset, depths, mm_y, mm_x, z_top=None, defer=False, tot=None):
        """The production chain (pipeline.mc3_vertices): finalised, sorted, duplicate-free vertex rows of this slab plus
        the table vertex id -> row index; the triangles are written later, through a table of GLOBAL indices.
        z_top: the mapped z of the plane shared with the rank above (its rows are counted on the device); defer: return
        without reading any count when size hints exist (SlabJob._numbering_deferred reads them, once, at the end)."""
        if not pipeline.MC3:
            return NotImplemented
        return pipeline.mc3_vertices(f, depths, mm_y, mm_x, True, z_offset=z_offset, with_faces=False, z_top=z_top, defer=defer, tot=tot)

    def mc3_ready(self, f, z_offset):
        return pipeline.mc3_hint_ready(f, z_offset)

    # device side of the numbering without host round trips (csrc/mesh.hip "Z-slab numbering on the device")
    def slab_top_rows(self, uniq, tot, cap_v, cap):
        from . import _lib
        msg = torch.empty((cap + 1) * 3, dtype=torch.float32, device=uniq.device)
        _lib.check(_lib.lib().tomo_slab_top_rows(uniq.data_ptr(), tot.data_ptr(), cap_v, cap, msg.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream), "tomo_slab_top_rows")
        return msg

    def slab_lookup(self, uniq, tot, cap_v, msg, cap, miss=None):
        """miss: a zeroed int64[1] the caller keeps between passes (slab_summary reads and clears it), or None: a fresh one."""
        from . import _lib
        out = torch.empty(cap, dtype=torch.int32, device=uniq.device)
        if miss is None:
            miss = torch.zeros(1, dtype=torch.int64, device=uniq.device)
        _lib.check(_lib.lib().tomo_slab_lookup(uniq.data_ptr(), tot.data_ptr(), cap_v, msg.data_ptr(), cap, out.data_ptr(),
                                               miss.data_ptr(), torch.cuda.current_stream().cuda_stream), "tomo_slab_lookup")
        return out, miss

    def slab_summary(self, tot, cap_v, msg_in, miss, cap_top, caller_flags):
        from . import _lib
        out = torch.empty(8, dtype=torch.int64, device=tot.device)
        _lib.check(_lib.lib().tomo_slab_summary(tot.data_ptr(), cap_v, None if msg_in is None else msg_in.data_ptr(),
                                                None if miss is None else miss.data_ptr(), cap_top, caller_flags, out.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream), "tomo_slab_summary")
        return out

    def slab_lookup_summary(self, uniq, tot, cap_v, msg_in, cap, cap_top, caller_flags, summary):
        """slab_lookup + slab_summary in one launch; `summary` = int64[8] to fill."""
        from . import _lib
        key = (str(uniq.device), torch.cuda.current_stream().cuda_stream)
        scratch = self._ls_scratch.get(key) if hasattr(self, "_ls_scratch") else None
        if scratch is None:                         # two ticket / miss words per stream, zeroed once: every call leaves them zero
            if not hasattr(self, "_ls_scratch"):
                self._ls_scratch = {}
            scratch = self._ls_scratch[key] = torch.zeros(2, dtype=torch.int64, device=uniq.device)
        out = None if msg_in is None else torch.empty(max(cap, 1), dtype=torch.int32, device=uniq.device)
        _lib.check(_lib.lib().tomo_slab_lookup_summary(uniq.data_ptr(), tot.data_ptr(), cap_v, None if msg_in is None else msg_in.data_ptr(),
                                                       cap, None if out is None else out.data_ptr(), cap_top, caller_flags,
                                                       scratch.data_ptr(), summary.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   "tomo_slab_lookup_summary")
        return out

    def download(self, t):
        return pipeline._download_vec(t)

    def download_start(self, t):
        """-> an object whose wait() returns the list of ints (the copy runs on the current stream, into its own page-locked buffer)."""
        return pipeline.PendingDownload(t)

    # reductions of the consumers (volume_calculator.py:23-35, 59-94) on the resident bit volume
    def slice_counts(self, vol):
        return pipeline.slice_counts(vol)

    def bbox(self, vol):
        return pipeline.bounding_box(vol)


# ----------------------------------------------------------------------------- the job
def slab_range(gz, rank, world):
    base, rem = divmod(gz, world)
    z0 = rank * base + min(rank, rem)
    return z0, z0 + base + (1 if rank < rem else 0)


def balanced_cuts(weights, world, min_slices=1):
    """Z cuts [0 = c_0 < c_1 < .. < c_world = gz] that give every rank about the same total WEIGHT (one number per slice: its
    share of a pass's time) instead of the same number of slices; every slab keeps at least min_slices slices.  Deterministic:
    every rank computes the same cuts from the same (all-gathered) weights."""
    w = np.maximum(np.asarray(weights, dtype=np.float64), 0.0)
    gz = len(w)
    if world < 1 or gz < world * max(min_slices, 1):
        raise ValueError("%d slices cannot be cut into %d slabs of at least %d" % (gz, world, min_slices))
    cum = np.concatenate([[0.0], np.cumsum(w)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        c = int(np.searchsorted(cum, total * r / world, side="left")) if total > 0 else gz * r // world
        if c > 0 and c <= gz and abs(cum[c - 1] - total * r / world) <= abs(cum[min(c, gz)] - total * r / world):
            c -= 1                                               # the nearer of the two candidates
        c = max(c, cuts[-1] + min_slices)                        # room for this slab ...
        c = min(c, gz - (world - r) * min_slices)                # ... and for those above
        cuts.append(c)
    cuts.append(gz)
    return cuts


class SlabJob:
    """One rank's share of: close ends -> smooth -> field -> marching cubes -> global mesh numbering."""

    def __init__(self, gz, ny, nx, comm, engine=None, iterations=3, create_manifold=True, close_ends=True, z_cuts=None):
        """z_cuts: world + 1 ascending slice indices from 0 to gz -- rank r owns [z_cuts[r], z_cuts[r + 1]) -- or None: equal
        numbers of slices (slab_range).  Every rank must pass the same cuts (balanced_cuts / SlabJob.work_balanced_cuts give them)."""
        self.gz, self.ny, self.nx = gz, ny, nx
        self.comm = comm
        self.rank, self.world = comm.rank, comm.world
        if z_cuts is None:
            z_cuts = [slab_range(gz, r, self.world)[0] for r in range(self.world)] + [gz]
        self.cuts = [int(c) for c in z_cuts]
        if len(self.cuts) != self.world + 1 or self.cuts[0] != 0 or self.cuts[-1] != gz or any(b <= a for a, b in zip(self.cuts, self.cuts[1:])):
            raise ValueError("z_cuts must be %d ascending slice indices from 0 to %d" % (self.world + 1, gz))
        self.z0, self.z1 = self.cuts[self.rank], self.cuts[self.rank + 1]
        self.eng = engine or HipEngine()
        self.iterations, self.create_manifold, self.close_ends = iterations, create_manifold, close_ends
        self.halo = 2 * (self.iterations + (1 if create_manifold else 0)) + 2
        self.thinnest = min(b - a for a, b in zip(self.cuts, self.cuts[1:]))
        if self.thinnest < self.halo + 1 and self.world > 1:
            raise ValueError("slab thinner than the halo (%d slices): use fewer ranks" % self.halo)
        self.active = None
        self.created = self.smoothed = self.mesh = None     # this rank's share of the last run(), for the consumers below
        # the pass without host round trips (_numbering_deferred): agreed by ALL ranks in the last pass's all-gather, with the
        # message capacities both neighbours derive from that pass's exact shared-plane counts
        self._deferred_ok = False
        self._cap_top = self._cap_prev = 0
        self.deferred_passes = self.deferred_redone = 0
        self.deferred_why = None

    # -- step 1: closed slab (bits tensor of the owned slices)
    def _close_ends(self, vol, buf=None, room=0):
        """buf / room: the halo-extended buffer `vol` sits in the middle of (HipEngine.pack_into), or None."""
        e, c = self.eng, self.comm
        first, last = self.rank == 0, self.rank == self.world - 1
        nzl = self.z1 - self.z0
        if first:
            e.fill_holes(vol, 0)
        if last:
            e.fill_holes(vol, nzl - 1)
        bits = e.bits(vol)
        if self.world == 1:
            e.close_scan(vol)
            return e.bits(vol)
        # c'[z] = c[z] | (c'[z-1] & c[z+1]) is the LOCAL stencil c[z] | (c[z-1] & c[z+1]) (where c[z] = 0 the updated
        # neighbour c'[z-1] is just c[z-1]: DESIGN.md 4.2), so a slab needs exactly one ORIGINAL slice from either
        # neighbour -- its last slice from below, its first from above -- and no carry from further away: one two-way
        # exchange, then the recurrence kernel on [slice below | slab | slice above] with those two as fixed ends.
        below, above = c.exchange(bits[:1], bits[nzl - 1:], torch.int64)
        if buf is not None:     # the two neighbour slices go in place, next to the slab
            lo_i, hi_i = room - (0 if first else 1), room + nzl + (0 if last else 1)
            if not last:
                buf[room + nzl].copy_(above[0])
            if not first:
                buf[room - 1].copy_(below[0])
            ext = e.from_bits(buf[lo_i:hi_i], (hi_i - lo_i, self.ny, self.nx))
        else:
            parts = ([] if first else [below]) + [bits] + ([] if last else [above])
            ext = e.from_bits(torch.cat(parts, 0), (nzl + len(parts) - 1, self.ny, self.nx))
        e.close_scan(ext)
        xb = e.bits(ext)
        return xb[(0 if first else 1):(0 if first else 1) + nzl]

    # -- the whole pass
    def run(self, mask, slice_depths, mm_y, mm_x):
        """mask: this rank's slices (nzl, ny, nx) uint8/bool on the device.  Returns (vertices, faces) of THIS
        rank: its unique vertices (globally sorted across ranks) and its faces with GLOBAL vertex indices, plus
        sets self.vertex_offset / self.n_vertices_global."""
        return self.result(self.submit(mask, slice_depths, mm_y, mm_x))

    def result(self, ticket):
        """Second half of run(): waits for the ONE download of a pass that submit() enqueued (a pass that could not run
        that way was completed inside submit()), checks the summaries of all ranks -- every rank decides alike and repeats
        the pass the exact way if anything did not fit -- and publishes the pass as the job's current one (mesh, created /
        smoothed volumes for the consumers below, vertex_offset / n_vertices_global)."""
        if ticket.get("pending") is not None:
            ticket["mesh"] = self._numbering_deferred_finish(ticket.pop("pending"))
            ticket["pending"] = None
            ticket["numbering"] = (self.vertex_offset, self.n_vertices_global)
        self.vertex_offset, self.n_vertices_global = ticket["numbering"]
        self.created, self.smoothed, self.mesh = ticket["created"], ticket["smoothed"], ticket["mesh"]
        return self.mesh

    def submit(self, mask, slice_depths, mm_y, mm_x):
        """First half of run(): enqueues the whole pass and returns a ticket for result().  From a job's second pass on
        (deferred numbering agreed by all ranks) nothing here waits for the GPU, so a caller with a sequence of stacks can
        submit stack n + 1 BEFORE it asks for the result of stack n: the host's read of the counters then no longer leaves
        the GPU idle between passes (~0.1 ms per pass in the one-GPU rehearsal).  Every rank must interleave submit() and
        result() the same way (they issue collective steps)."""
        e, c = self.eng, self.comm
        first, last = self.rank == 0, self.rank == self.world - 1
        nzl = self.z1 - self.z0
        H = self.halo
        Hu = H + 1          # one more slice from above: the field slice marching cubes needs beyond the slab is computed HERE
        buf = None
        fused = None
        halo_done = False
        if self.world > 1 and self.close_ends and hasattr(e, "pack_closed_slab") and mask.dtype in (torch.uint8, torch.bool):
            # one exchange of ORIGINAL edge slices instead of two (stencil neighbours, then closed halos): a joint decision,
            # so it goes by the thinnest slab of the job, not by this rank's
            merged = (SPLIT_PACK and self.thinnest >= MERGED_MIN and hasattr(c, "exchange_async")
                      and self.nx % 16 == 0 and pipeline.PACK_CLOSE_FUSED)
            fused = e.pack_closed_slab(mask.view(torch.uint8) if mask.dtype == torch.bool else mask, Hu, c, first, last, (H, Hu), merged)
        if fused is not None:
            buf, bits = fused                               # packed and closed in one pass over the mask, halos in place
            halo_done = True
        else:
            if self.world > 1 and hasattr(e, "pack_into"):
                buf, vol = e.pack_into(mask, Hu)            # the slab in the middle of its halo-extended buffer
            else:
                vol = e.pack(mask)
            bits = self._close_ends(vol, buf, Hu) if self.close_ends else e.bits(vol)
        closed = e.from_bits(bits, (nzl, self.ny, self.nx))
        # halo for morphology + Gaussian: H closed slices from below, H + 1 from above
        if self.world > 1:
            if not halo_done:
                lo, hi = c.exchange(bits[:Hu], bits[nzl - H:], torch.int64)
            if buf is not None:                             # room is Hu on either side: the lower halo leaves one slice unused
                if not halo_done:
                    if not first:
                        buf[Hu - H:Hu].copy_(lo)
                    if not last:
                        buf[Hu + nzl:].copy_(hi)
                a0, b0 = (Hu if first else Hu - H), Hu + nzl + (0 if last else Hu)
                ext = e.from_bits(buf[a0:b0], (b0 - a0, self.ny, self.nx))
            else:
                parts = ([] if first else [lo]) + [bits] + ([] if last else [hi])
                ext = e.from_bits(torch.cat(parts, 0), (nzl + (0 if first else H) + (0 if last else Hu), self.ny, self.nx))
        else:
            ext = closed
        sm = e.smooth(ext, self.iterations, self.create_manifold)
        sb = e.bits(sm)
        own = 0 if (first or self.world == 1) else H
        ticket = {"created": closed, "smoothed": e.from_bits(sb[own:own + nzl], (nzl, self.ny, self.nx)), "mesh": None, "pending": None}
        # the field of slices [z0 - 2, z1 + 3): exact on the owned slices AND on slice z1 -- the first owned slice of the
        # rank above, the one field slice marching cubes reads beyond the slab (round 1 / the first half of round 2 sent it
        # down as a float32 slice: 4 MB per neighbour and an exchange step, against one more bit-packed halo slice)
        a = 0 if first else H - 2
        b = sb.shape[0] - (0 if last else Hu - 3)
        fvol = e.from_bits(sb[a:b], (b - a, self.ny, self.nx))
        f = e.field(fvol)
        fa = 0 if first else 3
        fb = f.Nz - (0 if last else 4)              # local index of slice z1 (not last) / one past the padded volume (last)
        f = e.field_slices(f, fa, fb + (0 if last else 1))
        Za = 0 if first else self.z0 + 1            # global padded index of the first owned slice
        dev = mask.device
        if hasattr(e, "mc3_vertices"):
            z_top = None if (last or self.world == 1) else self._z_top(slice_depths, dev)
            deferred = self._deferred_ok and DEFERRED_NUMBERING and self.world > 1 and hasattr(e, "slab_lookup_summary")
            # the counters of a deferred pass live in ONE buffer -- the chain's eight, then every rank's summary -- so that one
            # copy brings them to the host
            counters = torch.empty(8 + 8 * self.world, dtype=torch.int64, device=dev) if deferred else None
            m = (e.mc3_vertices(f, Za, slice_depths, mm_y, mm_x, z_top=z_top, defer=True, tot=counters[:8]) if deferred else
                 e.mc3_vertices(f, Za, slice_depths, mm_y, mm_x, z_top=z_top, defer=False))
            if m is not NotImplemented:
                ready = m is not None and hasattr(e, "mc3_ready") and bool(e.mc3_ready(f, Za))
                if deferred:
                    ticket["pending"] = self._numbering_deferred(m, f, Za, slice_depths, mm_y, mm_x, z_top, dev, counters)
                else:
                    ticket["mesh"] = self._global_numbering_mc3(m, slice_depths, dev, ready)
                    ticket["numbering"] = (self.vertex_offset, self.n_vertices_global)
                return ticket
        mesh = e.marching_cubes(f, Za)
        vkey = ny = None
        if mesh is None:
            vpos = torch.zeros((0, 3), dtype=torch.float32, device=dev)
            faces32 = torch.zeros((0, 3), dtype=torch.int32, device=dev)
        else:
            vpos, faces32 = mesh.vpos, mesh.faces32
            vkey, ny = getattr(mesh, "vkey", None), getattr(mesh, "_ny", None)
            self._mesh_nz = getattr(mesh, "_nz", 0)
            e.finalize_vertices(vpos, slice_depths, mm_y, mm_x)
        ticket["mesh"] = self._global_numbering(vpos, faces32, slice_depths, dev, vkey, ny)
        ticket["numbering"] = (self.vertex_offset, self.n_vertices_global)
        return ticket

    # -- step 5: vertices on the plane shared with rank+1 belong to rank+1
    def _z_top(self, slice_depths, dev):
        """Mapped z (float32, through the vertex finalisation arithmetic) of padded plane z1 + 1: the plane this rank shares
        with the rank above.  Depends on the depth table only: computed once per table."""
        zkey = np.asarray(slice_depths, dtype=np.float64).tobytes()
        if getattr(self, "_zb_key", None) != zkey:
            zt = torch.tensor([[float(self.z1 + 1), 1.0, 1.0]], dtype=torch.float32, device=dev)
            self._zb, self._zb_key = float(self.eng.finalize_vertices(zt, slice_depths, 1.0, 1.0)[0, 0].item()), zkey
        return self._zb

    @staticmethod
    def _msg_cap(n):
        return int(n * 1.25) + 64

    def _await(self, dev):
        """Wait for the current stream of `dev` with the job's deadline (TIMEOUT_S) before a host read that follows a collective
        step: a neighbour that has left becomes a TomoError here instead of a hang inside .cpu()."""
        if self.world > 1 and TIMEOUT_S and torch.device(dev).type == "cuda":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            pipeline.wait_event(ev, TIMEOUT_S, "a collective step of the slab job")

    def _number_rows(self, uniq, slice_depths, dev, ready=False):
        """uniq: this rank's sorted, duplicate-free vertex rows (nu, 3).  -> (the rows this rank keeps, gid int64 (nu,):
        the GLOBAL index of every one of the nu rows).  Sets self.vertex_offset / self.n_vertices_global.
        ready: this rank could run its next pass from size hints; the all-gather below makes it a joint decision."""
        e, c = self.eng, self.comm
        first, last = self.rank == 0, self.rank == self.world - 1
        nu0 = nu = uniq.shape[0]
        if self.world == 1:
            self.vertex_offset, self.n_vertices_global = 0, nu
            self._numbering = None
            return uniq, torch.arange(nu, dtype=torch.int64, device=dev)
        # the rows on the plane shared with rank+1 (mapped z of padded plane z1 + 1, through the same finalisation
        # arithmetic) have the largest z here, so they close the sorted list; they belong to rank+1.  Their number is
        # counted on the device, handed up as a device scalar, and ONE download brings this rank's and the lower rank's.
        nt = torch.zeros(1, dtype=torch.int64, device=dev)
        if not last and nu:
            nt = (uniq[:, 0] == self._z_top(slice_depths, dev)).sum().reshape(1).to(torch.int64)
        cnt_prev, _ = c.exchange(None, nt, torch.int64)
        self._await(dev)
        pair = torch.cat([nt, cnt_prev.reshape(1) if cnt_prev is not None else torch.zeros_like(nt)]).cpu()
        n_top, n_from_prev = int(pair[0]), int(pair[1])
        k = nu - n_top                                             # rows this rank keeps
        # they go up; what arrives from below are copies of rows this rank has itself: look them up
        from_prev, _ = c.exchange(None, uniq[k:].contiguous(), torch.float32, recv_shape_prev=(n_from_prev, 3))
        idx_prev = torch.zeros(0, dtype=torch.int32, device=dev)
        remap = None                                               # old row index -> row index after a merge
        prev_rows = None
        miss = torch.zeros(1, dtype=torch.int64, device=dev)
        if from_prev is not None and from_prev.shape[0]:
            prev_rows = from_prev.reshape(-1, 3).contiguous()
            if hasattr(e, "lookup_async") and nu:
                idx_prev, miss = e.lookup_async(uniq, prev_rows)
            elif hasattr(e, "lookup") and nu:
                idx_prev, m = e.lookup(uniq, prev_rows)
                miss = torch.full((1,), int(m), dtype=torch.int64, device=dev)
            else:
                miss = torch.ones(1, dtype=torch.int64, device=dev)   # an engine without lookup: merge the two lists
        # the kept counts give every rank its offset; the same all-gather tells every rank whether ANY rank found a row from
        # below that is new to it -- that rank merges the two sorted lists properly, which changes its count, and the counts
        # are gathered once more (rare with the HIP engine: the lower rank's shared-plane rows are this rank's own)
        got = torch.stack(c.all_gather(torch.cat([torch.tensor([k], dtype=torch.int64, device=dev), miss,
                                                  torch.tensor([1 if ready else 0], dtype=torch.int64, device=dev)])))
        self._await(dev)
        got = got.cpu()
        counts, misses = [int(x) for x in got[:, 0]], [int(x) for x in got[:, 1]]
        self._deferred_ok = all(int(x) for x in got[:, 2])
        self._cap_top, self._cap_prev = self._msg_cap(n_top), self._msg_cap(n_from_prev)
        if any(misses):
            if misses[self.rank]:
                merged, r2 = e.unique(torch.cat([uniq, prev_rows], 0).contiguous())
                remap = r2[:nu].to(torch.int64)
                idx_prev = r2[nu:]
                uniq, nu = merged, merged.shape[0]
                k = nu - n_top
            got = torch.stack(c.all_gather(torch.tensor([k], dtype=torch.int64, device=dev))).cpu()
            counts = [int(x) for x in got[:, 0]]
        # the indices of the rows that came from below go back down
        _, ids_next = c.exchange(idx_prev.contiguous(), None, torch.int32, recv_shape_next=(n_top,))
        offs = np.concatenate([[0], np.cumsum(counts)])
        self.vertex_offset, self.n_vertices_global = int(offs[self.rank]), int(offs[-1])
        # own rows by position, shared-plane rows through the upper rank's ids
        gid = torch.arange(nu, dtype=torch.int64, device=dev) + int(offs[self.rank])
        if n_top:
            gid[k:] = ids_next.to(torch.int64) + int(offs[self.rank + 1])
        if remap is not None:
            gid = gid[remap]
        assert gid.shape[0] == nu0
        # for a caller that maps indices on the device (tomo_mc3_faces_slab): the numbers this mapping was made of
        self._numbering = {"counts": counts, "n_top": n_top, "ids_next": ids_next if n_top else None, "merged": remap is not None}
        return uniq[:k], gid

    def _global_numbering(self, vpos, faces32, slice_depths, dev, vkey=None, ny=None):
        e = self.eng
        nv = vpos.shape[0]
        fast = vkey is not None and ny is not None and hasattr(e, "unique_mc")     # rows in marching-cubes order, with keys
        # ALL vertices of this rank: sorted unique rows + the index of every provisional vertex in them
        if nv:
            uniq, rank = e.unique_mc(vpos, vkey, ny, getattr(self, "_mesh_nz", 0)) if fast else e.unique(vpos.contiguous())
        else:
            uniq, rank = vpos, torch.zeros(0, dtype=torch.int32, device=dev)
        kept, gid_rows = self._number_rows(uniq, slice_depths, dev)
        return kept, self._faces(faces32, gid_rows[rank.to(torch.int64)])

    def _global_numbering_mc3(self, m, slice_depths, dev, ready=False):
        """The same for the mc3 chain: m.uniq are this rank's sorted unique rows, m.table maps a vertex id to its row; the
        triangles are written once, straight through the table of GLOBAL indices."""
        if m is None:                                              # no surface in this slab: still take part in the exchanges
            empty = torch.zeros((0, 3), dtype=torch.float32, device=dev)
            kept, _ = self._number_rows(empty, slice_depths, dev)
            return kept, torch.zeros((0, 3), dtype=torch.int64, device=dev)
        nu = m.uniq.shape[0]
        kept, gid_rows = self._number_rows(m.uniq, slice_depths, dev, ready)
        if self.n_vertices_global >= 2 ** 31:
            raise pipeline._lib.TomoError("more than 2^31 vertices: the triangle table holds 32-bit indices")
        info = getattr(self, "_numbering", None)
        if self.world > 1 and info is not None and not info["merged"] and nu == m.nv and getattr(m, "_cap_v", 0) >= m.nv:
            # the triangle kernel maps this rank's rows to GLOBAL indices itself (own rows by position + the lower ranks' kept
            # counts, shared-plane rows through the upper rank's answer), as in the pass without round trips; the counts it
            # takes from `gathered` are the ones the host has just read (was: a table of global indices built with five
            # elementwise torch kernels over 4 entries per list position, 150 us per rank)
            g = np.zeros((self.world, 8), dtype=np.int64)
            g[:, 0] = info["counts"]
            gathered = torch.from_numpy(g).to(dev)
            faces = m.faces(again=True, slab_map=(gathered, self.rank, self.world, info["ids_next"], info["n_top"], m._cap_v))
            return kept, m.faces_checked(faces=faces)
        # rows merged with the lower rank's (a row from below that was new here) or re-sorted by the general unique: through a
        # table of global indices; entries that belong to no vertex hold whatever was in memory: clamp before they index anything
        table_g = gid_rows.to(torch.int32)[m.table.clamp_(0, max(nu - 1, 0))]
        return kept, m.faces_checked(table_g, again=True)

    def _numbering_deferred(self, m, f, Za, slice_depths, mm_y, mm_x, z_top, dev, counters):
        """_global_numbering_mc3 without a host round trip before the triangles are written: the chain ran from size hints
        and has not been read (m.deferred), the shared-plane rows travel in messages of the capacity both neighbours took
        from the last pass (count in the header row), the lookup, the offsets and the triangle kernel (which writes GLOBAL
        indices) take every count from device memory, and ONE download at the end brings this rank's counters and every
        rank's summary.
        Anything that did not fit or is not exact -- on ANY rank: the summaries are all-gathered, so all ranks decide alike --
        sends every rank through the exact pass once more.  Every rank issues the same three collective steps whatever its
        own state (a rank whose chain came back resolved or empty flags that in its summary)."""
        e, c = self.eng, self.comm
        r, w = self.rank, self.world
        first, last = r == 0, r == w - 1
        live = m is not None and getattr(m, "deferred", False)
        tot, gathered = counters[:8], counters[8:].view(w, 8)
        if live:
            assert m._tot.data_ptr() == tot.data_ptr()
            uniq, cap_v = m._uniq, m._cap_v
        else:                                                     # stand-ins: no rows, "overflow" set so that no count is trusted
            tot.copy_(torch.tensor([0, 0, 0, 1, 0, 0, 0, 0], dtype=torch.int64), non_blocking=True)
            uniq, cap_v = torch.zeros((1, 3), dtype=torch.float32, device=dev), 1
        cap_top, cap_prev = (0 if last else self._cap_top), (0 if first else self._cap_prev)
        up = torch.zeros(3, dtype=torch.float32, device=dev) if last else e.slab_top_rows(uniq, tot, cap_v, cap_top)
        from_prev, _ = c.exchange(None, up, torch.float32, recv_shape_prev=((cap_prev + 1) * 3,))
        summary = torch.empty(8, dtype=torch.int64, device=dev)
        idx_prev = e.slab_lookup_summary(uniq, tot, cap_v, None if first else from_prev, cap_prev, cap_top, 0 if live else 1, summary)
        c.all_gather_into(summary, gathered)
        down = torch.zeros(1, dtype=torch.int32, device=dev) if first else idx_prev
        _, ids_next = c.exchange(down, None, torch.int32, recv_shape_next=(cap_top,))
        faces = None
        if live:
            faces = m.faces(slab_map=(gathered, r, w, None if last else ids_next, cap_top, cap_v))
        dl = e.download_start(counters) if hasattr(e, "download_start") else None
        return {"down": dl, "counters": counters, "m": m, "faces": faces, "f": f, "Za": Za, "depths": slice_depths, "mm": (mm_y, mm_x),
                "z_top": z_top, "dev": dev}

    def _numbering_deferred_finish(self, t):
        """The host side of _numbering_deferred: read the counters (possibly one pass late), decide, publish."""
        e = self.eng
        r, w = self.rank, self.world
        m, faces, f, Za, slice_depths, z_top, dev = t["m"], t["faces"], t["f"], t["Za"], t["depths"], t["z_top"], t["dev"]
        mm_y, mm_x = t["mm"]
        if t["down"] is not None:
            host = t["down"].wait(TIMEOUT_S or None) if w > 1 else t["down"].wait()
        else:
            host = e.download(t["counters"])
        own, g = host[:8], [host[8 + 8 * i:16 + 8 * i] for i in range(w)]
        bad = any(row[1] or row[2] for row in g) or any(g[i][4] != g[i + 1][5] for i in range(w - 1))
        if bad:
            # rare: a hint or a message capacity was too small, rows that do not ascend strictly (the general sort decides),
            # a row from below that is new here -- everybody takes the exact pass
            self.deferred_redone += 1
            self.deferred_why = {"summaries": g, "tot": own, "hint": pipeline._MC3_HINT.get(getattr(m, "_hint_key", None)) if m is not None else None,
                                 "caps": (getattr(m, "_cap", None), getattr(m, "_cap_v", None), getattr(m, "_cap_f", None)) if m is not None else None}
            self._deferred_ok = False
            m2 = e.mc3_vertices(f, Za, slice_depths, mm_y, mm_x, z_top=z_top, defer=False)
            ready = m2 is not None and bool(e.mc3_ready(f, Za))
            return self._global_numbering_mc3(m2, slice_depths, dev, ready)
        self.deferred_passes += 1
        counts = [row[0] for row in g]
        offs = np.concatenate([[0], np.cumsum(counts)])
        self.vertex_offset, self.n_vertices_global = int(offs[r]), int(offs[-1])
        if self.n_vertices_global >= 2 ** 31:
            raise pipeline._lib.TomoError("more than 2^31 vertices: the triangle table holds 32-bit indices")
        if own[6]:
            raise pipeline._lib.TomoError("internal error: %d triangle corners reference a missing vertex" % own[6])
        m.na, m.nv, m.nf = own[0], own[1], own[2]
        m.uniq = m._uniq[:m.nv]
        pipeline._MC3_HINT[m._hint_key] = (max(m.na, 1), max(m.nv, 1), max(m.nf, 1))
        self._cap_top, self._cap_prev = self._msg_cap(g[r][4]), self._msg_cap(g[r][5])
        faces = faces[:m.nf]
        if own[5]:
            keep = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])
            faces = faces[keep]
        return m._uniq[:counts[r]], faces

    def _faces(self, faces32, gid):
        if faces32.shape[0] == 0:
            return torch.zeros((0, 3), dtype=torch.int64, device=faces32.device)
        if hasattr(self.eng, "remap_faces") and self.n_vertices_global < 2 ** 31:
            return self.eng.remap_faces(faces32, gid.to(torch.int32))
        f = gid[faces32.to(torch.int64)]
        keep = (f[:, 0] != f[:, 1]) & (f[:, 1] != f[:, 2]) & (f[:, 0] != f[:, 2])
        return f[keep]

    # ------------------------------------------------------------------------- consumers of a finished run()
    # BASELINE configs[3] asks for a "seam-free OBJ export" of the Z-slab job and configs[4] for a cross-check against
    # volume_calculator.py: both work on what every rank already holds, exchange a few numbers, and give exactly what
    # the single-GPU classes (OBJExporter, VolumeCalculator) give on the whole volume / the gathered mesh.
    def _volume(self, which):
        vol = {"smoothed": self.smoothed, "created": self.created}[which]
        if vol is None:
            raise RuntimeError("run() first")
        return vol

    def slice_counts(self, which="smoothed"):
        """np.sum(volume[z]) for every slice of the WHOLE stack (volume_calculator.py:33) -> int64 (gz,) on every rank:
        per-rank popcounts of the owned slices, one all-gather of ceil(gz / world) numbers."""
        vol = self._volume(which)
        c = self.eng.slice_counts(vol)
        if not torch.is_tensor(c):
            c = torch.as_tensor(np.asarray(c, dtype=np.int64))
        return self._gather_per_slice(c)

    def _gather_per_slice(self, mine):
        """One int64 per owned slice on every rank -> the (gz,) array of the whole stack on every rank (one all-gather)."""
        room = max(b - a for a, b in zip(self.cuts, self.cuts[1:]))
        padded = torch.zeros(room, dtype=torch.int64, device=mine.device)
        padded[: mine.shape[0]] = mine
        parts = self.comm.all_gather(padded)
        return np.concatenate([parts[r][: self.cuts[r + 1] - self.cuts[r]].cpu().numpy() for r in range(self.world)])

    def slice_vertex_counts(self, slice_depths):
        """Vertices of the current mesh per slice of the whole stack (a vertex between two planes counts for the lower one) ->
        int64 (gz,) on every rank: what the surface-sized kernels of a pass cost, slice by slice."""
        if self.mesh is None:
            raise RuntimeError("run() first")
        verts = self.mesh[0]
        dev = verts.device
        d = np.asarray(slice_depths, dtype=np.float64)
        z = verts[:, 0].detach().cpu().numpy().astype(np.float64)
        if len(d):
            # z' of padded plane Z is the cumulative depth in front of slice Z - 1 (surface_extractor.py:82-113): invert it
            adj = np.concatenate([[d[0]], d, [d[-1]]])
            cum = np.cumsum(np.concatenate([[0.0], adj]))
            plane = np.searchsorted(np.asarray(cum, np.float32).astype(np.float64), z, side="right") - 1
            sl = np.clip(plane, 0, self.gz - 1)                  # (padded plane Z' = slice Z' - 1 after the -1 shift: already slice units)
        else:
            sl = np.clip(np.floor(z).astype(np.int64), 0, self.gz - 1)
        counts = np.bincount(np.clip(sl - self.z0, 0, self.z1 - self.z0 - 1), minlength=self.z1 - self.z0).astype(np.int64)
        return self._gather_per_slice(torch.from_numpy(counts).to(dev))

    def work_balanced_cuts(self, slice_depths, surface_share=0.25, align=1):
        """Cuts for a NEW job on stacks like the current one: a slice weighs 1 (the volume-sized kernels: pack, smoothing,
        field) plus surface_share / (1 - surface_share) x its vertices relative to the mean (the marching-cubes chain).  A rank
        in the middle of a long body holds more surface per slice than one at its ends: with equal slice counts the 8 ranks of
        BASELINE configs[4] hold 3.0 - 3.6 M vertices each.  align: cuts are rounded to a multiple of this many slices."""
        v = self.slice_vertex_counts(slice_depths).astype(np.float64)
        mean = v.mean() if v.sum() > 0 else 1.0
        k = surface_share / max(1.0 - surface_share, 1e-6)
        cuts = balanced_cuts(1.0 + k * v / mean, self.world, max(self.halo + 1, align))
        if align > 1:
            inner = [int(round(c / align)) * align for c in cuts[1:-1]]
            cand = [0] + inner + [self.gz]
            if all(b - a >= self.halo + 1 for a, b in zip(cand, cand[1:])):
                cuts = cand
        return cuts

    def index_box(self, which="smoothed"):
        """(zmin, zmax, ymin, ymax, xmin, xmax) of the set voxels of the whole stack as np.int64, or None if it is empty
        (np.where + min / max, volume_calculator.py:40-44, 62-72): per-rank boxes, one all-gather of six numbers."""
        vol = self._volume(which)
        b = self.eng.bbox(vol)
        big = np.iinfo(np.int64).max
        mine = [big, -1, big, -1, big, -1] if b is None else [b[0] + self.z0, b[1] + self.z0, b[2], b[3], b[4], b[5]]
        dev = self.eng.bits(vol).device
        parts = np.stack([p.cpu().numpy() for p in self.comm.all_gather(torch.tensor(mine, dtype=torch.int64, device=dev))])
        if parts[:, 1].max() < 0:
            return None
        return tuple(np.int64(parts[:, k].min() if k % 2 == 0 else parts[:, k].max()) for k in range(6))

    def voxel_volume(self, mm_per_pixel_x, mm_per_pixel_y, slice_depths, which="smoothed"):
        """== VolumeCalculator.calculate_voxel_volume_variable_depth(whole volume, ...) (volume_calculator.py:23-35)."""
        from .volume_calculator import volume_from_slice_counts
        if len(slice_depths) == 0:
            return 0.0
        return volume_from_slice_counts(self.slice_counts(which), mm_per_pixel_x, mm_per_pixel_y, slice_depths)

    def bounding_box(self, mm_per_pixel_x, mm_per_pixel_y, slice_depths, which="smoothed"):
        """== VolumeCalculator.calculate_bounding_box_variable_depth(whole volume, ...) (volume_calculator.py:59-94)."""
        from .volume_calculator import box_variable_depth
        return box_variable_depth(self.index_box(which), mm_per_pixel_x, mm_per_pixel_y, slice_depths)

    def mesh_sha256(self, mesh=None):
        """SHA-256 of the WHOLE mesh -- the bytes of the (V, 3) float32 vertex array and of the (F, 3) int64 face array that a
        single-GPU run of the same stack returns, i.e. what the reference's global np.unique numbering gives
        (surface_extractor.py:115-126) and what tests/golden/ellipsoid_hashes*.json hold -- WITHOUT gathering the lists: rank 0
        hashes its rows, the two 112-byte running states travel up the ranks (world - 1 neighbour steps, every rank takes part
        in each), the last rank's digests come back in one all-gather.  `mesh`: this rank's (vertices, faces) of some pass
        (default: the job's current one).  -> (vertices hex, faces hex, n_vertices, n_faces) on every rank."""
        from . import _lib
        L = _lib.lib()
        mesh = self.mesh if mesh is None else mesh
        if mesh is None:
            raise RuntimeError("run() first")
        v = np.ascontiguousarray(mesh[0].detach().cpu().numpy(), dtype=np.float32)
        f = np.ascontiguousarray(mesh[1].detach().cpu().numpy(), dtype=np.int64)
        dev = mesh[0].device
        r, w = self.rank, self.world
        st = np.zeros(224 + 16, dtype=np.uint8)                          # two states + the running (vertices, faces) counts
        _lib.check(L.tomo_host_sha256_init(st[0:].ctypes.data), "tomo_host_sha256_init")
        _lib.check(L.tomo_host_sha256_init(st[112:].ctypes.data), "tomo_host_sha256_init")

        def mine(st):
            _lib.check(L.tomo_host_sha256_update(st[0:].ctypes.data, v.ctypes.data if v.size else None, v.nbytes, 0), "tomo_host_sha256_update")
            _lib.check(L.tomo_host_sha256_update(st[112:].ctypes.data, f.ctypes.data if f.size else None, f.nbytes, 0), "tomo_host_sha256_update")
            st[224:].view(np.int64)[:] += (len(v), len(f))
        if r == 0:
            mine(st)
        for step in range(w - 1):                                        # after step k rank k + 1 holds the state of ranks 0 .. k + 1
            got, _ = self.comm.exchange(None, torch.from_numpy(st.copy()).to(dev), torch.uint8)
            if r == step + 1:
                self._await(dev)
                st = got.cpu().numpy().copy()
                mine(st)
        out = np.zeros(64 + 16, dtype=np.uint8)
        _lib.check(L.tomo_host_sha256_digest(st[0:].ctypes.data, out[0:].ctypes.data), "tomo_host_sha256_digest")
        _lib.check(L.tomo_host_sha256_digest(st[112:].ctypes.data, out[32:].ctypes.data), "tomo_host_sha256_digest")
        out[64:] = st[224:]
        if w > 1:
            parts = self.comm.all_gather(torch.from_numpy(out).to(dev))
            self._await(dev)
            out = parts[w - 1].cpu().numpy()
        nv, nf = (int(x) for x in out[64:].view(np.int64))
        return out[:32].tobytes().hex(), out[32:64].tobytes().hex(), nv, nf

    def export_obj(self, path, nthreads=None):
        """ONE OBJ file of the whole mesh, byte for byte what OBJExporter.export_to_obj writes from the gathered
        (vertices, faces) (obj_exporter.py:17-37) -- without gathering it: every rank formats its run of the vertex list
        and of the face list (global indices) on its host cores, the byte counts are all-gathered, rank 0 lays the file
        out (header, total size) and every rank writes its two blocks at their offsets.  `path` must name the same file
        on every rank (the ranks of one node share the file system).  Returns the file size."""
        import os
        from . import _lib
        if self.mesh is None:
            raise RuntimeError("run() first")
        L = _lib.lib()
        verts, faces = self.mesh
        dev = verts.device
        v = np.ascontiguousarray(verts.detach().cpu().numpy(), dtype=np.float32).reshape(-1, 3)
        f = np.ascontiguousarray(faces.detach().cpu().numpy(), dtype=np.int64).reshape(-1, 3)
        nt = int(nthreads or max(1, min(16, (os.cpu_count() or 1) // max(self.world, 1))))
        blocks = []

        def agree(row, what):
            """All-gather one int64 row per rank whose LAST entry is this rank's error flag; a failure on ANY rank becomes the
            same exception on EVERY rank (nobody is left waiting in the next collective step) and rank 0 removes the file."""
            table = torch.stack(self.comm.all_gather(torch.tensor(row, dtype=torch.int64, device=dev))).cpu().numpy()
            bad = [r for r in range(self.world) if table[r, -1]]
            if bad:
                if self.rank == 0:
                    try:
                        os.unlink(path)
                    except OSError:
                        pass
                raise OSError("export_obj: %s failed on rank(s) %s%s" % (what, bad, (": %r" % (err[0],)) if err else ""))
            return table

        err = []
        try:
            import ctypes
            sizes = []
            try:
                for kind, rows in ((0, v), (2, f)):
                    h, nb = ctypes.c_void_p(), ctypes.c_int64()
                    _lib.check(L.tomo_obj_block_format(kind, rows.ctypes.data if len(rows) else None, len(rows), nt,
                                                       ctypes.byref(h), ctypes.byref(nb)), "tomo_obj_block_format")
                    blocks.append(h)
                    sizes.append(int(nb.value))
            except Exception as e:                                    # noqa: BLE001  (out of memory while formatting, ...)
                err.append(e)
                sizes = (sizes + [0, 0])[:2]
            table = agree(sizes + [len(v), len(f), 1 if err else 0], "formatting")
            header = b"# Tomography reconstruction model\n# %d vertices, %d faces\n\n" % (int(table[:, 2].sum()), int(table[:, 3].sum()))
            v_at = len(header) + int(table[: self.rank, 0].sum())
            sep_at = len(header) + int(table[:, 0].sum())
            f_at = sep_at + 1 + int(table[: self.rank, 1].sum())
            total = sep_at + 1 + int(table[:, 1].sum())
            if self.rank == 0:
                try:
                    with open(path, "wb") as fh:
                        fh.write(header)
                        fh.truncate(total)
                        fh.seek(sep_at)
                        fh.write(b"\n")
                except OSError as e:                                  # an unwritable path, a full disk
                    err.append(e)
            agree([1 if err else 0], "creating the file")             # the file exists before anyone writes into it
            for h, at in zip(blocks, (v_at, f_at)):
                rc = L.tomo_obj_block_pwrite(os.fsencode(path), at, h)
                if rc:
                    err.append(OSError(-rc, os.strerror(-rc), path) if rc < -1 else _lib.TomoError("tomo_obj_block_pwrite"))
                    break
            agree([1 if err else 0], "writing")                       # complete on return, on every rank
            return total
        finally:
            for h in blocks:
                L.tomo_obj_block_free(h)
