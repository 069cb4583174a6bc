#!/usr/bin/env python3
"""Per-rank cost of a Z-slab pass WITH RCCL in the loop, on a one-GPU box.

RCCL refuses two ranks on one device, so this is a rehearsal, not a measurement of a multi-GPU run: ONE process plays a
middle rank (rank 1 of a pretend world of 3) and every message it would send to a neighbour goes to ITSELF through a
real world-size-1 `nccl` process group -- the same batched isend / irecv of uint8 views and the same all-gather the job
uses, enqueued on the same streams.  What arrives "from below" is what was sent "up" (and vice versa); the edge slices of
the pass's front are then overwritten with what the real neighbours would have sent (generated here: the stack is synthetic), so
the kernels see a seam-free stack -- with the rank's own slices as halo the seam put flat caps of 10^5 vertices into single sort
buckets and the segmented sort took 0.9 ms instead of 0.1 (round 3).  Sizes, kernels, launches, host round trips and the RCCL
calls are those of a middle rank; the mesh is not checked.  What is missing is the time on the xGMI link (3 MB per pass).
usage: slab_selfloop_bench.py [slices_per_rank] [ny] [nx] [steps] [world=3] [rank=world//2]
(BASELINE configs[4] per rank: 512 2048 2048 10 8 3 -- slab 3 of the 8 slabs of the 2048 x 2048 x 4096 stack)"""
import datetime
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as td

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import pipeline, slab  # noqa: E402


class SelfLoopComm(slab.TorchDistComm):
    def __init__(self, device, rank=1, world=3):
        super().__init__(device)
        self.rank, self.world = rank, world

    def exchange(self, to_prev, to_next, dtype, recv_shape_prev=None, recv_shape_next=None, defer=False):
        t_in = time.perf_counter()
        ops, keep = [], []
        from_prev = from_next = None
        for send, shape in ((to_next, recv_shape_prev), (to_prev, recv_shape_next)):
            if send is None:
                continue
            got = torch.empty(tuple(shape) if shape is not None else tuple(send.shape), dtype=dtype, device=self.device)
            if send is to_next:
                from_prev = got
            else:
                from_next = got
            if send.numel() and got.numel():
                assert send.numel() * send.element_size() == got.numel() * got.element_size(), "self loop: sizes differ"
                keep.append(send.contiguous())
                ops.append(td.P2POp(td.isend, self._bytes(keep[-1]), 0))
                ops.append(td.P2POp(td.irecv, self._bytes(got), 0))
        works = td.batch_isend_irecv(ops) if ops else []
        if not defer:
            for q in works:
                q.wait()
        self.stats["bytes_sent"] += sum(k.numel() * k.element_size() for k in keep)
        self.stats["calls"] += 1
        self.stats["seconds"] += time.perf_counter() - t_in
        if not defer:
            return from_prev, from_next
        return from_prev, from_next, (lambda works=works, keep=keep: [q.wait() for q in works])

    def all_gather_into(self, src, out):
        out.copy_(torch.stack([g.reshape(-1) for g in self.all_gather(src)]).view_as(out))
        return out

    def all_gather(self, t):
        t_in = time.perf_counter()
        out = [torch.empty_like(t)]
        td.all_gather(out, t.contiguous())
        self.stats["bytes_sent"] += t.numel() * t.element_size()
        self.stats["calls"] += 1
        self.stats["seconds"] += time.perf_counter() - t_in
        return out * self.world


class SelfLoopRccl:
    """The same play on rccl.RcclComm (RCCL's C API on the compute stream): logical rank 1 of 3, every peer is rank 0 of the
    1-rank communicator."""

    def __new__(cls, device, rank=1, world=3):
        from tomography_3d_reconstructor_amd import rccl

        class _Loop(rccl.RcclComm):
            def _peer(self, r):
                return 0

            def exchange(self, to_prev, to_next, dtype, recv_shape_prev=None, recv_shape_next=None):
                got = rccl.RcclComm.exchange(self, to_prev, to_next, dtype, recv_shape_prev, recv_shape_next)
                # the edge slices of the pass's front: what arrived is this rank's own (sent to itself) -- a seam, with flat caps
                # of 10^5 vertices in the halo that a real neighbour would not produce.  Overwritten (two small device copies)
                # with what the real neighbours WOULD have sent, so that the kernels behind see a seam-free stack.
                truth = getattr(self, "truth", None)
                if truth is not None and dtype == torch.int64 and got[0] is not None and got[1] is not None \
                        and got[0].shape == truth[0].shape and got[1].shape == truth[1].shape:
                    got[0].copy_(truth[0])
                    got[1].copy_(truth[1])
                return got

            def all_gather(self, t):
                return [rccl.RcclComm.all_gather(self, t)[0]] * self.world

            def all_gather_into(self, src, out):         # the real 1-rank all-gather fills one row; the pretend neighbours' rows are copies
                w, self.world = self.world, 1
                try:
                    rccl.RcclComm.all_gather_into(self, src, out[:1])
                finally:
                    self.world = w
                out[1:].copy_(out[:1].expand(w - 1, -1))
                return out

        c = _Loop(device)
        c._real_world = c.world
        c.rank, c.world = rank, world
        return c


def main():
    nzr = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    ny = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    nx = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    world = int(sys.argv[5]) if len(sys.argv) > 5 else 3          # the pretend job: `world` slabs of nzr slices, this process plays
    rank = int(sys.argv[6]) if len(sys.argv) > 6 else world // 2  # slab `rank` (a middle one)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29618")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    td.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
    direct = os.environ.get("TOMO_RCCL_DIRECT", "1") not in ("", "0")
    comm = SelfLoopRccl(dev, rank, world) if direct else SelfLoopComm(dev, rank, world)
    gz = world * nzr
    cuts = os.environ.get("TOMO_SELFLOOP_CUTS")              # "0,606,...,4096": slabs of equal work (tools/balanced_cuts.py prints them)
    cuts = [int(c) for c in cuts.split(",")] if cuts else None
    job = slab.SlabJob(gz, ny, nx, comm, z_cuts=cuts)
    mask = pipeline.ellipsoid_mask(gz, ny, nx, dev, job.z0, job.z1).view(torch.uint8)
    depths = np.full(gz, 1.0)
    if direct and 0 < rank < world - 1:
        H = job.halo                       # original edge slices the neighbours send: H + 1 from below, H + 2 from above (slab.py)
        lo = pipeline.pack(pipeline.ellipsoid_mask(gz, ny, nx, dev, job.z0 - (H + 1), job.z0).view(torch.uint8)).bits
        hi = pipeline.pack(pipeline.ellipsoid_mask(gz, ny, nx, dev, job.z1, job.z1 + H + 2).view(torch.uint8)).bits
        comm.truth = (lo, hi)
    res = None
    for _ in range(4):
        res = job.run(mask, depths, 1.0, 1.0)
    comm.reset_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    serial = os.environ.get("TOMO_READ_EVERY_PASS", "0") not in ("", "0")
    pend = None
    for _ in range(steps):
        h0 = time.perf_counter()
        if serial:
            res = job.run(mask, depths, 1.0, 1.0)
        else:                                   # pass k + 1 is enqueued before the host reads the counters of pass k
            nxt = job.submit(mask, depths, 1.0, 1.0)
            if pend is not None:
                res = job.result(pend)
            pend = nxt
        host += time.perf_counter() - h0
    if pend is not None:
        res = job.result(pend)
    torch.cuda.synchronize()
    slab_ms = (time.perf_counter() - t0) / steps * 1e3
    st = dict(comm.stats)
    nv, nf = int(res[0].shape[0]), int(res[1].shape[0])
    if os.environ.get("DUMP_SEGS"):                         # sizes of the sort's segments of one more pass of this rank
        from tomography_3d_reconstructor_amd import _lib
        keep = {}
        orig = job._numbering_deferred_finish
        def spy(t):
            keep["m"] = t["m"]
            return orig(t)
        job._numbering_deferred_finish = spy
        job.run(mask, depths, 1.0, 1.0)
        m = keep["m"]; f = m._f
        nseg = int(_lib.lib().tomo_mc3_sort_segments(f.Nz, f.Ny))
        tab = m._slice_tab.cpu().numpy().view(np.uint32)
        off = tab[2 * (f.Nz + 1): 2 * (f.Nz + 1) + nseg + 1].astype(np.int64)
        sz = np.diff(off)
        per = sz.reshape(f.Nz, nseg // f.Nz)
        print("segments: Nz %d Ny %d, %d segments, sizes max %d mean %.0f, > 4096: %d; monotone: %s; cap_v %d nv %d; per-slice max per column %s; slice 0 %s, slice -1 %s"
              % (f.Nz, f.Ny, nseg, sz.max(), sz.mean(), (sz > 4096).sum(), bool((sz >= 0).all()), m._cap_v, m.nv, per.max(axis=0).tolist(),
                 per[0].tolist(), per[-1].tolist()), flush=True)
    del res
    # host time of submit() alone: enqueue 4 passes back to back without reading any, then collect them
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    tickets = [job.submit(mask, depths, 1.0, 1.0) for _ in range(4)]
    enq_ms = (time.perf_counter() - h0) / 4 * 1e3
    for t in tickets:
        job.result(t)
    torch.cuda.synchronize()
    del tickets
    # the single-GPU pass on a stack of the slab's size, same box
    m1 = pipeline.ellipsoid_mask(nzr, ny, nx, dev).view(torch.uint8)
    d1 = np.full(nzr, 1.0)
    pend = None
    for it in range(steps + 3):
        if it == 3:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        nxt = pipeline.extract_surface_submit(pipeline.smooth(pipeline.pack_closed(m1), 3, True), d1, 1.0, 1.0)
        if serial:
            nxt.result()
        else:
            if pend is not None:
                pend.result()
            pend = nxt
    if pend is not None:
        pend.result()
    torch.cuda.synchronize()
    one_ms = (time.perf_counter() - t0) / steps * 1e3
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    pends = [pipeline.extract_surface_submit(pipeline.smooth(pipeline.pack_closed(m1), 3, True), d1, 1.0, 1.0) for _ in range(4)]
    enq1_ms = (time.perf_counter() - h0) / 4 * 1e3
    for q in pends:
        q.result()
    del pends
    how = "RCCL communicator (C API, compute stream)" if direct else "nccl group (torch.distributed)"
    print(("rank %d of %d x (%d, %d, %d), messages to self over a 1-rank " + how + ": %.3f ms per pass "
           "(%d deferred, %d redone; %.1f RCCL calls and %.0f KB sent per pass, %.3f ms of host time inside them, "
           "%.3f ms of host time per pass in run()); %d vertices, %d faces kept; single-GPU pass on a slab-sized ellipsoid "
           "stack: %.3f ms; host time to enqueue a pass: slab %.3f ms, single %.3f ms") % (comm.rank, comm.world, nzr, ny, nx, slab_ms, job.deferred_passes, job.deferred_redone, st["calls"] / steps,
                                st["bytes_sent"] / steps / 1e3, st["seconds"] / steps * 1e3, host / steps * 1e3, nv, nf, one_ms, enq_ms, enq1_ms),
          flush=True)
    if hasattr(comm, "close"):
        comm.close()
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
