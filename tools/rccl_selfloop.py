#!/usr/bin/env python3
"""What of the RCCL transport CAN run on a one-GPU box: a world-size-1 `nccl` process group.  all_gather, all_reduce,
barrier and a batched isend/irecv of uint8 VIEWS to self go through librccl (RCCL refuses two ranks on one device, so
the neighbour exchange between distinct ranks first runs on a multi-GPU node).  Also runs SlabJob on that group."""
import datetime, os, sys
import numpy as np
import torch
import torch.distributed as td
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import pipeline, slab  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29617")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
td.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
comm = slab.TorchDistComm(dev)
print("preflight:", slab.preflight(comm), flush=True)
# byte-view send/recv to self, batched like TorchDistComm.exchange does
a = torch.arange(4096, dtype=torch.int64, device=dev)
b = torch.empty_like(a)
ops = [td.P2POp(td.isend, a.view(torch.uint8).reshape(-1), 0), td.P2POp(td.irecv, b.view(torch.uint8).reshape(-1), 0)]
for q in td.batch_isend_irecv(ops):
    q.wait()
torch.cuda.synchronize()
assert torch.equal(a, b), "self send/recv of a uint8 view"
f = torch.rand(1026 * 1056, device=dev)
g = torch.empty_like(f)
for q in td.batch_isend_irecv([td.P2POp(td.isend, f.view(torch.uint8).reshape(-1), 0), td.P2POp(td.irecv, g.view(torch.uint8).reshape(-1), 0)]):
    q.wait()
torch.cuda.synchronize()
assert torch.equal(f, g)
print("self-loop batch_isend_irecv on uint8 views: ok", flush=True)
nz, ny, nx = 256, 256, 256
job = slab.SlabJob(nz, ny, nx, comm)
mask = pipeline.ellipsoid_mask(nz, ny, nx, dev).view(torch.uint8)
v, fc = job.run(mask, np.full(nz, 1.0), 1.0, 1.0)
v1, f1 = pipeline.extract_surface(pipeline.smooth(pipeline.close_ends(pipeline.pack(mask)), 3, True), np.full(nz, 1.0), 1.0, 1.0)
assert torch.equal(v, v1) and torch.equal(fc, f1)
print("SlabJob over a 1-rank nccl group == single-GPU path: %d vertices, %d faces" % (v.shape[0], fc.shape[0]), flush=True)
# the same through RCCL's C API on the compute stream (rccl.RcclComm, what bench.py uses over nccl): a rank that plays the
# middle of three and whose neighbours are itself -- what it sends up must arrive as what comes from below, and vice versa
from tomography_3d_reconstructor_amd import rccl  # noqa: E402


class Loop(rccl.RcclComm):
    def _peer(self, r):
        return 0


direct = Loop(dev)
assert (direct.rank, direct.world) == (0, 1)
print("preflight (direct):", slab.preflight(direct), flush=True)
direct.rank, direct.world = 1, 3
up = torch.arange(3 * 1024 * 16, dtype=torch.int64, device=dev).reshape(3, 1024, 16)
down = -torch.arange(5 * 1024 * 16, dtype=torch.int64, device=dev).reshape(5, 1024, 16)
from_prev, from_next = direct.exchange(down, up, torch.int64)
assert torch.equal(from_prev, up) and torch.equal(from_next, down), "self loop over ncclSend / ncclRecv"
rows = torch.rand((77, 3), device=dev)
got, none = direct.exchange(None, rows, torch.float32, recv_shape_prev=(77, 3))
assert none is None and torch.equal(got, rows)
none, ids = direct.exchange(torch.arange(9, dtype=torch.int32, device=dev), None, torch.int32, recv_shape_next=(9,))
assert none is None and torch.equal(ids, torch.arange(9, dtype=torch.int32, device=dev))
a2, b2, fin = direct.exchange_async(down, up, torch.int64)
fin()
assert torch.equal(a2, up) and torch.equal(b2, down)
direct.rank, direct.world = 0, 1
g = direct.all_gather(torch.tensor([5, 6, 7], dtype=torch.int64, device=dev))
assert len(g) == 1 and g[0].tolist() == [5, 6, 7]
job = slab.SlabJob(nz, ny, nx, direct)
v2, f2 = job.run(mask, np.full(nz, 1.0), 1.0, 1.0)
assert torch.equal(v2, v1) and torch.equal(f2, f1)
print("RcclComm (C API, compute stream): self-loop exchange, all-gather and a 1-rank SlabJob ok; calls %d, bytes %d"
      % (direct.stats["calls"], direct.stats["bytes_sent"]), flush=True)
direct.close()
td.barrier()
td.destroy_process_group()
print("rccl selfloop ok")
