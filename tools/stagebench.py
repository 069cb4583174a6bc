#!/usr/bin/env python3
"""Per-stage wall time (HIP events) of one pass of the hot path on an ellipsoid: python tools/stagebench.py [N | NZ NY NX]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import pipeline  # noqa: E402

if len(sys.argv) > 3:
    nz, ny, nx = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
else:
    nz = ny = nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
mask = pipeline.ellipsoid_mask(nz, ny, nx, dev).view(torch.uint8)
depths = np.full(nz, 1.0)
stages = ["pack+close", "smooth", "field", "mc3 chain"] if pipeline.MC3 else ["pack", "close", "smooth", "field", "mc", "finalize", "unique"]
acc = {s: [] for s in stages}
for it in range(6):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(stages) + 1)]
    ev[0].record()
    if pipeline.MC3:                                       # the production chain (TOMO_MC_PATH=old times the round-1 stages)
        vol = pipeline.pack_closed(mask); ev[1].record()
        vol = pipeline.smooth(vol, 3, True); ev[2].record()
        f = pipeline.make_field(vol, True, True); ev[3].record()
        m = pipeline.mc3_vertices(f, depths, 1.0, 1.0, True); ev[4].record()
        torch.cuda.synchronize()
        if it >= 2:
            for k, s in enumerate(stages):
                acc[s].append(ev[k].elapsed_time(ev[k + 1]))
        del f, m, vol
        continue
    vol = pipeline.pack(mask); ev[1].record()
    vol = pipeline.close_ends(vol, inplace=True); ev[2].record()
    vol = pipeline.smooth(vol, 3, True); ev[3].record()
    f = pipeline.make_field(vol, True, True); ev[4].record()
    mesh = pipeline.marching_cubes(f, 0.5); ev[5].record()
    mesh._mc = None
    pipeline.finalize_vertices(mesh.vpos, depths, 1.0, 1.0, True, True); ev[6].record()
    v, fa = pipeline.ensure_manifold_mesh(mesh); ev[7].record()
    torch.cuda.synchronize()
    if it >= 2:
        for k, s in enumerate(stages):
            acc[s].append(ev[k].elapsed_time(ev[k + 1]))
    del f, mesh, v, fa, vol
print("unique path:", pipeline.COUNTERS)
print(" | ".join("%s %.3f" % (s, float(np.mean(acc[s]))) for s in stages), "| total %.3f ms" % sum(float(np.mean(acc[s])) for s in stages))
