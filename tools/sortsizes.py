#!/usr/bin/env python3
"""Sizes of the sort segments of a pass (slice_tab offsets) and the time of the sort stage.  usage: sortsizes.py gz ny nx z0 z1"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import _lib, pipeline
gz, ny, nx, z0, z1 = [int(x) for x in sys.argv[1:6]]
dev = torch.device("cuda:0")
L = _lib.lib()
mask = pipeline.ellipsoid_mask(gz, ny, nx, dev, z0, z1).view(torch.uint8)
vol = pipeline.smooth(pipeline.pack(mask), 3, True)          # no close-ends: the sub-stack's end planes stay open
f = pipeline.make_field(vol, True, True)
depths = np.full(z1 - z0, 1.0)
for it in range(3):
    m = pipeline.mc3_vertices(f, depths, 1.0, 1.0, True, with_faces=False)
torch.cuda.synchronize()
nseg = int(L.tomo_mc3_sort_segments(f.Nz, f.Ny))
tab = m._slice_tab.cpu().numpy().view(np.uint32)
off = tab[2 * (f.Nz + 1): 2 * (f.Nz + 1) + nseg + 1].astype(np.int64)
sz = np.diff(off)
nb1 = nseg // f.Nz
print("Nz %d Ny %d: %d segments (%d per slice), %d vertices; sizes: max %d, mean %.0f, > 4096: %d, > 3000: %d, > 2048: %d, == 0: %d"
      % (f.Nz, f.Ny, nseg, nb1, off[-1], sz.max(), sz.mean(), (sz > 4096).sum(), (sz > 3000).sum(), (sz > 2048).sum(), (sz == 0).sum()))
per = sz.reshape(f.Nz, nb1)
print("per-slice segment sizes, middle slice:", per[f.Nz // 2].tolist(), " max per column:", per.max(axis=0).tolist())
