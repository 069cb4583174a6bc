#!/usr/bin/env python3
"""Z cuts of equal WORK for the ellipsoid stack (nz, ny, nx) and `world` ranks, from the per-slice vertex counts of ONE single-GPU
pass (what bench.py's ranks all-gather after their first pass): prints the cuts, the vertices per rank with equal slice counts and
with the cuts.  usage: balanced_cuts.py NZ NY NX WORLD [surface_share]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import pipeline, slab
nz, ny, nx, world = (int(a) for a in sys.argv[1:5])
share = float(sys.argv[5]) if len(sys.argv) > 5 else min(0.5, 0.29 * 1024.0 / (ny * nx) ** 0.5)
dev = torch.device("cuda:0")
mask = pipeline.ellipsoid_mask(nz, ny, nx, dev).view(torch.uint8)
v, f = pipeline.extract_surface(pipeline.smooth(pipeline.pack_closed(mask), 3, True), np.full(nz, 1.0), 1.0, 1.0)
z = v[:, 0].cpu().numpy()
counts = np.bincount(np.clip(np.floor(z).astype(np.int64), 0, nz - 1), minlength=nz).astype(np.float64)
k = share / (1 - share)
cuts = slab.balanced_cuts(1.0 + k * counts / counts.mean(), world, 11)
cuts = [0] + [int(round(c / 4)) * 4 for c in cuts[1:-1]] + [nz]
eq = [slab.slab_range(nz, r, world) for r in range(world)]
print("surface share assumed %.3f" % share)
print("equal slices : slices", [b - a for a, b in eq], "vertices (k)", [int(counts[a:b].sum() / 1e3) for a, b in eq])
print("equal work   : slices", [b - a for a, b in zip(cuts, cuts[1:])], "vertices (k)", [int(counts[a:b].sum() / 1e3) for a, b in zip(cuts, cuts[1:])])
print("TOMO_SELFLOOP_CUTS=" + ",".join(str(c) for c in cuts))
