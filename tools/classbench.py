#!/usr/bin/env python3
"""Host-boundary timing of the drop-in classes (host arrays in, host arrays out: PCIe inclusive) on the N^3 ellipsoid;
never the bench `value` (that is quoted with inputs resident in HBM).  python tools/classbench.py [N]"""
import contextlib, io, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import SurfaceExtractor, VoxelProcessor, pipeline  # noqa: E402
from tomography_3d_reconstructor_amd.volume_calculator import VolumeCalculator  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
masks = [np.ascontiguousarray(m) for m in pipeline.ellipsoid_mask(n, n, n, torch.device('cuda:0')).cpu().numpy()]   # SURVEY 8(d) generator
vp, se, vc = VoxelProcessor(), SurfaceExtractor(), VolumeCalculator()
for rep in range(3):
    t = [time.perf_counter()]
    with contextlib.redirect_stdout(io.StringIO()):
        vol = vp.create_voxel_data(masks, True, 0, n, 0); t.append(time.perf_counter())
        depths = vp.calculate_slice_depths(float(n))
        sm = vp.smooth_voxel_data(vol, iterations=3, create_manifold=True); t.append(time.perf_counter())
        v, f = se.extract_manifold_surface(sm, depths, 1.0, 1.0); t.append(time.perf_counter())
        vv = vc.calculate_voxel_volume_variable_depth(sm, 1.0, 1.0, depths); t.append(time.perf_counter())
    d = np.diff(t)
    print("rep %d: create %.1f ms | smooth %.1f ms | extract %.1f ms | voxel volume %.1f ms | total %.1f ms = %.0f Mvoxels/s (V %d F %d)"
          % (rep, d[0] * 1e3, d[1] * 1e3, d[2] * 1e3, d[3] * 1e3, (t[3] - t[0]) * 1e3, n ** 3 / (t[3] - t[0]) / 1e6, len(v), len(f)), flush=True)
