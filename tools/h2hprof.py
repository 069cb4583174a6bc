#!/usr/bin/env python3
"""Where does a host-to-host run of the three class methods spend its time?  (1024^3 ellipsoid, three runs in one process)"""
import contextlib, io, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import SurfaceExtractor, VoxelProcessor, pipeline, voxel_processor as VP, _hostbuf
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
stack = pipeline.ellipsoid_mask(n, n, n, dev).cpu().numpy()
masks = [stack[i].copy() for i in range(n)]
del stack
acc = {}
def wrap(mod, name, sync=True):
    orig = getattr(mod, name)
    def f(*a, **k):
        if sync: torch.cuda.synchronize()
        t0 = time.perf_counter(); r = orig(*a, **k)
        if sync: torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return r
    setattr(mod, name, f)
for nm in ("_stage_masks", "to_host_array", "to_device_volume", "upload_volume"):
    wrap(VP, nm)
for nm in ("pack_closed", "smooth", "extract_surface", "unpack"):
    wrap(pipeline, nm)
wrap(_hostbuf, "take", sync=False)
import tomography_3d_reconstructor_amd.surface_extractor as SE
SE.to_host_array = VP.to_host_array
if os.environ.get("NO_RESERVE"):
    _hostbuf.reserve = lambda *a, **k: None
if os.environ.get("INNER"):
    import concurrent.futures as cf
    orig_copy = torch.Tensor.copy_
    tacc = {"copy": 0.0, "n": 0}
    def timed_copy(self, src, *a, **k):
        t0 = time.perf_counter(); r = orig_copy(self, src, *a, **k); tacc["copy"] += (time.perf_counter() - t0) * 1e3; tacc["n"] += 1; return r
    torch.Tensor.copy_ = timed_copy
for run in range(3):
    acc.clear()
    vp, se = VoxelProcessor(), SurfaceExtractor()
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        vol = vp.create_voxel_data(masks, True, 0, n, 0); t1 = time.perf_counter()
        depths = vp.calculate_slice_depths(float(n))
        sm = vp.smooth_voxel_data(vol, iterations=3, create_manifold=True); t2 = time.perf_counter()
        res = se.extract_manifold_surface(sm, depths, 1.0, 1.0, smooth=True, manifold=True, add_padding=True); t3 = time.perf_counter()
    print("run %d: total %.1f ms = create %.1f + smooth %.1f + extract %.1f | %s" % (
        run, (t3 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, ", ".join("%s %.1f" % kv for kv in sorted(acc.items()))), flush=True)
    if os.environ.get("INNER"):
        print("   copy_ calls: %d, %.1f ms in total" % (tacc["n"], tacc["copy"])); tacc["copy"] = 0.0; tacc["n"] = 0
    t0 = time.perf_counter(); del vol, sm, res; print("   (freeing the results: %.1f ms)" % ((time.perf_counter() - t0) * 1e3))
