#!/usr/bin/env python3
"""Time tomo_fill_holes_slice on NON-EMPTY end slices (the bench ellipsoid's end slices are empty, so its trace only shows
the early-out): the filled ellipse of a half-ellipsoid stack, a ring, noise, a square spiral -- at 1024^2 and 2048^2, checked
against scipy.ndimage.binary_fill_holes (the call the reference makes, voxel_processor.py:60-68).
    python tools/fillholestime.py [out.md]"""
import os, sys
import numpy as np
import torch
from scipy import ndimage
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import _lib, pipeline  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()


def shapes(n):
    yy, xx = np.mgrid[0:n, 0:n]
    c = (n - 1) / 2.0
    r2 = ((xx - c) / (0.42 * n)) ** 2 + ((yy - c) / (0.40 * n)) ** 2
    out = {"ellipse (half-ellipsoid end slice)": r2 <= 1.0,
           "ring (one big hole)": (r2 <= 1.0) & (r2 >= 0.5),
           "ellipse with 2000 pinholes": (r2 <= 1.0) & (np.random.default_rng(1).random((n, n)) > 2000.0 / n / n),
           "noise 50 %": np.random.default_rng(2).random((n, n)) < 0.5}
    sp = np.zeros((n, n), bool)                      # square spiral wall, 8 px pitch: background is one long corridor
    k = 0
    lo, hi = 4, n - 5
    while hi - lo > 16:
        sp[lo, lo:hi + 1] = True; sp[lo:hi + 1, hi] = True; sp[hi, lo + 8:hi + 1] = True; sp[lo + 8:hi + 1, lo + 8] = True
        sp[lo + 8, lo + 8:hi - 7] = True
        lo += 16; hi -= 16; k += 1
    out["spiral corridor (%d turns)" % k] = sp
    return out


rows = []
for n in (1024, 2048):
    for name, sl in shapes(n).items():
        vol = np.zeros((3, n, n), bool)
        vol[0] = sl
        bv = pipeline.pack(torch.from_numpy(vol.view(np.uint8)).to(dev))
        scratch = torch.empty(n * bv.bits.shape[2] + 8, dtype=torch.int64, device=dev)
        ref = ndimage.binary_fill_holes(sl)
        times = []
        for rep in range(5):
            work = pipeline.BitVolume(bv.bits.clone(), bv.shape)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(L.tomo_fill_holes_slice(work.bits.data_ptr(), 3, n, n, 0, scratch.data_ptr(), torch.cuda.current_stream().cuda_stream), "fill")
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) * 1e3)
        got = pipeline.unpack(work).cpu().numpy()[0]
        ok = np.array_equal(got, ref)
        rows.append("| %d^2 | %s | %.1f | %s |" % (n, name, min(times), "== scipy" if ok else "MISMATCH"))
        print(rows[-1], flush=True)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        f.write("| slice | content | tomo_fill_holes_slice, us (best of 5, HIP events) | result |\n|---|---|---|---|\n" + "\n".join(rows) + "\n")
