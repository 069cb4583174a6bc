"""Ad-hoc fuzz of the unique stage (csrc/mesh.hip: uq3_sortrank_kernel and its fall-backs) against the CPU oracle: thin, wide
stacks whose sort segments hold hundreds to thousands of vertices -- runs of 512 + merge rounds, the two-half path beyond 2 048
entries, the clamped run of a mask in the first slice (two stable passes) and, where a segment is too long for LDS, the library
path; noise (ties between buckets: the general unique decides), flat faces of random size, zero depths, padding on / off.
python tests/fuzz_sort.py [seed] [cases]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import pipeline
from oracle import oracle as O
dev = torch.device("cuda:0")


def run(seed=0, cases=20):
    """-> number of mismatches (0 expected); prints one summary line."""
    rng = np.random.default_rng(seed)
    bad = 0
    c0 = dict(pipeline.COUNTERS)
    for it in range(cases):
        nz = int(rng.integers(2, 7))
        ny = int(rng.integers(40, 900))
        nx = int(rng.integers(16, 420))
        kind = it % 5
        yy, xx = np.mgrid[0:ny, 0:nx]
        if kind == 0:                                   # noise: every bucket full of distinct and equal keys
            v = rng.random((nz, ny, nx)) < 0.15 + 0.7 * rng.random()
        elif kind == 1:                                 # smoothed noise: terraces
            v = O.smooth(rng.random((nz, ny, nx)) < 0.55, 1, True)
        elif kind == 2:                                 # a flat-faced body of random extent, somewhere in the stack (cut faces: long buckets)
            v = np.zeros((nz, ny, nx), bool)
            ry, rx = rng.uniform(0.05, 0.5) * ny, rng.uniform(0.05, 0.5) * nx
            disc = ((yy - ny * rng.uniform(0.3, 0.7)) / ry) ** 2 + ((xx - nx * rng.uniform(0.3, 0.7)) / rx) ** 2 <= 1.0
            a = int(rng.integers(0, nz)); b = int(rng.integers(a + 1, nz + 1))
            v[a:b] = disc
            v ^= rng.random(v.shape) < 0.002
        elif kind == 3:                                 # a mask in the first slice: the clamped run (bucket of slice 0 + plane of slice 1)
            v = np.zeros((nz, ny, nx), bool)
            ry, rx = rng.uniform(0.02, 0.45) * ny, rng.uniform(0.02, 0.2) * nx
            v[0:int(rng.integers(1, nz + 1))] = ((yy - ny / 2) / ry) ** 2 + ((xx - nx / 2) / rx) ** 2 <= 1.0
            v ^= rng.random(v.shape) < 0.001
        else:                                           # blobs
            zz = np.arange(nz)[:, None, None]
            v = ((zz - nz / 2) / (nz * 0.6)) ** 2 + ((yy[None] - ny / 2) / (ny * 0.45)) ** 2 + ((xx[None] - nx / 2) / (nx * 0.4)) ** 2 <= 1
            v = v ^ (rng.random(v.shape) < 0.01)
        depths = rng.random(nz) * 0.9 + 0.1
        if it % 7 == 3:
            depths[int(rng.integers(0, nz))] = 0.0      # a slice without depth: rows of two buckets coincide
        my, mx = float(rng.random() + 0.5), float(rng.random() + 0.5)
        pad = bool(it % 4)
        ref = O.SurfaceExtractor().extract_manifold_surface(v, depths, my, mx, True, True, pad)
        mask = torch.from_numpy(np.ascontiguousarray(v).view(np.uint8)).to(dev)
        for rep in range(2):                            # first pass (exact sizes), then from the size hints
            got = pipeline.extract_surface(pipeline.pack(mask), depths, my, mx, True, pad)
            if ref is None or got is None:
                ok = (ref is None) == (got is None)
            else:
                gv, gf = got[0].cpu().numpy(), got[1].cpu().numpy()
                ok = gv.shape == ref[0].shape and bool(np.array_equal(gv.view(np.int32), np.ascontiguousarray(ref[0]).view(np.int32)))
                ok &= gf.shape == np.asarray(ref[1]).shape and bool(np.array_equal(gf, ref[1]))
            if not ok:
                bad += 1
                print("MISMATCH", (nz, ny, nx), "kind", kind, "pad", pad, "rep", rep, "seed", seed, "case", it)
    d = {k: pipeline.COUNTERS.get(k, 0) - c0.get(k, 0) for k in ("mc3_sort_fused", "mc3_sort_library", "mc3_general_unique", "mc3_exact", "mc3_hint_miss")}
    print("sort fuzz: %d cases x 2 passes, %d mismatches; paths %s" % (cases, bad, d))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 20) else 0)
