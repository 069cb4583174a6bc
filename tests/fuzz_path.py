"""Ad-hoc fuzz of the whole hot path against the CPU oracle on random small volumes (close ends, smoothing, field, marching
cubes, slice depths, unique / remap), with the size hints and -- second half of the cases -- the sparse field switched on."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run as: python tests/fuzz_path.py [seed] [cases]
from tomography_3d_reconstructor_amd import pipeline
from oracle import oracle as O
dev = torch.device("cuda:0")


def run(seed=0, cases=30):
    """-> number of mismatches (0 expected); prints one summary line."""
    rng = np.random.default_rng(seed)
    n = cases
    bad = 0
    shapes = [(int(rng.integers(2, 30)), int(rng.integers(2, 70)), int(rng.integers(2, 300))) for _ in range(6)]
    for it in range(n):
        nz, ny, nx = shapes[it % len(shapes)] if it % 3 else (int(rng.integers(1, 30)), int(rng.integers(1, 70)), int(rng.integers(1, 300)))
        if it % 4 == 1:
            nx = max(16, nx // 16 * 16)                      # layouts the fused pack + close kernel takes
        kind = it % 3
        if kind == 0:
            v = rng.random((nz, ny, nx)) < 0.3 + 0.5 * rng.random()
        elif kind == 1:
            zz, yy, xx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
            v = ((zz - nz / 2) / (nz * 0.45 + 1)) ** 2 + ((yy - ny / 2) / (ny * 0.4 + 1)) ** 2 + ((xx - nx / 2) / (nx * 0.42 + 1)) ** 2 <= 1
            v ^= rng.random((nz, ny, nx)) < 0.02
        else:
            v = rng.random((nz, ny, nx)) < 0.6
            v = O.smooth(v, 1, True)
        iters, cm, ce = int(rng.integers(0, 4)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        depths = rng.random(nz) * 0.9 + 0.1
        my, mx = float(rng.random() + 0.5), float(rng.random() + 0.5)
        pipeline.FIELD_SPARSE = it >= n // 2
        # oracle
        w = v.copy()
        if ce:
            w = O.close_ends(w)
        w = O.smooth(w, iters, cm)
        pad = bool(it % 5)
        ref = O.SurfaceExtractor().extract_manifold_surface(w, depths, my, mx, True, True, pad)
        # device
        mask = torch.from_numpy(np.ascontiguousarray(v).view(np.uint8)).to(dev)
        vol = pipeline.pack_closed(mask) if ce else pipeline.pack(mask)      # one pass over the mask where the layout allows
        vol = pipeline.smooth(vol, iters, cm)
        ok = bool(np.array_equal(pipeline.unpack(vol).cpu().numpy().astype(bool), w))
        got = pipeline.extract_surface(vol, depths, my, mx, True, pad)
        if ref is None or got is None:
            ok &= (ref is None) == (got is None)
        else:
            gv, gf = got[0].cpu().numpy(), got[1].cpu().numpy()
            ok &= gv.shape == ref[0].shape and bool(np.array_equal(gv.view(np.int32), np.ascontiguousarray(ref[0]).view(np.int32)))
            ok &= gf.shape == np.asarray(ref[1]).shape and bool(np.array_equal(gf, ref[1]))
        if not ok:
            bad += 1
            print("MISMATCH", (nz, ny, nx), "kind", kind, "iters", iters, cm, ce, "sparse", pipeline.FIELD_SPARSE)
    print("path fuzz: %d cases, %d mismatches; counters %s" % (n, bad, pipeline.COUNTERS))
    pipeline.FIELD_SPARSE = False
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 30) else 0)
