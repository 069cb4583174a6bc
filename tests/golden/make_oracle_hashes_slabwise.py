#!/usr/bin/env python3
"""ORACLE-derived hashes for ellipsoid stacks too large for ONE pass of the oracle in the build container (BASELINE
configs[4], 2048x2048x4096: the bool stack alone is 17 GB and orc_field holds two float64 copies of the padded volume).

The pinned C/NumPy oracle (oracle/) runs the path chunk by chunk along z, every chunk with enough extra slices on either
side that what is kept of it is exactly what one pass over the whole stack gives:
  * orc_close_ends on a chunk fills the holes of the CHUNK's end slices -- wrong where they are not the stack's ends -- and
    that reaches one slice further (the recurrence is a 3-tap stencil): 2 slices;
  * opening + 3 closings: 8 passes of a radius-1 structuring element: 8 slices;
  * the 5-tap Gaussian along z: 2 slices; marching cubes reads one field slice beyond the owned cell layers: 1 slice
so MARGIN = 13 slices are computed and thrown away on either side (none at the stack's own ends).
Marching cubes runs on the owned cell layers with the chunk's z offset; np.unique per chunk; the rows on the plane a
chunk shares with the next one belong to the next one (they close the chunk's sorted list and open the next chunk's), so
the global list is the concatenation of what the chunks keep, faces in chunk order = the reference's cell order.
This script is written directly on the oracle's stage functions (it does not use tomography_3d_reconstructor_amd.slab) and
is CHECKED against the REFERENCE-derived hashes of tests/golden/ellipsoid_hashes.json:

    python tests/golden/make_oracle_hashes_slabwise.py --check 512 512 512 64      # 8 chunks; compares, writes nothing
    python tests/golden/make_oracle_hashes_slabwise.py 4096 2048 2048 128 --workers 4

Entries go to tests/golden/ellipsoid_hashes_oracle.json labelled "derived_from": "oracle (slab-wise)".
"""
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "ellipsoid_hashes_oracle.json")
MARGIN = 13


def mask_slices(gz, ny, nx, a, b):
    """Slices [a, b) of O.ellipsoid_masks(gz, ny, nx) (same float64 formula, slice by slice)."""
    cx, cy, cz = (nx - 1) / 2.0, (ny - 1) / 2.0, (gz - 1) / 2.0
    ax, ay, az = 0.42 * nx, 0.40 * ny, 0.45 * gz
    x = np.arange(nx, dtype=np.float64)[None, :]
    y = np.arange(ny, dtype=np.float64)[:, None]
    exy = ((x - cx) / ax) ** 2 + ((y - cy) / ay) ** 2
    return np.stack([exy + ((float(z) - cz) / az) ** 2 <= 1.0 for z in range(a, b)])


def chunk(args):
    gz, ny, nx, z0, z1, tmp, k = args
    t0 = time.time()
    first, last = z0 == 0, z1 == gz
    a, b = max(0, z0 - MARGIN), min(gz, z1 + MARGIN)
    m = mask_slices(gz, ny, nx, a, b)
    own = slice(z0 - a, z1 - a)
    res = {"k": k, "z0": z0, "z1": z1}
    np.save(os.path.join(tmp, "bm%04d.npy" % k), np.packbits(m[own]))
    created = O.close_ends(m)
    del m
    np.save(os.path.join(tmp, "bc%04d.npy" % k), np.packbits(created[own]))
    res["active"] = int(created[own].sum())
    sm = O.smooth(created, 3, True)
    del created
    so = sm[own]
    np.save(os.path.join(tmp, "bs%04d.npy" % k), np.packbits(so))
    res["slice_counts"] = so.sum(axis=(1, 2)).astype(np.int64)
    idx = np.nonzero(so.any(axis=0))
    zany = np.nonzero(so.any(axis=(1, 2)))[0]
    res["box"] = None if len(zany) == 0 else (int(zany.min()) + z0, int(zany.max()) + z0, int(idx[0].min()), int(idx[0].max()),
                                               int(idx[1].min()), int(idx[1].max()))
    # field of slices [lo, hi): exact on [z0, z1] (the Gaussian reaches 2 slices, the chunk's zero padding is the stack's own
    # only at the stack's ends)
    lo, hi = (0 if first else z0 - 2), (gz if last else z1 + 3)
    f = O.field(sm[lo - a:hi - a], True, True)            # local padded index l <-> global padded index lo + l
    del sm
    Za = 0 if first else z0 + 1                            # owned cell layers [Za, Zb) in padded coordinates
    Zb = gz + 1 if last else z1 + 1
    sub = f[Za - lo:Zb - lo + 1]
    try:
        v, fc = O.marching_cubes(sub, 0.5, Za)
    except (ValueError, RuntimeError):
        v, fc = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)
    del f, sub
    depths = np.full(gz, 1.0)                              # VoxelProcessor.calculate_slice_depths(float(gz)) with sides (0, gz, 0)
    rows = O.finalize_vertices(v, depths, 1.0, 1.0, True, True)
    ztop = None if last else float(O.finalize_vertices(np.array([[Zb, 1.0, 1.0]], np.float32), depths, 1.0, 1.0, True, True)[0, 0])
    uniq, inv = np.unique(rows, axis=0, return_inverse=True)
    faces = np.asarray(inv).reshape(-1)[fc].astype(np.int64) if len(fc) else np.zeros((0, 3), np.int64)
    n_top = 0 if ztop is None else int((uniq[:, 0] == ztop).sum())
    if n_top:
        assert (uniq[len(uniq) - n_top:, 0] == ztop).all() and (uniq[:len(uniq) - n_top, 0] < ztop).all()
    np.save(os.path.join(tmp, "u%04d.npy" % k), uniq)
    np.save(os.path.join(tmp, "f%04d.npy" % k), faces)
    res["n_top"], res["nu"], res["seconds"] = n_top, len(uniq), round(time.time() - t0, 1)
    print("chunk %d [%d, %d): %d rows (%d shared with the next), %d faces, %.0f s" % (k, z0, z1, len(uniq), n_top, len(faces), res["seconds"]),
          flush=True)
    return res


def row_view(a):
    return np.ascontiguousarray(a).view([("z", "<f4"), ("y", "<f4"), ("x", "<f4")]).reshape(-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", type=int, nargs=3)
    ap.add_argument("thickness", type=int, nargs="?", default=128)
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--check", action="store_true", help="compare with tests/golden/ellipsoid_hashes.json (reference-derived), write nothing")
    a = ap.parse_args()
    gz, ny, nx = a.shape
    t0 = time.time()
    cuts = list(range(0, gz, a.thickness)) + [gz]
    if len(cuts) > 2 and cuts[-1] - cuts[-2] < MARGIN + 1:
        cuts.pop(-2)
    tmp = tempfile.mkdtemp(prefix="slabwise_", dir=os.environ.get("TMPDIR", "/tmp"))
    jobs = [(gz, ny, nx, cuts[i], cuts[i + 1], tmp, i) for i in range(len(cuts) - 1)]
    with mp.get_context("fork").Pool(a.workers) as pool:
        parts = sorted(pool.imap_unordered(chunk, jobs), key=lambda r: r["k"])
    hm, hc, hs, hv, hf = (hashlib.sha256() for _ in range(5))
    for p in parts:
        for h, tag in ((hm, "bm"), (hc, "bc"), (hs, "bs")):
            h.update(np.load(os.path.join(tmp, "%s%04d.npy" % (tag, p["k"]))).tobytes())
    counts = np.concatenate([p["slice_counts"] for p in parts])
    vc = O.VolumeCalculator()
    depths = np.full(gz, 1.0)
    total = 0.0                                           # volume_calculator.py:26-35, sequential over z
    for z in range(gz):
        total += counts[z] * (1.0 * 1.0 * depths[z])
    boxes = [p["box"] for p in parts if p["box"] is not None]
    ent = {"shape": [gz, ny, nx], "derived_from": "oracle (slab-wise, %d chunks)" % len(parts),
           "active": int(sum(p["active"] for p in parts)), "smoothed_active": int(counts.sum()),
           "mask_sha256": hm.hexdigest(), "created_sha256": hc.hexdigest(), "smoothed_sha256": hs.hexdigest(),
           "voxel_volume": float(total)}
    if boxes:
        zmin, zmax = min(b[0] for b in boxes), max(b[1] for b in boxes)
        ymin, ymax = min(b[2] for b in boxes), max(b[3] for b in boxes)
        xmin, xmax = min(b[4] for b in boxes), max(b[5] for b in boxes)
        cum = np.cumsum(np.concatenate([[0], depths]))    # volume_calculator.py:62-94
        bz = (float(cum[zmin]), float(cum[min(zmax + 1, len(cum) - 1)]))
        bx, by = (xmin * 1.0, xmax * 1.0), (ymin * 1.0, ymax * 1.0)
        ent["bbox"] = {"x": list(bx), "y": list(by), "z": list(bz), "dimensions": [bx[1] - bx[0], by[1] - by[0], bz[1] - bz[0]]}
    # global numbering: what a chunk keeps = its rows without those on the plane shared with the next chunk
    kept = [p["nu"] - p["n_top"] for p in parts]
    offs = np.concatenate([[0], np.cumsum(kept)]).astype(np.int64)
    nfaces = 0
    area = 0.0
    nxt = np.load(os.path.join(tmp, "u%04d.npy" % 0))
    for i, p in enumerate(parts):
        uniq = nxt
        nxt = np.load(os.path.join(tmp, "u%04d.npy" % (i + 1))) if i + 1 < len(parts) else None
        faces = np.load(os.path.join(tmp, "f%04d.npy" % i))
        gid = np.arange(len(uniq), dtype=np.int64) + offs[i]
        if p["n_top"]:
            top = uniq[kept[i]:]
            pos = np.searchsorted(row_view(nxt), row_view(top))
            assert (pos < len(nxt)).all() and np.array_equal(nxt[pos], top), "a shared-plane row is missing in the next chunk"
            gid[kept[i]:] = pos + offs[i + 1]
        hv.update(np.ascontiguousarray(uniq[:kept[i]]).tobytes())
        if len(faces):
            g = gid[faces]
            keep = (g[:, 0] != g[:, 1]) & (g[:, 1] != g[:, 2]) & (g[:, 0] != g[:, 2])
            g = np.ascontiguousarray(g[keep])
            hf.update(g.tobytes())
            nfaces += len(g)
            v0, v1, v2 = (uniq[faces[keep][:, j]].astype(np.float64) for j in range(3))
            area += float(0.5 * np.linalg.norm(np.cross(v1 - v0, v2 - v0), axis=1).sum())
    ent.update({"n_vertices": int(offs[-1]), "n_faces": int(nfaces), "vertices_f32_sha256": hv.hexdigest(),
                "faces_i64_sha256": hf.hexdigest(), "surface_area_f64": area, "oracle_seconds": round(time.time() - t0, 1),
                "chunk_thickness": a.thickness, "workers": a.workers})
    for fn in os.listdir(tmp):
        os.unlink(os.path.join(tmp, fn))
    os.rmdir(tmp)
    key = "%dx%dx%d" % (gz, ny, nx)
    print(json.dumps(ent))
    if a.check:
        ref = json.load(open(os.path.join(ROOT, "tests", "golden", "ellipsoid_hashes.json")))[key]
        bad = [k for k in ("active", "smoothed_active", "mask_sha256", "created_sha256", "smoothed_sha256", "n_vertices", "n_faces",
                           "vertices_f32_sha256", "faces_i64_sha256") if ref[k] != ent[k]]
        print("CHECK against the reference-derived entry %s: %s" % (key, "all equal" if not bad else "MISMATCH in %r" % bad))
        sys.exit(1 if bad else 0)
    allh = json.load(open(OUT)) if os.path.exists(OUT) else {}
    allh[key] = ent
    json.dump(allh, open(OUT, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
