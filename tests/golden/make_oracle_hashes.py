#!/usr/bin/env python3
"""ORACLE-derived hashes for ellipsoid stacks the reference itself cannot process in the build container (its float64
field of the 1024x1024x2048 stack alone is 17 GB and it holds several): the pinned C/NumPy oracle (oracle/, bit for
bit equal to the reference on every fixture of make_golden.py up to 1024^3) runs the whole path and the counts and
SHA-256 values go to tests/golden/ellipsoid_hashes_oracle.json.  Entries are labelled "derived_from": "oracle".

    python tests/golden/make_oracle_hashes.py 2048 1024 1024
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "ellipsoid_hashes_oracle.json")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    nz, ny, nx = [int(x) for x in sys.argv[1:4]]
    t0 = time.time()
    masks = O.ellipsoid_masks(nz, ny, nx)
    vp, se = O.VoxelProcessor(), O.SurfaceExtractor()
    created = vp.create_voxel_data(masks, True, 0, nz, 0)
    ent = {"shape": [nz, ny, nx], "derived_from": "oracle", "active": int(created.sum()),
           "mask_sha256": sha(np.packbits(np.stack(masks))), "created_sha256": sha(np.packbits(created))}
    del masks
    sm = vp.smooth_voxel_data(created, 3, True)
    del created
    ent["smoothed_active"] = int(sm.sum())
    ent["smoothed_sha256"] = sha(np.packbits(sm))
    depths = vp.calculate_slice_depths(float(nz))
    vc = O.VolumeCalculator()
    ent["voxel_volume"] = float(vc.calculate_voxel_volume_variable_depth(sm, 1.0, 1.0, depths))
    bb = vc.calculate_bounding_box_variable_depth(sm, 1.0, 1.0, depths)
    ent["bbox"] = {k: [float(x) for x in bb[k]] for k in ("x", "y", "z", "dimensions")}
    v, f = se.extract_manifold_surface(sm, depths, 1.0, 1.0)
    ent.update({"n_vertices": int(len(v)), "n_faces": int(len(f)), "vertices_f32_sha256": sha(v), "faces_i64_sha256": sha(f),
                "surface_area": float(se.calculate_surface_area(v, f)), "oracle_seconds": round(time.time() - t0, 1)})
    allh = json.load(open(OUT)) if os.path.exists(OUT) else {}
    allh["%dx%dx%d" % (nz, ny, nx)] = ent
    json.dump(allh, open(OUT, "w"), indent=1, sort_keys=True)
    print(json.dumps(ent))


if __name__ == "__main__":
    main()
