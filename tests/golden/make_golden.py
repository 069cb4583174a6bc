#!/opt/conda/bin/python3.9
"""Generate the golden fixtures in tests/golden/ FROM THE REFERENCE ITSELF.

Run only in the build container, under the oracle interpreter (CPython 3.9 with
NumPy 1.26.4, SciPy 1.7.1, scikit-image 0.18.3 -- the pinned oracle of SURVEY.md
section 8c):

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_golden.py [--big]

It imports the reference's two hot-path modules unmodified from /root/reference
(`voxel_processor.VoxelProcessor`, `surface_extractor.SurfaceExtractor`; for the `consumers` fixture also
`volume_calculator.VolumeCalculator` and `obj_exporter.OBJExporter`) and
calls them (plus, for the per-stage vectors, the exact third-party calls the
reference makes at voxel_processor.py:62,68,88,91 and surface_extractor.py:51,55)
on seeded inputs, then stores inputs + expected outputs as .npz / .json.  The
fixtures are DATA (arrays, counts, SHA-256); nothing of the reference's source
is stored.  The reference never travels to the GPU box; these files do.
"""
import contextlib
import hashlib
import io
import json
import os
import sys
import time

sys.path.insert(0, "/root/reference")
import numpy as np  # noqa: E402
import warnings  # noqa: E402

warnings.filterwarnings("ignore")
import voxel_processor as RVP  # noqa: E402
import surface_extractor as RSE  # noqa: E402
from scipy import ndimage  # noqa: E402
from skimage import measure  # noqa: E402

assert RVP.SKIMAGE_AVAILABLE and RVP.SCIPY_AVAILABLE and RSE.SKIMAGE_AVAILABLE and RSE.SCIPY_AVAILABLE
HERE = os.path.dirname(os.path.abspath(__file__))


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ellipsoid_masks(nz, ny, nx):
    """Synthetic full ellipsoid of SURVEY.md section 8(d) (deterministic, float64)."""
    cx, cy, cz = (nx - 1) / 2.0, (ny - 1) / 2.0, (nz - 1) / 2.0
    ax, ay, az = 0.42 * nx, 0.40 * ny, 0.45 * nz
    x = np.arange(nx, dtype=np.float64)[None, :]
    y = np.arange(ny, dtype=np.float64)[:, None]
    ex = ((x - cx) / ax) ** 2
    ey = ((y - cy) / ay) ** 2
    out = []
    for z in range(nz):
        ez = ((float(z) - cz) / az) ** 2
        out.append((ex + ey) + ez <= 1.0)
    return out


def ref_field(volume_bool, manifold=True, add_padding=True):
    """The scalar field marching cubes sees (surface_extractor.py:43-55)."""
    v = volume_bool
    if manifold and add_padding:
        v = np.pad(v, 1, mode="constant", constant_values=False)
    vol = v.astype(float)
    if manifold:
        vol = ndimage.gaussian_filter(vol, sigma=0.5)
    return np.ascontiguousarray(vol, np.float32)


# --------------------------------------------------------------------------- (i) per-cell MC
def gen_mc_cells():
    rng = np.random.default_rng(20240601)
    vals = []
    K = 40
    for idx in range(256):
        bits = np.array([(idx >> i) & 1 for i in range(8)], bool)
        for k in range(K):
            if k < 28:  # generic continuous values
                hi = 0.5 + rng.random(8) * 0.5 + 1e-3
                lo = rng.random(8) * 0.499
            elif k < 36:  # lattice symmetric about the iso level -> exact / near ties in the MC33 tests
                d = rng.integers(1, 5, 8) * 0.125
                hi = 0.5 + d
                lo = 0.5 - rng.integers(1, 5, 8) * 0.125
            else:  # some corners exactly on the iso level (counts as "outside")
                hi = 0.5 + rng.integers(1, 4, 8) * 0.125
                lo = 0.5 - rng.integers(0, 3, 8) * 0.25
            vals.append(np.where(bits, hi, lo))
    vals = np.asarray(vals, np.float32)  # corner order v0..v7 of Lewiner
    # Lewiner corner i -> (z,y,x) of the 2x2x2 volume
    pos = [(0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0), (1, 0, 0), (1, 0, 1), (1, 1, 1), (1, 1, 0)]
    nv, nf, V, F = [], [], [], []
    for row in vals:
        vol = np.zeros((2, 2, 2), np.float32)
        for i, p in enumerate(pos):
            vol[p] = row[i]
        try:
            v, f, _, _ = measure.marching_cubes(vol, level=0.5)
            v = np.ascontiguousarray(v, np.float32)
            f = np.ascontiguousarray(f, np.int32)
        except (RuntimeError, ValueError):
            v = np.zeros((0, 3), np.float32)
            f = np.zeros((0, 3), np.int32)
        nv.append(len(v)); nf.append(len(f)); V.append(v); F.append(f)
    np.savez_compressed(os.path.join(HERE, "mc_cells.npz"), vals=vals,
                        nv=np.asarray(nv, np.int32), nf=np.asarray(nf, np.int32),
                        verts=np.concatenate(V), faces=np.concatenate(F))
    print("mc_cells:", len(vals), "cells,", sum(nv), "verts,", sum(nf), "faces")


# --------------------------------------------------------------------------- (ii) noise volumes through MC only
def gen_mc_noise():
    rng = np.random.default_rng(7)
    out = {}
    vols = {
        "uniform": rng.random((20, 22, 24)).astype(np.float32),
        "lattice": (0.5 + (rng.integers(-3, 4, (14, 15, 16)) * 0.125)).astype(np.float32),
        "smooth": ndimage.gaussian_filter(rng.random((18, 20, 22)), 1.0).astype(np.float32),
    }
    bin_ = rng.random((14, 16, 18)) < 0.5
    vols["binfield"] = ref_field(bin_)
    for k, vol in vols.items():
        v, f, _, _ = measure.marching_cubes(vol, level=0.5)
        out[k + "_vol"] = vol
        out[k + "_verts"] = np.ascontiguousarray(v, np.float32)
        out[k + "_faces"] = np.ascontiguousarray(f, np.int32)
        print("mc_noise", k, vol.shape, v.shape, f.shape)
    np.savez_compressed(os.path.join(HERE, "mc_noise.npz"), **out)


# --------------------------------------------------------------------------- (iii)/(vi) small volumes through the whole reference path
def blobs(rng, shape, sigma, thr):
    return ndimage.gaussian_filter(rng.random(shape), sigma) > thr


def run_reference_case(masks, sides, total_depth, mm_y, mm_x, close_ends=True, iterations=3,
                       create_manifold=True, manifold=True, add_padding=True, smooth_first=True):
    vp = RVP.VoxelProcessor()
    se = RSE.SurfaceExtractor()
    created = quiet(vp.create_voxel_data, list(masks), close_ends, *sides)
    depths = quiet(vp.calculate_slice_depths, total_depth)
    smoothed = quiet(vp.smooth_voxel_data, created, iterations=iterations, create_manifold=create_manifold) \
        if smooth_first else created
    res = quiet(se.extract_manifold_surface, smoothed, depths, mm_y, mm_x, smooth=True,
                manifold=manifold, add_padding=add_padding)
    d = {
        "masks": np.packbits(np.stack(masks, 0)), "shape": np.asarray(np.stack(masks, 0).shape, np.int64),
        "sides": np.asarray(sides, np.int64), "total_depth": np.float64(total_depth),
        "mm_y": np.float64(mm_y), "mm_x": np.float64(mm_x),
        "close_ends": np.bool_(close_ends), "iterations": np.int64(iterations),
        "create_manifold": np.bool_(create_manifold), "manifold": np.bool_(manifold),
        "add_padding": np.bool_(add_padding), "smooth_first": np.bool_(smooth_first),
        "created": np.packbits(created), "smoothed": np.packbits(smoothed),
        "depths": np.asarray(depths, np.float64),
        "field": ref_field(smoothed, manifold, add_padding),
        "is_none": np.bool_(res is None),
    }
    if res is not None:
        v, f = res
        d["verts"] = np.ascontiguousarray(v)
        d["faces"] = np.ascontiguousarray(f)
        d["verts_dtype"] = str(v.dtype); d["faces_dtype"] = str(f.dtype)
        if len(f) and f.ndim == 2:
            d["mesh_volume"] = np.float64(se.calculate_mesh_volume(v, f))
            d["surface_area"] = np.float64(se.calculate_surface_area(v, f))
        pc = vp.generate_point_cloud(smoothed, mm_x, mm_y, depths, subsample_factor=3) if len(depths) else None
        if pc is not None:
            d["point_cloud"] = np.asarray(pc)
    return d


def gen_pipeline_small():
    rng = np.random.default_rng(0)
    cases = {}
    # a: blobs with empty slices, holes in the end slices, three sides
    v = blobs(rng, (11, 24, 20), 1.5, 0.5)
    v[3] = False
    v[0] = False; v[0, 4:20, 3:17] = True; v[0, 8:14, 7:12] = False; v[0, 10:12, 9:10] = True   # nested hole
    v[-1] = False; v[-1, 2:9, 2:9] = True; v[-1, 4:6, 4:7] = False; v[-1, 12:22, 10:19] = True; v[-1, 15:18, 12:16] = False
    v[-1, 12:22, 14] |= True
    cases["a_blobs"] = run_reference_case(list(v), (2, 7, 2), 6.0, 0.7, 0.9)
    # b: dense binary noise -> every MC33 case on a binary-derived (near-tie) field
    v = rng.random((12, 14, 16)) < 0.5
    cases["b_noise"] = run_reference_case(list(v), (0, 12, 0), 12.0, 1.0, 1.0, iterations=1)
    # c: sparse noise, smoothing disabled pieces
    v = rng.random((9, 17, 13)) < 0.3
    cases["c_noise_nomanifold_smooth"] = run_reference_case(list(v), (3, 3, 3), 4.5, 95.03 / 17, 143.1 / 13,
                                                             iterations=2, create_manifold=False)
    # d: add_padding=False (Gaussian reflect acts on real data at the border)
    v = blobs(rng, (10, 18, 21), 1.2, 0.48)
    cases["d_nopad"] = run_reference_case(list(v), (0, 10, 0), 5.0, 0.5, 0.25, add_padding=False)
    # e: manifold=False: raw 0/1 field, no shift, no unique
    v = blobs(rng, (8, 15, 14), 1.2, 0.5)
    cases["e_nomanifold"] = run_reference_case(list(v), (2, 4, 2), 3.0, 1.5, 2.5, manifold=False)
    # f: close_ends=False, iterations=0, uniform depths (side_1 == 0)
    v = blobs(rng, (7, 16, 16), 1.0, 0.5)
    cases["f_noclose"] = run_reference_case(list(v), (3, 0, 4), 7.0, 1.0, 1.0, close_ends=False, iterations=0)
    # g: empty volume -> extract returns None
    v = np.zeros((5, 8, 8), bool)
    cases["g_empty"] = run_reference_case(list(v), (0, 5, 0), 5.0, 1.0, 1.0)
    # h: full volume -> surface only from the padding
    v = np.ones((4, 6, 7), bool)
    cases["h_full"] = run_reference_case(list(v), (1, 2, 1), 2.0, 1.0, 1.0)
    # i: noise, unsmoothed volume straight into extraction (all ambiguous cases on the raw close-ends output)
    v = rng.random((16, 20, 24)) < 0.45
    cases["i_noise_raw"] = run_reference_case(list(v), (4, 8, 4), 8.0, 0.37, 0.41, smooth_first=False)
    # j: wider than one 64-bit word in x, odd sizes
    v = blobs(rng, (13, 37, 150), 2.0, 0.5)
    cases["j_wide"] = run_reference_case(list(v), (3, 7, 3), 9.0, 0.8, 0.6)
    flat = {}
    for cn, d in cases.items():
        for k, val in d.items():
            flat[cn + "__" + k] = val
        print("pipeline_small", cn, d["shape"], "none" if d["is_none"] else (d["verts"].shape, d["faces"].shape))
    np.savez_compressed(os.path.join(HERE, "pipeline_small.npz"), **flat)


# --------------------------------------------------------------------------- (iii') binary stages alone
def gen_binary_stages():
    rng = np.random.default_rng(3)
    out = {}
    for i, (shape, p) in enumerate([((9, 20, 70), 0.5), ((6, 33, 129), 0.7), ((12, 16, 64), 0.35), ((3, 5, 7), 0.5),
                                    ((2, 9, 9), 0.6), ((1, 6, 6), 0.5)]):
        v = rng.random(shape) < p
        if i == 1:
            v = ndimage.gaussian_filter(rng.random(shape), 1.5) > 0.5
        vp = RVP.VoxelProcessor()
        created = quiet(vp.create_voxel_data, list(v), True, 0, shape[0], 0)
        out["s%d_shape" % i] = np.asarray(shape, np.int64)
        out["s%d_in" % i] = np.packbits(v)
        out["s%d_created" % i] = np.packbits(created)
        out["s%d_active" % i] = np.int64(created.sum())
        for it, cm in [(3, True), (1, False), (0, True)]:
            sm = vp.smooth_voxel_data(v, iterations=it, create_manifold=cm)
            out["s%d_smooth_%d_%d" % (i, it, int(cm))] = np.packbits(sm)
        # 2-D fill holes of each slice (voxel_processor.py:62 uses it on slices 0 and -1)
        out["s%d_fill" % i] = np.packbits(np.stack([ndimage.binary_fill_holes(s) for s in v]))
    np.savez_compressed(os.path.join(HERE, "binary_stages.npz"), **out)
    print("binary_stages done")


# --------------------------------------------------------------------------- (iv) physical-units ellipsoid, full mesh
def gen_ellipsoid_cfg1():
    nz, ny, nx = 64, 128, 128
    masks = ellipsoid_masks(nz, ny, nx)
    d = run_reference_case(masks, (8, 48, 8), 6.0, 95.03 / ny, 143.1 / nx)
    keep = {k: d[k] for k in ("shape", "sides", "total_depth", "mm_y", "mm_x", "depths", "verts", "faces",
                              "mesh_volume", "surface_area")}
    keep["mask_sha"] = sha(np.packbits(np.stack(masks, 0)))
    keep["field_sha"] = sha(d["field"])
    keep["created_sha"] = sha(d["created"]); keep["smoothed_sha"] = sha(d["smoothed"])
    np.savez_compressed(os.path.join(HERE, "ellipsoid_64x128x128.npz"), **keep)
    print("ellipsoid cfg1", d["verts"].shape, d["faces"].shape)


# --------------------------------------------------------------------------- (v) hashes at benchmark sizes
def gen_hashes(sizes):
    path = os.path.join(HERE, "ellipsoid_hashes.json")
    res = json.load(open(path)) if os.path.exists(path) else {}
    for (nz, ny, nx) in sizes:
        key = "%dx%dx%d" % (nz, ny, nx)
        t0 = time.time()
        masks = ellipsoid_masks(nz, ny, nx)
        vp = RVP.VoxelProcessor(); se = RSE.SurfaceExtractor()
        created = quiet(vp.create_voxel_data, masks, True, 0, nz, 0)
        depths = quiet(vp.calculate_slice_depths, float(nz))
        smoothed = vp.smooth_voxel_data(created, iterations=3, create_manifold=True)
        field = ref_field(smoothed)
        v, f = quiet(se.extract_manifold_surface, smoothed, depths, 1.0, 1.0)
        res[key] = {
            "shape": [nz, ny, nx], "active": int(created.sum()), "smoothed_active": int(smoothed.sum()),
            "mask_sha256": sha(np.packbits(np.stack(masks, 0))), "created_sha256": sha(np.packbits(created)),
            "smoothed_sha256": sha(np.packbits(smoothed)), "field_f32_sha256": sha(field),
            "n_vertices": int(len(v)), "n_faces": int(len(f)),
            "vertices_f32_sha256": sha(np.ascontiguousarray(v, np.float32)),
            "faces_i64_sha256": sha(np.ascontiguousarray(f, np.int64)),
            "surface_area": float(se.calculate_surface_area(v, f)),
            "reference_seconds": round(time.time() - t0, 2),
        }
        print("hash", key, res[key]["n_vertices"], res[key]["n_faces"], res[key]["reference_seconds"], "s")
        json.dump(res, open(path, "w"), indent=1, sort_keys=True)


def gen_consumers():
    """Rows N1 / N4 of SURVEY.md 8(f): the reference's VolumeCalculator and OBJExporter on seeded inputs."""
    import tempfile
    import volume_calculator as RVC
    import obj_exporter as ROE
    rng = np.random.default_rng(11)
    vc = RVC.VolumeCalculator()
    out = {}
    cases = []
    for ci, (shape, sigma, thr, sides, depth) in enumerate([((20, 24, 11), 1.5, 0.5, (4, 12, 4), 6.0),
                                                           ((9, 70, 130), 2.0, 0.52, (0, 9, 0), 3.3),
                                                           ((16, 40, 64), 1.0, 0.45, (5, 6, 5), 10.0),
                                                           ((6, 10, 10), 1.0, 2.0, (0, 6, 0), 1.0)]):      # last one: empty volume
        vol = blobs(rng, shape, sigma, thr)
        vp = RVP.VoxelProcessor()
        vp.side_0_count, vp.side_1_count, vp.side_2_count = sides
        depths = np.asarray(quiet(vp.calculate_slice_depths, depth), np.float64)
        mmx, mmy, mms = 95.03 / shape[2], 143.1 / shape[1], depth / shape[0]
        k = "c%d_" % ci
        out[k + "vol"] = np.packbits(vol); out[k + "shape"] = np.asarray(shape, np.int64)
        out[k + "depths"] = depths; out[k + "mm"] = np.asarray([mmx, mmy, mms], np.float64)
        out[k + "voxel_volume"] = np.float64(vc.calculate_voxel_volume(vol, mmx, mmy, mms))
        out[k + "voxel_volume_var"] = np.float64(vc.calculate_voxel_volume_variable_depth(vol, mmx, mmy, depths))
        out[k + "voxel_volume_var_short"] = np.float64(vc.calculate_voxel_volume_variable_depth(vol, mmx, mmy, depths[:3]))
        if vol.any():
            b = vc.calculate_bounding_box(vol, mmx, mmy, mms)
            out[k + "bbox"] = np.asarray([*b["x"], *b["y"], *b["z"], *b["dimensions"]], np.float64)
        b = vc.calculate_bounding_box_variable_depth(vol, mmx, mmy, depths)
        out[k + "bbox_var"] = np.asarray([*b["x"], *b["y"], *b["z"], *b["dimensions"]], np.float64)
        out[k + "density"] = np.float64(vc.calculate_density(12.5, 95.03, 143.1, depth))
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            props = vc.analyze_object_properties(vol, 3.25, None if ci % 2 else 2.75, 0.0 if ci == 2 else 7.5, mmx, mmy,
                                                 depths, 95.03, 143.1, depth)
        out[k + "analyze_stdout"] = np.frombuffer(buf.getvalue().encode("utf-8"), np.uint8)
        out[k + "analyze_nums"] = np.asarray([props["volume_mm3"], props["voxel_volume_mm3"], props["density"],
                                              *props["dimensions"]], np.float64)
        cases.append(ci)
    out["n_cases"] = np.int64(len(cases))
    # OBJ export of a real (small) mesh + an empty-face mesh
    m = run_reference_case(ellipsoid_masks(12, 20, 28), (2, 8, 2), 6.0, 143.1 / 20, 95.03 / 28)
    oe = ROE.OBJExporter()
    with tempfile.TemporaryDirectory() as d:
        for name, (v, f) in {"mesh": (m["verts"], m["faces"]), "nofaces": (m["verts"][:5], np.array([]))}.items():
            pth = os.path.join(d, name + ".obj")
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                ok = oe.export_to_obj(v, f, pth)
            out["obj_%s_verts" % name] = np.ascontiguousarray(v)
            out["obj_%s_faces" % name] = np.ascontiguousarray(f)
            out["obj_%s_ok" % name] = np.bool_(ok)
            out["obj_%s_bytes" % name] = np.frombuffer(open(pth, "rb").read(), np.uint8)
            out["obj_%s_stdout" % name] = np.frombuffer(buf.getvalue().replace(d, "<DIR>").encode("utf-8"), np.uint8)
    np.savez_compressed(os.path.join(HERE, "consumers.npz"), **out)
    print("consumers:", len(cases), "volume cases,", len(m["verts"]), "vertices in the OBJ")


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if not a.startswith("--")]
    def want(n):
        return not which or n in which
    if want("cells"): gen_mc_cells()
    if want("noise"): gen_mc_noise()
    if want("small"): gen_pipeline_small()
    if want("binary"): gen_binary_stages()
    if want("cfg1"): gen_ellipsoid_cfg1()
    if want("consumers"): gen_consumers()
    if want("hashes"):
        sizes = [(64, 128, 128), (96, 80, 112), (256, 256, 256)]
        if "--big" in sys.argv:
            sizes += [(512, 512, 512)]
        if "--huge" in sys.argv:
            sizes = [(1024, 1024, 1024)]
        gen_hashes(sizes)
