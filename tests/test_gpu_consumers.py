"""GPU tier for the rows either side of the hot path: VolumeCalculator on the resident bit volume,
ImageLoader with the fused threshold + pack upload, and the orchestrator's call sequence
(tomography_3d_reconstruction.py:271-323) replayed against the drop-in classes -- all against the fixtures generated
from the reference and the CPU oracle."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tomography_3d_reconstructor_amd import _devcache, pipeline
from tomography_3d_reconstructor_amd import SurfaceExtractor, VoxelProcessor
from tomography_3d_reconstructor_amd.image_loader import ImageLoader
from tomography_3d_reconstructor_amd.obj_exporter import OBJExporter
from tomography_3d_reconstructor_amd.volume_calculator import VolumeCalculator
from test_consumers_cpu import G, check_volume_calculator

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("ci", range(int(G["n_cases"])))
def test_volume_calculator_matches_reference(dev, ci):
    check_volume_calculator(VolumeCalculator(), ci)


@pytest.mark.parametrize("shape", [(3, 5, 7), (7, 33, 130), (65, 16, 64), (2, 1, 1), (5, 40, 1100)])
def test_slice_counts_and_bbox(dev, shape):
    rng = np.random.default_rng(3)
    a = rng.random(shape) < 0.3
    a[:, :, : shape[2] // 3] = False
    a[0] = False
    vol = pipeline.pack(torch.from_numpy(a.view(np.uint8)).to(dev))
    assert np.array_equal(pipeline.slice_counts(vol).cpu().numpy(), a.reshape(shape[0], -1).sum(1))
    z, y, x = np.where(a)
    exp = None if len(z) == 0 else (z.min(), z.max(), y.min(), y.max(), x.min(), x.max())
    assert pipeline.bounding_box(vol) == (None if exp is None else tuple(int(v) for v in exp))
    empty = pipeline.pack(torch.zeros(shape, dtype=torch.uint8, device=dev))
    assert pipeline.bounding_box(empty) is None and int(pipeline.slice_counts(empty).sum()) == 0


@pytest.mark.parametrize("shape,thr", [((3, 5, 7), 200), ((4, 9, 64), 0), ((2, 17, 130), 255), ((3, 8, 1024), 128),
                                       ((2, 4, 48), 256), ((2, 3, 1100), 1)])
def test_pack_threshold(dev, shape, thr):
    rng = np.random.default_rng(4)
    g = rng.integers(0, 256, shape, dtype=np.uint8)
    g[0, 0, :3] = [thr - 1 if thr > 0 else 0, min(thr, 255), min(thr + 1, 255)]
    vol = pipeline.pack_threshold(torch.from_numpy(g).to(dev), thr)
    assert np.array_equal(pipeline.unpack(vol).cpu().numpy(), g >= thr)


def write_stack(root, masks_u8, sides):
    from PIL import Image
    k = 0
    for si, n in enumerate(sides):
        os.makedirs(os.path.join(root, "Section_%d" % si), exist_ok=True)
        for j in range(n):
            # descending / negative numbering on side 0, like the reference's generator produces (simple_generator.py)
            num = (j - n) if si == 0 else j + 1
            Image.fromarray(masks_u8[k], mode="L").save(os.path.join(root, "Section_%d" % si, "Mask_Patient_%d.png" % num))
            k += 1


def test_image_loader_matches_host_restatement_and_feeds_the_device(dev, tmp_path):
    nz, ny, nx = 20, 48, 80
    sides = (4, 12, 4)
    rng = np.random.default_rng(6)
    masks = np.stack(O.ellipsoid_masks(nz, ny, nx))
    grey = np.where(masks, 255, 0).astype(np.uint8)
    grey[rng.random(grey.shape) < 0.05] = rng.integers(150, 256, int((rng.random(grey.shape) < 0.05).sum()) or 1)[0]
    write_stack(str(tmp_path), grey, sides)
    exp, counts, files = O.load_masks(str(tmp_path), 200)
    ld = ImageLoader()
    with contextlib.redirect_stdout(io.StringIO()) as out:
        assert ld.load_mask_images(str(tmp_path), 200, [True, True, True]) is True
    assert "Found masks - Side_0: 4, Side_1: 12, Side_2: 4" in out.getvalue()
    assert ld.get_side_counts() == counts == sides and ld.get_num_slices() == nz
    assert ld.get_image_dimensions() == (nx, ny) and ld.mask_files == files
    got = ld.get_mask_images()
    assert len(got) == nz and all(m.dtype == np.bool_ and np.array_equal(m, e) for m, e in zip(got, exp))
    # the packed volume is already on the device: create_voxel_data must not upload (cache hit through the views' base)
    base = got[0].base
    cached = _devcache.get(base)
    assert cached is not None and np.array_equal(pipeline.unpack(cached).cpu().numpy(), np.stack(exp))
    vp, ovp = VoxelProcessor(), O.VoxelProcessor()
    with contextlib.redirect_stdout(io.StringIO()):
        created = vp.create_voxel_data(got, True, *sides)
        ocreated = ovp.create_voxel_data(exp, True, *sides)
    assert np.array_equal(created, ocreated)
    assert np.array_equal(pipeline.unpack(_devcache.get(base)).cpu().numpy(), np.stack(exp)), "loader volume was overwritten"
    # missing folder / disabled sides, as the reference reports them
    with contextlib.redirect_stdout(io.StringIO()) as out:
        assert ImageLoader().load_mask_images(str(tmp_path / "nowhere"), 200) is False
    assert "Folder Section_0 not found" in out.getvalue()
    ld2 = ImageLoader()
    with contextlib.redirect_stdout(io.StringIO()):
        assert ld2.load_mask_images(str(tmp_path), 200, [False, True, False]) is True
    assert ld2.get_side_counts() == (0, 12, 0) and ld2.get_num_slices() == 12


def test_orchestrator_sequence_end_to_end(dev, tmp_path):
    """main() of the reference (tomography_3d_reconstruction.py:271-323) as a call sequence: load -> create ->
    depths -> volumes (raw / smoothed) -> mesh volume -> surface area -> analyse -> OBJ; drop-in classes vs the CPU
    oracle on the same PNG stack (config 1 of BASELINE.json: 128x128x64 half-ellipsoid-like stack)."""
    nz, ny, nx = 64, 128, 128
    sides = (8, 48, 8)
    x_mm, y_mm, depth_mm = 95.03, 143.1, 6.0
    masks = np.stack(O.ellipsoid_masks(nz, ny, nx))
    write_stack(str(tmp_path), np.where(masks, 255, 0).astype(np.uint8), sides)
    ld = ImageLoader()
    with contextlib.redirect_stdout(io.StringIO()):
        assert ld.load_mask_images(str(tmp_path), 200, [True, True, True])
    w, h = ld.get_image_dimensions()
    mmx, mmy = x_mm / w, y_mm / h
    omasks, _, _ = O.load_masks(str(tmp_path), 200)

    def run(vp, se, vc, mask_list):
        with contextlib.redirect_stdout(io.StringIO()) as out:
            vol = vp.create_voxel_data(mask_list, True, *sides)
            depths = vp.calculate_slice_depths(depth_mm)
            raw = vc.calculate_voxel_volume_variable_depth(vol, mmx, mmy, depths)
            sm = vp.smooth_voxel_data(vol, iterations=3, create_manifold=True)
            proc = vc.calculate_voxel_volume_variable_depth(sm, mmx, mmy, depths)
            v, f = se.extract_manifold_surface(sm, depths, mmy, mmx, smooth=True, manifold=True, add_padding=True)
            mv = se.calculate_mesh_volume(v, f)
            area = se.calculate_surface_area(v, f)
            props = vc.analyze_object_properties(vol, proc, mv, area, mmx, mmy, depths, x_mm, y_mm, depth_mm)
        return vol, depths, raw, sm, proc, v, f, mv, area, props, out.getvalue()

    a = run(VoxelProcessor(), SurfaceExtractor(), VolumeCalculator(), ld.get_mask_images())
    b = run(O.VoxelProcessor(), O.SurfaceExtractor(), O.VolumeCalculator(), omasks)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])
    assert np.float64(a[2]).tobytes() == np.float64(b[2]).tobytes() and np.float64(a[4]).tobytes() == np.float64(b[4]).tobytes()
    assert a[5].dtype == np.float32 and a[5].tobytes() == b[5].tobytes() and a[6].dtype == np.int64 and np.array_equal(a[6], b[6])
    assert abs(a[7] - b[7]) <= 1e-6 * abs(b[7]) and abs(float(a[8]) - float(b[8])) <= 1e-5 * abs(float(b[8]))   # tree reductions
    assert a[9]["dimensions"] == b[9]["dimensions"] and a[9]["voxel_volume_mm3"] == b[9]["voxel_volume_mm3"]
    la, lb = a[10].splitlines(), b[10].splitlines()
    assert la[0] == "Voxels: (64, 128, 128), active: %s" % format(int(a[0].sum()), ",")        # voxel_processor.py:52
    assert "Surface: %d vertices, %d faces" % (len(a[5]), len(a[6])) in la                       # surface_extractor.py:70
    assert [x for x in la if x.startswith("Dimensions")] == [x for x in lb if x.startswith("Dimensions")]
    assert la[-1] == lb[-1] and la[-1].startswith("Density: ")
    # mesh volume vs voxel volume: sanity band only (the two discretisations differ by a few 1e-3 here)
    assert abs(a[7] - a[4]) <= 1e-2 * a[4]
    pth = str(tmp_path / "model.obj")
    with contextlib.redirect_stdout(io.StringIO()):
        assert OBJExporter().export_to_obj(a[5], a[6], pth)
    assert open(pth).read() == O.obj_text(b[5], b[6])


def test_create_voxel_data_input_forms(dev):
    """Boundary behaviour of create_voxel_data for the input forms the reference accepts through np.stack: a ragged list
    raises ValueError like np.stack does, uint8 0/255 masks count as `!= 0`, a list of views of one array (what the
    loader hands out) and a list of separate arrays give the same volume."""
    rng = np.random.default_rng(8)
    v = rng.random((7, 40, 70)) < 0.4
    sides = (2, 3, 2)
    ref = O.VoxelProcessor()
    with contextlib.redirect_stdout(io.StringIO()):
        exp = ref.create_voxel_data(list(v), True, *sides)
        a = VoxelProcessor().create_voxel_data([m.copy() for m in v], True, *sides)
        b = VoxelProcessor().create_voxel_data([v[i] for i in range(len(v))], True, *sides)            # views of one base
        c = VoxelProcessor().create_voxel_data([(m * 255).astype(np.uint8) for m in v], True, *sides)   # grey-level masks
        d = VoxelProcessor().create_voxel_data([m.copy() for m in v], False, *sides)
    assert np.array_equal(a, exp) and np.array_equal(b, exp) and np.array_equal(c, exp)
    assert d.dtype == np.bool_ and np.array_equal(d, v) and not np.shares_memory(d, v)
    with pytest.raises(ValueError):
        VoxelProcessor().create_voxel_data([v[0], v[1][:-1]], True, 0, 2, 0)
    with pytest.raises(ValueError, match="Load masks first, hmm."):
        VoxelProcessor().create_voxel_data([], True)


def test_repeated_calls_reuse_the_device_result(dev):
    """the orchestrator repeats smooth_voxel_data / extract_manifold_surface with the same arguments
    (tomography_3d_reconstruction.py:201-243).  The device result is remembered; every call still returns FRESH, equal
    host arrays; other parameters or another volume never get a remembered result."""
    from tomography_3d_reconstructor_amd import _memo
    _memo.clear()
    nz, ny, nx = 40, 64, 72
    rng = np.random.default_rng(4)
    v = np.stack(O.ellipsoid_masks(nz, ny, nx)) ^ (rng.random((nz, ny, nx)) < 0.004)
    depths = np.full(nz, 0.3)
    vp, se = VoxelProcessor(), SurfaceExtractor()
    with contextlib.redirect_stdout(io.StringIO()):
        vol = vp.create_voxel_data(list(v), True, 0, nz, 0)
        h0 = dict(_memo.STATS)
        sms = [vp.smooth_voxel_data(vol, iterations=3, create_manifold=True) for _ in range(3)]
        assert _memo.STATS["hit"] == h0["hit"] + 2
        assert all(np.array_equal(s, sms[0]) for s in sms) and len({id(s) for s in sms}) == 3
        assert np.array_equal(sms[0], O.smooth(vol, 3, True))
        other = vp.smooth_voxel_data(vol, iterations=2, create_manifold=True)               # other parameters: computed
        assert np.array_equal(other, O.smooth(vol, 2, True))
        h1 = dict(_memo.STATS)
        meshes = [se.extract_manifold_surface(s, depths, 0.7, 0.9) for s in sms]              # three arrays, ONE device volume
        assert _memo.STATS["hit"] == h1["hit"] + 2
        ref = O.SurfaceExtractor().extract_manifold_surface(sms[0].copy(), depths, 0.7, 0.9)
        for m in meshes:
            assert m[0].tobytes() == ref[0].tobytes() and np.array_equal(m[1], ref[1])
        assert meshes[0][0] is not meshes[1][0] and meshes[0][0].flags.writeable
        meshes[0][0][:] = 0                                                                   # a caller scribbling on ITS copy ...
        again = se.extract_manifold_surface(sms[2], depths, 0.7, 0.9)
        assert again[0].tobytes() == ref[0].tobytes()                                         # ... changes nobody else's
        m2 = se.extract_manifold_surface(sms[0], depths, 0.7, 0.8)                            # other pixel size: computed
        ref2 = O.SurfaceExtractor().extract_manifold_surface(sms[0].copy(), depths, 0.7, 0.8)
        assert m2[0].tobytes() == ref2[0].tobytes()
        edited = sms[0].copy()
        edited[20, 30, 30:40] ^= True                                                         # another volume: computed
        m3 = se.extract_manifold_surface(edited, depths, 0.7, 0.9)
        ref3 = O.SurfaceExtractor().extract_manifold_surface(edited.copy(), depths, 0.7, 0.9)
        assert m3[0].tobytes() == ref3[0].tobytes() and np.array_equal(m3[1], ref3[1])
        empty = np.zeros((6, 8, 8), bool)
        assert se.extract_manifold_surface(empty, np.full(6, 1.0), 1.0, 1.0) is None
    _memo.clear()


def test_mesh_measures_check_their_indices_like_numpy(dev):
    """calculate_mesh_volume / calculate_surface_area index `vertices` with the caller's `faces` (surface_extractor.py:133-136,
    144-146): NumPy wraps negative indices and raises IndexError beyond the list; the device kernel must never read there."""
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    f = np.array([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]], np.int64)
    se, ose = SurfaceExtractor(), O.SurfaceExtractor()
    assert np.isclose(se.calculate_mesh_volume(v, f), ose.calculate_mesh_volume(v, f), rtol=1e-6)
    wrapped = f.copy(); wrapped[wrapped == 3] = -1                           # -1 is the last vertex, as in NumPy
    assert np.isclose(se.calculate_mesh_volume(v, wrapped), ose.calculate_mesh_volume(v, f), rtol=1e-6)
    assert np.isclose(float(se.calculate_surface_area(v, wrapped)), float(ose.calculate_surface_area(v, f)), rtol=1e-6)
    for bad in (4, 10 ** 9, -5):
        g = f.copy(); g[2, 1] = bad
        with pytest.raises(IndexError):
            se.calculate_mesh_volume(v, g)
        with pytest.raises(IndexError):
            se.calculate_surface_area(v, g)
    assert se.calculate_mesh_volume(v, np.zeros((0, 3), np.int64)) == 0.0


def test_fractional_threshold_rounds_up(dev, tmp_path):
    """image_loader.py:108 compares integer grey levels with `>= threshold`: for a threshold of 199.5 the level 199 is out and
    200 is in, on the host masks AND on the device copy the loader caches (pack_threshold truncated it in round 1)."""
    from PIL import Image
    rng = np.random.default_rng(9)
    grey = rng.integers(195, 205, (5, 16, 32), dtype=np.uint8)
    for k, sec in enumerate(("Section_0", "Section_1", "Section_2")):
        os.makedirs(tmp_path / sec)
    for z in range(5):
        Image.fromarray(grey[z], mode="L").save(tmp_path / "Section_1" / ("Mask_P_%d.png" % z))
    ld = ImageLoader()
    with contextlib.redirect_stdout(io.StringIO()):
        assert ld.load_mask_images(str(tmp_path), 199.5, [True, True, True])
        masks = ld.get_mask_images()
        assert np.array_equal(np.stack(masks), grey >= 199.5) and np.array_equal(np.stack(masks), grey >= 200)
        vol = VoxelProcessor().create_voxel_data(masks, False, 0, 5, 0)           # served from the loader's device copy
    assert np.array_equal(vol, grey >= 200)
    dv = pipeline.pack_threshold(torch.from_numpy(grey).to(dev), 199.5)
    assert np.array_equal(pipeline.unpack(dv).cpu().numpy(), grey >= 200)


def test_cfg1_literal_half_ellipsoid_stack(dev, tmp_path):
    """BASELINE configs[0] as written: "128x128x64 synthetic half-ellipsoid mask stack (simple_generator.py)".  One base mask
    -> 48 copies as the main body (Section_1) and the two flanks written by generate_slices_from_mask
    (/root/reference/simple_generator.py:6-20 -> ellipsoid_slice_generator.py:107-143; 8 slices each: Section_0 numbered
    downwards from the body's first slice, Section_2 upwards from its last) -> PNG folders -> ImageLoader -> create / smooth /
    extract on the HIP path, with the physical units of /root/reference/config.py:12-14 -- bytes-equal to the CPU oracle on
    the same loaded masks.  (The generator's parity with OpenCV is unpinned -- SURVEY 8f N3 -- the path from the PNG files
    on is not: the loader's masks equal the oracle's restatement of image_loader.py:37-120.)"""
    from PIL import Image
    from tomography_3d_reconstructor_amd.slice_generator import generate_slices_from_mask
    ny = nx = 128
    s0, s1, s2 = 8, 48, 8
    yy, xx = np.mgrid[0:ny, 0:nx]
    base = (((xx - 63.5) / 46.0) ** 2 + ((yy - 63.5) / 38.0) ** 2) <= 1.0
    body = tmp_path / "Section_1"
    body.mkdir()
    for k in range(1, s1 + 1):
        Image.fromarray(np.where(base, 255, 0).astype(np.uint8), mode="L").save(body / ("Mask_Patient_%d.png" % k))
    with contextlib.redirect_stdout(io.StringIO()) as said:
        generate_slices_from_mask(str(body / "Mask_Patient_1.png"), s0, str(tmp_path / "Section_0"), 1, False)
        generate_slices_from_mask(str(body / ("Mask_Patient_%d.png" % s1)), s2, str(tmp_path / "Section_2"), s1, True)
    assert said.getvalue().count("Generated %d slices" % (s0 + 2)) == 2
    assert len(os.listdir(tmp_path / "Section_0")) == s0 and len(os.listdir(tmp_path / "Section_2")) == s2
    exp, counts, files = O.load_masks(str(tmp_path), 200)
    ld = ImageLoader()
    with contextlib.redirect_stdout(io.StringIO()):
        assert ld.load_mask_images(str(tmp_path), 200, [True, True, True]) is True
    assert ld.get_side_counts() == counts == (s0, s1, s2) and ld.get_num_slices() == 64 and ld.get_image_dimensions() == (nx, ny)
    masks = ld.get_mask_images()
    assert all(np.array_equal(m, e) for m, e in zip(masks, exp))
    areas = [int(m.sum()) for m in masks]
    assert areas[:s0] == sorted(areas[:s0]) and areas[s0 + s1:] == sorted(areas[s0 + s1:], reverse=True)     # a half-ellipsoid at either end
    assert 0 < areas[0] < areas[s0] == int(base.sum()) and areas[-1] < areas[s0]
    mm_x, mm_y, depth = 143.1 / nx, 95.03 / ny, 6.0
    vp, se, ovp, ose = VoxelProcessor(), SurfaceExtractor(), O.VoxelProcessor(), O.SurfaceExtractor()
    with contextlib.redirect_stdout(io.StringIO()):
        vol = vp.create_voxel_data(masks, True, s0, s1, s2)
        depths = vp.calculate_slice_depths(depth)
        sm = vp.smooth_voxel_data(vol, iterations=3, create_manifold=True)
        res = se.extract_manifold_surface(sm, depths, mm_y, mm_x, smooth=True, manifold=True, add_padding=True)
        ovol = ovp.create_voxel_data(exp, True, s0, s1, s2)
        odepths = ovp.calculate_slice_depths(depth)
        osm = ovp.smooth_voxel_data(ovol, 3, True)
        ores = ose.extract_manifold_surface(osm, odepths, mm_y, mm_x)
    assert np.array_equal(vol, ovol) and np.array_equal(sm, osm) and depths.tobytes() == odepths.tobytes()
    assert res is not None and ores is not None
    assert res[0].dtype == np.float32 and res[0].tobytes() == ores[0].tobytes(), "vertices differ from the oracle"
    assert res[1].dtype == np.int64 and np.array_equal(res[1], ores[1]), "faces differ from the oracle"
    assert len(res[0]) > 10000 and abs(se.calculate_mesh_volume(*res) - ose.calculate_mesh_volume(*ores)) <= 1e-6 * ose.calculate_mesh_volume(*ores)


def test_results_are_checksummed_while_they_download_and_verified_next_to_the_work(dev, monkeypatch):
    """Round 4 (writeable results by default): the checksum a result needs is taken piece by piece WHILE it downloads
    (voxel_processor._download_with_digest) and the verification of an array that comes back runs on a helper thread next to the
    device work (with_device_volume): same digests as the one-shot checksum, an edited array is still never served from the cache."""
    from tomography_3d_reconstructor_amd import _devcache, pipeline
    from tomography_3d_reconstructor_amd import voxel_processor as VP
    monkeypatch.setattr(_devcache, "WRITEABLE_RESULTS", True)
    monkeypatch.setattr(VP, "BIG", 1 << 20)                  # the volume-sized path for a 5 MiB volume, in pieces of 1 MiB
    monkeypatch.setattr(VP, "PIECE", 1 << 20)
    _devcache.clear()
    rng = np.random.default_rng(8)
    nz, ny, nx = 37, 300, 480                                 # 5 328 000 bytes: not a whole number of pieces
    v = np.stack(O.ellipsoid_masks(nz, ny, nx)) ^ (rng.random((nz, ny, nx)) < 0.004)
    vp, se = VP.VoxelProcessor(), SurfaceExtractor()
    created = vp.create_voxel_data([m.copy() for m in v], True, 0, nz, 0)
    ent = _devcache._cache[id(created)]
    assert created.flags.writeable and ent[2] == _devcache.checksum(created)          # the pipelined digest IS the checksum
    assert np.array_equal(created, O.close_ends(v))
    before = dict(_devcache.STATS)
    sm = vp.smooth_voxel_data(created)                                                 # cached copy, verified next to the work
    assert _devcache.STATS["hit_verified"] == before["hit_verified"] + 1 and np.array_equal(sm, O.smooth(created, 3, True))
    created[20, 150, 200:260] ^= True                                                  # edited: the speculative result is thrown away
    before = dict(_devcache.STATS)
    sm2 = vp.smooth_voxel_data(created)
    assert _devcache.STATS["miss_edited"] == before["miss_edited"] + 1 and _devcache.STATS["hit_verified"] == before["hit_verified"]
    assert np.array_equal(sm2, O.smooth(created, 3, True))
    depths = np.full(nz, 0.5)
    sm2[:, :, 1::32] ^= True
    ref = O.SurfaceExtractor().extract_manifold_surface(sm2.copy(), depths, 0.7, 0.9)
    got = se.extract_manifold_surface(sm2, depths, 0.7, 0.9)
    assert got[0].tobytes() == ref[0].tobytes() and np.array_equal(got[1], ref[1])
    _devcache.clear()
