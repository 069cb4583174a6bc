"""GPU tier (-m gpu): the HIP path, called through the C ABI, against the pinned oracle and the golden
fixtures generated from the reference.  Bar: bit-exact for the boolean volumes, the float32 field, the
vertex positions, the face list (order, winding, indices) -- stricter than north_star's 1e-6 on positions.
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tomography_3d_reconstructor_amd import _lib, pipeline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from tomography_3d_reconstructor_amd import SurfaceExtractor, VoxelProcessor

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def unpack(bits, shape):
    shape = tuple(int(s) for s in shape)
    return np.unpackbits(bits)[: int(np.prod(shape))].reshape(shape).astype(bool)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def to_vol(arr, dev):
    return pipeline.pack(torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8)).to(dev))


def to_np(vol):
    return pipeline.unpack(vol).cpu().numpy()


# ------------------------------------------------------------------ bits
@pytest.mark.parametrize("shape", [(3, 5, 7), (2, 9, 64), (4, 3, 65), (5, 17, 130), (2, 2, 1024), (1, 1, 1), (3, 4, 1100)])
def test_pack_unpack_popcount(dev, shape):
    rng = np.random.default_rng(1)
    a = rng.random(shape) < 0.4
    vol = to_vol(a, dev)
    ref_bits = np.packbits(a, axis=2, bitorder="little")
    got = vol.bits.cpu().numpy().view(np.uint8).reshape(shape[0], shape[1], -1)[:, :, : ref_bits.shape[2]]
    assert np.array_equal(got, ref_bits)
    assert np.array_equal(to_np(vol), a)
    assert int(pipeline.popcount_async(vol).item()) == int(a.sum())


@pytest.mark.parametrize("i", range(6))
def test_binary_stages_golden(dev, i):
    d = np.load(os.path.join(G, "binary_stages.npz"))
    shape = tuple(int(s) for s in d["s%d_shape" % i])
    v = unpack(d["s%d_in" % i], shape)
    vol = to_vol(v, dev)
    created = pipeline.close_ends(vol)
    assert np.array_equal(to_np(created), unpack(d["s%d_created" % i], shape))
    assert int(pipeline.popcount_async(created).item()) == int(d["s%d_active" % i])
    assert np.array_equal(to_np(vol), v), "input must not be mutated"
    for it, cm in [(3, True), (1, False), (0, True)]:
        sm = pipeline.smooth(vol, it, cm)
        assert np.array_equal(to_np(sm), unpack(d["s%d_smooth_%d_%d" % (i, it, int(cm))], shape)), (it, cm)
    # fill holes of every slice
    L = _lib.lib()
    work = pipeline.BitVolume(vol.bits.clone(), vol.shape)
    scratch = torch.empty(shape[1] * work.bits.shape[2] + 8, dtype=torch.int64, device=dev)
    for z in range(shape[0]):
        _lib.check(L.tomo_fill_holes_slice(work.bits.data_ptr(), shape[0], shape[1], shape[2], z, scratch.data_ptr(),
                                           None), "fill")
    assert np.array_equal(to_np(work), unpack(d["s%d_fill" % i], shape))


@pytest.mark.parametrize("shape,p", [((40, 30, 200), 0.5), ((130, 9, 70), 0.3), ((67, 33, 65), 0.6)])
def test_binary_stages_vs_oracle(dev, shape, p):
    rng = np.random.default_rng(5)
    v = rng.random(shape) < p
    v[0, 2:20, 3:40] = True
    v[0, 5:9, 6:30] = False
    vol = to_vol(v, dev)
    assert np.array_equal(to_np(pipeline.close_ends(vol)), O.close_ends(v))
    assert np.array_equal(to_np(pipeline.smooth(vol, 2, True)), O.smooth(v, 2, True))


@pytest.mark.parametrize("shape", [(20, 130, 257), (9, 70, 262), (37, 57, 320), (12, 120, 513), (70, 9, 64), (5, 200, 1025),
                                   (11, 64, 255), (3, 1, 700), (1, 90, 129),
                                   # even word counts take the LDS-staged path: last strip of 2 words, tail in it / not
                                   (20, 130, 384), (12, 120, 1000), (37, 57, 1090), (6, 300, 128), (50, 61, 1152)])
@pytest.mark.parametrize("it,cm", [(3, True), (1, True), (2, False), (3, False), (0, True)])
def test_smooth_tile_edges_vs_oracle(dev, shape, it, cm):
    """The one-wave-per-tile smoothing kernel: rows wider than a 4-word strip with the tail in every position relative
    to the strip and its 8-bit halo, several 56-row tiles, slices across z chunks, 2 / 4 / 6 / 8 passes."""
    rng = np.random.default_rng(shape[2] + it)
    v = rng.random(shape) < 0.82
    v[:, : shape[1] // 3, -(shape[2] // 5 + 1):] = True          # solid block against the right border and the tail word
    v[shape[0] // 2:, shape[1] // 2:, : shape[2] // 7 + 1] = rng.random((shape[0] - shape[0] // 2, shape[1] - shape[1] // 2,
                                                                       shape[2] // 7 + 1)) < 0.5
    got = to_np(pipeline.smooth(to_vol(v, dev), it, cm))
    assert np.array_equal(got, O.smooth(v, it, cm))


@pytest.mark.parametrize("path", ["generic", "direct"])
def test_smooth_alternative_paths(path):
    """The smoothing launcher picks the LDS-staged kernel when it applies; the direct and the generic (run-time pass
    mask) kernels behind it are forced through TOMO_MORPH_PATH, which is read once per process -- hence the child."""
    import subprocess, sys, os, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import numpy as np, torch
        from tomography_3d_reconstructor_amd import pipeline
        from oracle import oracle as O
        dev = torch.device("cuda:0")
        for shape in [(20, 130, 384), (37, 57, 1090), (9, 70, 262), (66, 64, 256)]:
            rng = np.random.default_rng(shape[2])
            v = rng.random(shape) < 0.8
            v[:, : shape[1] // 3, -(shape[2] // 5 + 1):] = True
            vol = pipeline.pack(torch.from_numpy(v.view(np.uint8)).to(dev))
            for it, cm in [(3, True), (2, False)]:
                got = pipeline.unpack(pipeline.smooth(vol, it, cm)).cpu().numpy().astype(bool)
                assert np.array_equal(got, O.smooth(v, it, cm)), (shape, it, cm)
        print("ok")
    """)
    env = dict(os.environ, TOMO_MORPH_PATH=path, PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_fill_holes_spiral(dev):
    # a long corridor: the flood has to travel far (many iterations of the device loop)
    n = 96
    img = np.zeros((n, n), bool)
    img[2:-2, 2:-2] = True
    y = x = 4
    for k in range(4, n // 2, 4):   # carve a spiral of background inside the solid square
        img[k, k:n - k] = False
        img[k:n - k, n - k - 1] = False
        img[n - k - 1, k + 2:n - k] = False
        img[k + 4:n - k, k + 2] = False
    img[4, 0:6] = False             # connect the spiral to the outside
    v = np.stack([img, img, img])
    vol = to_vol(v, dev)
    L = _lib.lib()
    scratch = torch.empty(n * vol.bits.shape[2] + 8, dtype=torch.int64, device=dev)
    _lib.check(L.tomo_fill_holes_slice(vol.bits.data_ptr(), 3, n, n, 1, scratch.data_ptr(), None), "fill")
    got = to_np(vol)
    assert np.array_equal(got[1], O.fill_holes_2d(img))
    assert np.array_equal(got[0], img) and np.array_equal(got[2], img)


@pytest.mark.parametrize("shape", [(3, 5, 16), (4, 9, 48), (33, 7, 64), (34, 6, 1040), (35, 3, 2064), (66, 11, 80), (70, 40, 128),
                                   (2, 8, 32), (1, 8, 32), (5, 6, 20), (40, 12, 1024),
                                   # from 128 inner slices on: the kernel that hands the boundary words over inside the workgroup
                                   # (round 4) -- every remainder of (nz - 2) mod 4, one to three workgroups per column, wide rows
                                   (130, 5, 48), (131, 4, 64), (132, 3, 80), (133, 5, 32), (200, 3, 2064), (263, 4, 1040), (397, 2, 96)])
def test_pack_closed_one_pass_vs_oracle(dev, shape, monkeypatch):
    """pipeline.pack_closed (end slices packed + filled, then the fused pack + three-tap stencil kernel) == the oracle's
    _close_volume_ends recurrence == the separate pack / fill / carry-chain kernels, incl. runs that end at the last slice,
    rows wider than one 1024-voxel group, and the layouts that take the fallback (nz < 3, nx % 16 != 0)."""
    rng = np.random.default_rng(sum(shape))
    for density in (0.3, 0.7):
        v = rng.random(shape) < density
        if shape[0] > 4:
            v[1] = v[0] & v[2]                                  # single-voxel gaps along z: what the recurrence closes
            v[shape[0] // 2] = False
        mask = torch.from_numpy(v.view(np.uint8)).to(dev)
        got = to_np(pipeline.pack_closed(mask))
        assert np.array_equal(got, O.close_ends(v)), density
        with monkeypatch.context() as m:
            m.setattr(pipeline, "PACK_CLOSE_FUSED", False)
            assert np.array_equal(to_np(pipeline.pack_closed(mask)), got)
        assert np.array_equal(to_np(pipeline.pack(mask)), v)    # the mask itself is untouched


def _fill_cases(ny, nx, rng):
    yy, xx = np.mgrid[0:ny, 0:nx]
    r2 = ((xx - (nx - 1) / 2) / (0.42 * nx)) ** 2 + ((yy - (ny - 1) / 2) / (0.40 * ny)) ** 2
    spiral = np.zeros((ny, nx), bool)
    spiral[2:-2, 2:-2] = True
    for k in range(4, min(ny, nx) // 2, 4):                # a corridor of background winding into a solid block
        spiral[k, k:nx - k] = False
        spiral[k:ny - k, nx - k - 1] = False
        spiral[ny - k - 1, k + 2:nx - k] = False
        spiral[k + 4:ny - k, k + 2] = False
    spiral[4, 0:6] = False
    comb = np.zeros((ny, nx), bool)                        # vertical teeth: floods travel straight through many bands
    comb[3:-3, 3:-3:5] = True
    comb[ny // 2, 3:-3] = True
    return {"ellipse": r2 <= 1.0, "ring": (r2 <= 1.0) & (r2 >= 0.4), "noise50": rng.random((ny, nx)) < 0.5,
            "noise62": rng.random((ny, nx)) < 0.38, "spiral": spiral, "comb": comb, "empty": np.zeros((ny, nx), bool),
            "full": np.ones((ny, nx), bool), "frame": ~np.pad(np.zeros((ny - 2, nx - 2), bool), 1, constant_values=True) ^ True}


@pytest.mark.parametrize("shape", [(200, 300), (257, 130), (96, 1000), (513, 64), (1024, 1024)])
def test_fill_holes_many_workgroups_vs_oracle(dev, shape, monkeypatch):
    """tomo_fill_holes_ends / _slice on slices large enough for the band kernel (row bands in LDS, grid barrier, carry
    fold across bands): == ndimage.binary_fill_holes as restated by the oracle, for both end slices at once with
    DIFFERENT content."""
    ny, nx = shape
    rng = np.random.default_rng(ny + nx)
    cases = _fill_cases(ny, nx, rng)
    names = list(cases)
    L = _lib.lib()
    for i, name in enumerate(names):
        a, b = cases[name], cases[names[(i + 3) % len(names)]]
        mid = rng.random((ny, nx)) < 0.5
        v = np.stack([a, mid, b])
        exp = v.copy()
        for z in (0, 2):
            if exp[z].any():
                exp[z] = O.fill_holes_2d(exp[z])
        vol = to_vol(v, dev)
        scratch = torch.full((ny * vol.bits.shape[2] + 8,), -1, dtype=torch.int64, device=dev)   # garbage: the launch must not rely on zeros
        _lib.check(L.tomo_fill_holes_ends(vol.bits.data_ptr(), 3, ny, nx, scratch.data_ptr(), None), "fill")
        got = to_np(vol)
        assert np.array_equal(got, exp), (name, "ends")
        ctrl = scratch[:16].cpu().numpy()
        assert ctrl[2] == 0 and ctrl[10] == 0, "a grid barrier of the band kernel was abandoned"
        vol1 = to_vol(v, dev)
        _lib.check(L.tomo_fill_holes_slice(vol1.bits.data_ptr(), 3, ny, nx, 0, scratch.data_ptr(), None), "fill")
        assert np.array_equal(to_np(vol1)[0], exp[0]) and np.array_equal(to_np(vol1)[2], v[2]), (name, "slice")


def test_fill_holes_abandoned_grid_barrier_is_redone(dev):
    """The band kernel's grid barrier gives up when its workgroups are not all resident in time (a shared GPU); the slice is
    then redone by the one-workgroup kernel that runs behind it.  TOMO_FILL_SPIN_LIMIT=0 (read once per process, hence the
    child process) makes EVERY workgroup but the last one to arrive abandon the first barrier: the result must still be
    ndimage.binary_fill_holes, for both end slices."""
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np, torch
from oracle import oracle as O
from tomography_3d_reconstructor_amd import _lib, pipeline
dev = torch.device("cuda:0")
L = _lib.lib()
rng = np.random.default_rng(5)
for ny, nx in ((256, 512), (1024, 1024), (300, 200)):
    yy, xx = np.mgrid[0:ny, 0:nx]
    ring = (np.hypot(yy - ny / 2, xx - nx / 2) < 0.4 * min(ny, nx)) & (np.hypot(yy - ny / 2, xx - nx / 2) > 0.3 * min(ny, nx))
    noise = rng.random((ny, nx)) < 0.62
    v = np.stack([ring, rng.random((ny, nx)) < 0.5, noise])
    vol = pipeline.pack(torch.from_numpy(v.view(np.uint8)).to(dev))
    scratch = torch.full((ny * vol.bits.shape[2] + 8,), -1, dtype=torch.int64, device=dev)
    _lib.check(L.tomo_fill_holes_ends(vol.bits.data_ptr(), 3, ny, nx, scratch.data_ptr(), None), "fill")
    got = pipeline.unpack(vol).cpu().numpy()
    assert np.array_equal(got[0], O.fill_holes_2d(v[0])) and np.array_equal(got[2], O.fill_holes_2d(v[2])), (ny, nx)
    assert np.array_equal(got[1], v[1])
print("REDONE OK")
""" % ROOT
    env = dict(os.environ, TOMO_FILL_SPIN_LIMIT="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "REDONE OK" in out.stdout, out.stderr[-3000:]


# ------------------------------------------------------------------ field
@pytest.mark.parametrize("shape", [(5, 6, 7), (4, 9, 64), (6, 5, 150), (3, 12, 255), (7, 8, 256), (5, 40, 260),
                                   (2, 3, 1), (1, 1, 5), (9, 7, 1030)])
@pytest.mark.parametrize("pad", [True, False])
def test_field_vs_oracle(dev, shape, pad):
    rng = np.random.default_rng(shape[2] + pad)
    v = rng.random(shape) < 0.5
    f = pipeline.make_field(to_vol(v, dev), True, pad)
    got = f.dense().cpu().numpy()
    exp = O.field(v, True, pad)
    assert got.shape == exp.shape
    assert got.tobytes() == exp.tobytes()


@pytest.mark.parametrize("shape", [(6, 40, 2100), (5, 12, 4200), (4, 9, 4096), (3, 7, 8200)])
def test_field_wide_rows_vs_oracle(dev, shape):
    """rows wider than one 256-thread block (576 / 1024 threads) and wider than one block (multi-block rows)"""
    rng = np.random.default_rng(shape[2])
    v = rng.random(shape) < 0.5
    v[:, :, 1000:1900] = True          # a constant stretch: const-wave path next to general waves
    v[:, :, 2500:3000] = False
    for pad in (True, False):
        f = pipeline.make_field(to_vol(v, dev), True, pad)
        assert f.dense().cpu().numpy().tobytes() == O.field(v, True, pad).tobytes()


def test_whole_path_wide_noise_vs_oracle(dev):
    rng = np.random.default_rng(21)
    v = rng.random((14, 30, 2300)) < 0.45
    v[:, 10:20, 500:1500] = True
    depths = np.linspace(0.2, 0.9, 14)
    vol = pipeline.smooth(pipeline.close_ends(to_vol(v, dev)), 2, True)
    osm = O.smooth(O.close_ends(v), 2, True)
    assert np.array_equal(to_np(vol), osm)
    got = pipeline.extract_surface(vol, depths, 0.3, 1.7)
    ev, ef = O.SurfaceExtractor().extract_manifold_surface(osm, depths, 0.3, 1.7)
    assert got[0].cpu().numpy().tobytes() == ev.tobytes() and np.array_equal(got[1].cpu().numpy(), ef)


@pytest.mark.parametrize("shape", [(2, 2, 2), (2, 3, 5), (3, 2, 70), (1, 9, 9), (9, 1, 9), (9, 9, 1), (5, 5, 257)])
@pytest.mark.parametrize("pad", [True, False])
def test_whole_path_tiny_and_degenerate_shapes(dev, shape, pad):
    rng = np.random.default_rng(sum(shape))
    v = rng.random(shape) < 0.6
    depths = np.full(shape[0], 0.5)
    vol = to_vol(v, dev)
    got = pipeline.extract_surface(vol, depths, 1.0, 2.0, True, pad)
    exp = O.SurfaceExtractor().extract_manifold_surface(v, depths, 1.0, 2.0, True, True, pad)
    if exp is None:
        assert got is None
    else:
        assert got is not None
        ev, ef = exp
        gf = got[1].cpu().numpy()
        assert got[0].cpu().numpy().tobytes() == ev.tobytes()
        assert (gf.shape[0] == 0 and len(ef) == 0) or np.array_equal(gf, ef)


def test_field_uniform_fast_paths(dev):
    # all-ones interior, all-zero exterior and a sharp boundary in one volume
    v = np.zeros((40, 70, 600), bool)
    v[4:36, 5:65, 10:590] = True
    for pad in (True, False):
        f = pipeline.make_field(to_vol(v, dev), True, pad)
        assert f.dense().cpu().numpy().tobytes() == O.field(v, True, pad).tobytes()
    f = pipeline.make_field(to_vol(v, dev), False, False)
    assert np.array_equal(f.dense().cpu().numpy(), v.astype(np.float32))


# ------------------------------------------------------------------ marching cubes
def raw_mesh_as_triangles(mesh):
    vkey = mesh.vkey.cpu().numpy()
    vpos = mesh.vpos.cpu().numpy()
    idx = mesh.faces32.cpu().numpy()
    assert np.all(np.diff(vkey) > 0), "vertex keys must be strictly ascending"
    assert int(mesh._stats[7].item()) == 0
    assert idx.min() >= 0 and idx.max() < len(vpos)
    return vpos[idx], vpos


@pytest.mark.parametrize("name", ["uniform", "lattice", "smooth", "binfield"])
def test_mc_noise_golden(dev, name):
    d = np.load(os.path.join(G, "mc_noise.npz"))
    vol = d[name + "_vol"]
    mesh = pipeline.marching_cubes(pipeline.field_from_dense(torch.from_numpy(vol).to(dev)), 0.5)
    tri, vpos = raw_mesh_as_triangles(mesh)
    ev, ef = d[name + "_verts"], d[name + "_faces"]
    assert vpos.shape == ev.shape and tri.shape[0] == ef.shape[0]
    assert tri.tobytes() == ev[ef].tobytes()          # same triangles, same order, same winding, same bits
    assert sorted(map(bytes, vpos)) == sorted(map(bytes, ev))


def test_mc_matches_oracle_on_wide_noise(dev):
    rng = np.random.default_rng(11)
    vol = rng.random((7, 9, 530)).astype(np.float32)
    mesh = pipeline.marching_cubes(pipeline.field_from_dense(torch.from_numpy(vol).to(dev)), 0.5)
    tri, vpos = raw_mesh_as_triangles(mesh)
    ev, ef = O.marching_cubes(vol, 0.5)
    assert tri.tobytes() == ev[ef].tobytes() and len(vpos) == len(ev)


def test_first_touch_numbering_on_noise(dev):
    """manifold=False: skimage's own vertex numbering (first touch in the serial scan), all MC33 cases"""
    d = np.load(os.path.join(G, "mc_noise.npz"))
    for name in ("uniform", "lattice", "binfield"):
        vol = d[name + "_vol"]
        mesh = pipeline.marching_cubes(pipeline.field_from_dense(torch.from_numpy(vol).to(dev)), 0.5)
        v, f = pipeline.first_touch_order(mesh)
        assert v.cpu().numpy().tobytes() == d[name + "_verts"].tobytes()
        assert np.array_equal(f.cpu().numpy(), d[name + "_faces"])


def test_mc_none_cases(dev):
    z = torch.zeros((4, 5, 6), device=dev)
    assert pipeline.marching_cubes(pipeline.field_from_dense(z), 0.5) is None
    assert pipeline.marching_cubes(pipeline.field_from_dense(z + 1), 0.5) is None
    assert pipeline.marching_cubes(pipeline.field_from_dense(z + 0.5), 0.5) is None   # == iso counts as outside


# ------------------------------------------------------------------ whole path
CASES = ["a_blobs", "b_noise", "c_noise_nomanifold_smooth", "d_nopad", "e_nomanifold", "f_noclose", "g_empty", "h_full",
         "i_noise_raw", "j_wide"]


def load_case(cn):
    d = np.load(os.path.join(G, "pipeline_small.npz"))
    return {k[len(cn) + 2:]: d[k] for k in d.files if k.startswith(cn + "__")}


@pytest.mark.parametrize("cn", CASES)
def test_pipeline_small_golden(dev, cn, capsys):
    c = load_case(cn)
    shape = c["shape"]
    masks = list(unpack(c["masks"], shape))
    vp, se = VoxelProcessor(), SurfaceExtractor()
    created = vp.create_voxel_data(masks, bool(c["close_ends"]), *[int(s) for s in c["sides"]])
    assert isinstance(created, np.ndarray) and created.dtype == np.bool_
    assert np.array_equal(created, unpack(c["created"], shape))
    assert "Voxels: %s, active: %s" % (created.shape, format(int(created.sum()), ",")) in capsys.readouterr().out
    depths = vp.calculate_slice_depths(float(c["total_depth"]))
    assert depths.tobytes() == c["depths"].tobytes()
    if c["smooth_first"]:
        sm = vp.smooth_voxel_data(created, iterations=int(c["iterations"]), create_manifold=bool(c["create_manifold"]))
    else:
        sm = created
    assert np.array_equal(sm, unpack(c["smoothed"], shape))
    res = se.extract_manifold_surface(sm, depths, float(c["mm_y"]), float(c["mm_x"]), smooth=True,
                                      manifold=bool(c["manifold"]), add_padding=bool(c["add_padding"]))
    if c["is_none"]:
        assert res is None
        return
    v, f = res
    assert v.dtype == np.float32 and str(f.dtype) == str(c["faces_dtype"]) and v.flags.c_contiguous
    assert v.shape == c["verts"].shape and f.shape == c["faces"].shape
    assert np.array_equal(f, c["faces"])
    assert v.tobytes() == c["verts"].tobytes()
    assert "Surface: %d vertices, %d faces" % (len(v), len(f)) in capsys.readouterr().out
    assert np.isclose(se.calculate_mesh_volume(v, f), float(c["mesh_volume"]), rtol=1e-6)
    assert np.isclose(se.calculate_surface_area(v, f), float(c["surface_area"]), rtol=1e-6)
    pc = vp.generate_point_cloud(sm, float(c["mm_x"]), float(c["mm_y"]), depths, 3)
    assert np.array_equal(pc, c["point_cloud"])


def test_api_errors(dev):
    with pytest.raises(ValueError, match="Load masks first, hmm."):
        VoxelProcessor().create_voxel_data([])
    vp = VoxelProcessor()
    assert vp.voxel_data is None and vp.side_1_count == 0
    assert len(vp.calculate_slice_depths(3.0)) == 0


def test_inputs_not_mutated_and_cache_detects_edits(dev, monkeypatch):
    """The reference reads the array it is handed (voxel_processor.py:84, surface_extractor.py:43-46): the cached device
    copy of a returned volume must never stand in for an array that was edited -- at ANY offset of a >= 1 MiB volume."""
    from tomography_3d_reconstructor_amd import _devcache
    from tomography_3d_reconstructor_amd.volume_calculator import VolumeCalculator
    rng = np.random.default_rng(2)
    nz, ny, nx = 128, 128, 136
    v = np.stack(O.ellipsoid_masks(nz, ny, nx)) ^ (rng.random((nz, ny, nx)) < 0.003)
    masks = [m.copy() for m in v]
    depths = np.full(nz, 0.5)
    for writeable_results in (False, True):
        monkeypatch.setattr(_devcache, "WRITEABLE_RESULTS", writeable_results)
        _devcache.clear()
        vp, se, vc = VoxelProcessor(), SurfaceExtractor(), VolumeCalculator()
        out = vp.create_voxel_data(masks, True, 0, nz, 0)
        assert all(np.array_equal(a, b) for a, b in zip(masks, v))
        assert out.flags.writeable == writeable_results
        s1 = vp.smooth_voxel_data(out)                                  # served from the cached device copy
        assert np.array_equal(s1, O.smooth(out, 3, True))
        if not writeable_results:
            with pytest.raises(ValueError):
                out[64, 64, 65] = False                                  # hand-outs are write-protected ...
            out = out.copy()                                             # ... editing takes a copy (a new object: never cached)
        hits = dict(_devcache.STATS)
        out[64, 64, 65] = not out[64, 64, 65]                            # ONE voxel, at an offset a sampled checksum skips
        edited = out.copy()
        s2 = vp.smooth_voxel_data(out)
        assert np.array_equal(s2, O.smooth(edited, 3, True))
        assert _devcache.STATS["hit_readonly"] == hits["hit_readonly"] and _devcache.STATS["hit_verified"] == hits["hit_verified"]
        s2w = s2 if writeable_results else s2.copy()
        s2w[:, :, 1::32] ^= True                                         # strided edit of the smoothed volume
        ref = O.SurfaceExtractor().extract_manifold_surface(s2w.copy(), depths, 0.7, 0.9)
        got = se.extract_manifold_surface(s2w, depths, 0.7, 0.9)
        assert got[0].tobytes() == ref[0].tobytes() and np.array_equal(got[1], ref[1])
        s2w[3, 100:110, 5:9] = True                                      # edit again; the volume calculator must see it too
        assert vc.calculate_voxel_volume_variable_depth(s2w, 0.9, 0.7, depths) == \
            O.VolumeCalculator().calculate_voxel_volume_variable_depth(s2w.copy(), 0.9, 0.7, depths)
        # an untouched volume is still served from the cache (verified in full when it is writeable)
        s3 = vp.smooth_voxel_data(edited.copy())
        before = dict(_devcache.STATS)
        se.extract_manifold_surface(s3, depths, 0.7, 0.9)
        key = "hit_verified" if writeable_results else "hit_readonly"
        assert _devcache.STATS[key] == before[key] + 1
    _devcache.clear()


def test_device_failures_return_none_like_the_reference(dev, monkeypatch, capsys):
    """surface_extractor.py:42,74-75: ANY exception inside extract_manifold_surface -> None (the orchestrator then falls
    back to the point cloud).  Every guard of the device path is forced in turn; only a missing GPU / library raises."""
    nz, ny, nx = 24, 40, 48
    v = np.stack(O.ellipsoid_masks(nz, ny, nx))
    depths = np.full(nz, 0.5)
    se, vp = SurfaceExtractor(), VoxelProcessor()
    good = se.extract_manifold_surface(v, depths, 1.0, 1.0)
    assert good is not None
    with monkeypatch.context() as m:                     # (1) active-voxel list beyond the 32-bit index range
        m.setattr(pipeline, "LIST_LIMIT", 16)
        pipeline._NA_HINT.clear()
        pipeline._MC3_HINT.clear()
        assert se.extract_manifold_surface(v, depths, 1.0, 1.0) is None
    with monkeypatch.context() as m:                     # (2) vertex / triangle totals beyond it
        m.setattr(pipeline, "MESH_LIMIT", 16)
        assert se.extract_manifold_surface(v, depths, 1.0, 1.0) is None
    with monkeypatch.context() as m:                     # (3) the internal consistency check of the unique / remap stage

        def boom(*a, **k):
            raise _lib.TomoError("internal error: 3 triangle corners reference a missing vertex")
        m.setattr(pipeline, "mc3_vertices" if pipeline.MC3 else "ensure_manifold_mesh", boom)
        assert se.extract_manifold_surface(v, depths, 1.0, 1.0) is None
    with monkeypatch.context() as m:                     # (4) a failing launch (status from the C ABI)
        m.setattr(pipeline._lib, "check", lambda code, what: (_ for _ in ()).throw(_lib.TomoError(what + " failed")))
        assert se.extract_manifold_surface(v.copy(), depths, 1.0, 1.0) is None
    err = capsys.readouterr().err
    assert err.count("surface extraction failed") == 4 and "32-bit indices" in err
    # input the device path cannot take at all: the reference's own except clause would swallow its error as well
    assert se.extract_manifold_surface(np.zeros((4, 4)), depths, 1.0, 1.0) is None
    # smooth_voxel_data: a failure of the full sequence falls back to the closings alone (voxel_processor.py:93-95)
    calls = []
    real = pipeline.smooth

    def flaky(vol, iterations, create_manifold):
        calls.append(create_manifold)
        if create_manifold:
            raise _lib.TomoError("forced")
        return real(vol, iterations, create_manifold)
    with monkeypatch.context() as m:
        m.setattr(pipeline, "smooth", flaky)
        out = vp.smooth_voxel_data(v, 3, True)
    assert calls == [True, False] and np.array_equal(out, O.smooth(v, 3, False))
    # and the unavailable-GPU case still raises: there is no CPU path to fall back to
    with monkeypatch.context() as m:
        m.setattr(torch.cuda, "is_available", lambda: False)
        with pytest.raises(_lib.TomoUnavailable):
            se.extract_manifold_surface(v.copy(), depths, 1.0, 1.0)
        with pytest.raises(_lib.TomoUnavailable):
            vp.smooth_voxel_data(v.copy(), 3, True)
    again = se.extract_manifold_surface(v, depths, 1.0, 1.0)
    assert again[0].tobytes() == good[0].tobytes() and np.array_equal(again[1], good[1])


def test_ellipsoid_cfg1_full_mesh(dev):
    c = np.load(os.path.join(G, "ellipsoid_64x128x128.npz"))
    nz, ny, nx = [int(s) for s in c["shape"]]
    mask = pipeline.ellipsoid_mask(nz, ny, nx, dev)
    assert sha(np.packbits(mask.cpu().numpy())) == str(c["mask_sha"])
    vol = pipeline.smooth(pipeline.close_ends(pipeline.pack(mask)), 3, True)
    f = pipeline.make_field(vol)
    assert sha(f.dense().cpu().numpy()) == str(c["field_sha"])
    v, fc = pipeline.extract_surface(vol, c["depths"], float(c["mm_y"]), float(c["mm_x"]))
    assert tuple(v.shape) == (36320, 3) and tuple(fc.shape) == (72636, 3)
    assert np.array_equal(fc.cpu().numpy(), c["faces"]) and v.cpu().numpy().tobytes() == c["verts"].tobytes()
    vol_mm3, area = pipeline.mesh_volume_area(v, fc)
    assert np.isclose(vol_mm3, float(c["mesh_volume"]), rtol=1e-6)
    assert np.isclose(area, float(c["surface_area"]), rtol=1e-6)


def run_ellipsoid(dev, nz, ny, nx):
    mask = pipeline.ellipsoid_mask(nz, ny, nx, dev)
    vol0 = pipeline.pack(mask)
    del mask
    created = pipeline.close_ends(vol0)
    active = int(pipeline.popcount_async(created).item())
    sm = pipeline.smooth(created, 3, True)
    depths = np.full(nz, float(nz) / nz)
    v, f = pipeline.extract_surface(sm, depths, 1.0, 1.0)
    return vol0, created, active, sm, v, f


@pytest.mark.parametrize("key", ["64x128x128", "96x80x112", "256x256x256", "512x512x512", "1024x1024x1024"])
def test_ellipsoid_hashes(dev, key):
    h = json.load(open(os.path.join(G, "ellipsoid_hashes.json")))
    if key not in h:
        pytest.skip("no golden hash for " + key)
    h = h[key]
    nz, ny, nx = h["shape"]
    vol0, created, active, sm, v, f = run_ellipsoid(dev, nz, ny, nx)
    assert active == h["active"]
    if nz <= 512:
        assert sha(np.packbits(to_np(vol0))) == h["mask_sha256"]
        assert sha(np.packbits(to_np(created))) == h["created_sha256"]
        assert sha(np.packbits(to_np(sm))) == h["smoothed_sha256"]
    if nz <= 256:
        assert sha(pipeline.make_field(sm).dense().cpu().numpy()) == h["field_f32_sha256"]
    assert (v.shape[0], f.shape[0]) == (h["n_vertices"], h["n_faces"])
    assert sha(v.cpu().numpy()) == h["vertices_f32_sha256"]
    assert sha(f.cpu().numpy()) == h["faces_i64_sha256"]
    _, area = pipeline.mesh_volume_area(v, f)
    assert np.isclose(area, h["surface_area"], rtol=1e-5)


def test_mesh_properties_and_determinism_256(dev):
    """Size-independent properties: closed 2-manifold (every edge shared by exactly two faces with opposite
    orientation), Euler characteristic 2, sorted unique vertices, no degenerate faces, run-to-run identical."""
    _, _, _, sm, v, f = run_ellipsoid(dev, 256, 256, 256)
    v2, f2 = pipeline.extract_surface(sm, np.full(256, 1.0), 1.0, 1.0)
    assert torch.equal(v, v2) and torch.equal(f, f2)
    vn, fn = v.cpu().numpy(), f.cpu().numpy()
    order = np.lexsort((vn[:, 2], vn[:, 1], vn[:, 0]))
    assert np.array_equal(order, np.arange(len(vn)))
    assert len(np.unique(vn, axis=0)) == len(vn)
    assert np.all(fn[:, 0] != fn[:, 1]) and np.all(fn[:, 1] != fn[:, 2]) and np.all(fn[:, 0] != fn[:, 2])
    e = np.concatenate([fn[:, [0, 1]], fn[:, [1, 2]], fn[:, [2, 0]]])
    directed = e[:, 0] * len(vn) + e[:, 1]
    assert len(np.unique(directed)) == len(directed)                 # each directed edge once
    rev = e[:, 1] * len(vn) + e[:, 0]
    assert np.array_equal(np.sort(directed), np.sort(rev))           # and its reverse exists
    assert len(vn) - len(e) // 2 + len(fn) == 2                      # V - E + F = 2


@pytest.mark.parametrize("world", [2, 3])
def test_slab_ranks_on_one_gpu_match_single_gpu(dev, world, tmp_path):
    """The multi-GPU (Z-slab) path with the HIP engine: `world` rank threads share this GPU and talk through the
    in-process communicator; the concatenated result must be byte-identical to the single-GPU mesh, the OBJ file the
    ranks write together to the single-GPU export, and the whole-stack volume / bounding box every rank reports to
    VolumeCalculator's numbers (variable slice depths, anisotropic pixels)."""
    import contextlib
    import io
    import threading
    from tomography_3d_reconstructor_amd import slab
    from tomography_3d_reconstructor_amd.obj_exporter import OBJExporter
    from tomography_3d_reconstructor_amd.volume_calculator import VolumeCalculator
    nz, ny, nx = 96, 80, 112
    rng = np.random.default_rng(3)
    v = np.stack(O.ellipsoid_masks(nz, ny, nx))
    v ^= rng.random(v.shape) < 0.01
    v[0, 5:60, 7:90] = True
    v[0, 20:30, 30:50] = False
    v[nz // 2 - 1:nz // 2 + 1, ny // 2] = rng.random((2, nx)) < 0.5
    depths = np.concatenate([np.full(16, 0.5), np.full(64, 0.25), np.full(16, 0.5)])
    vol = pipeline.smooth(pipeline.close_ends(to_vol(v, dev)), 3, True)
    rv, rf = pipeline.extract_surface(vol, depths, 0.7, 0.9)
    out, errs = [None] * world, []

    def target(c):
        try:
            job = slab.SlabJob(nz, ny, nx, c)
            with torch.cuda.stream(torch.cuda.Stream()):
                mask = torch.from_numpy(v[job.z0:job.z1].astype(np.uint8)).to(dev)
                verts, faces = job.run(mask, depths, 0.7, 0.9)
                extras = (job.voxel_volume(0.9, 0.7, depths), job.bounding_box(0.9, 0.7, depths),
                          job.voxel_volume(0.9, 0.7, depths, "created"), job.export_obj(str(tmp_path / "slab.obj"), nthreads=2))
                torch.cuda.current_stream().synchronize()
            out[c.rank] = (verts.cpu().numpy(), faces.cpu().numpy(), job.vertex_offset, job.n_vertices_global, extras)
        except BaseException as e:   # noqa: BLE001
            errs.append(e)
            raise

    ts = [threading.Thread(target=target, args=(c,)) for c in slab.ThreadComm.make(world)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not errs, errs
    verts = np.concatenate([o[0] for o in out])
    faces = np.concatenate([o[1] for o in out])
    assert out[0][3] == rv.shape[0]
    assert verts.tobytes() == rv.cpu().numpy().tobytes()
    assert np.array_equal(faces, rf.cpu().numpy())
    # consumers: VolumeCalculator on the whole (host) volumes, OBJExporter on the single-GPU mesh
    vc = VolumeCalculator()
    created_h, smoothed_h = to_np(pipeline.close_ends(to_vol(v, dev))), to_np(vol)
    with contextlib.redirect_stdout(io.StringIO()):
        assert OBJExporter().export_to_obj(rv.cpu().numpy(), rf.cpu().numpy(), str(tmp_path / "single.obj"))
    single = open(tmp_path / "single.obj", "rb").read()
    assert open(tmp_path / "slab.obj", "rb").read() == single
    ref_box = vc.calculate_bounding_box_variable_depth(smoothed_h, 0.9, 0.7, depths)
    for o in out:
        vol_s, box_s, vol_c, obj_bytes = o[4]
        assert vol_s == vc.calculate_voxel_volume_variable_depth(smoothed_h, 0.9, 0.7, depths)
        assert vol_c == vc.calculate_voxel_volume_variable_depth(created_h, 0.9, 0.7, depths)
        assert box_s == ref_box and obj_bytes == len(single)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "SO_PATH", "/nonexistent/libtomo_hip.so")
    with pytest.raises(_lib.TomoError):
        _lib.lib()


def test_unique_one_sort_path_detects_disorder_and_falls_back(dev):
    """tomo_mesh_unique_presorted is exact only for rows in marching-cubes order; it must count every place where its
    result descends, and pipeline.ensure_manifold_mesh must then fall back to the two-sort path: np.unique either way."""
    rng = np.random.default_rng(9)
    L = _lib.lib()
    for ordered in (True, False):
        v = rng.integers(0, 6, (5000, 3)).astype(np.float32) * np.float32(0.37)
        if ordered:     # marching-cubes-like: plane vertices of slice Z = index of the z value, rows sorted, duplicates kept
            v = v[np.lexsort((v[:, 2], v[:, 1], v[:, 0]))]
            zidx = np.searchsorted(np.unique(v[:, 0]), v[:, 0]).astype(np.int64)
            key = (zidx << 22)                                   # Ny = 1, slot 0: bucket 2 Z, sub-key y
        else:
            key = np.zeros(len(v), np.int64)                     # everything in one bucket, sorted by y only
        f = rng.integers(0, len(v), (7000, 3)).astype(np.int32)
        vt, ft, kt = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev), torch.from_numpy(key).to(dev)
        totals = torch.zeros(4, dtype=torch.int64, device=dev)
        uniq = torch.empty_like(vt)
        rank = torch.empty(len(v), dtype=torch.int32, device=dev)
        wsb = L.tomo_mesh_unique_workspace_bytes(len(v))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        _lib.check(L.tomo_mesh_unique_presorted(vt.data_ptr(), kt.data_ptr(), len(v), 1, 6 if ordered else 1, uniq.data_ptr(), rank.data_ptr(),
                                                totals.data_ptr(), ws.data_ptr(), wsb, torch.cuda.current_stream().cuda_stream),
                   "presorted")
        nviol = int(totals[2].item())
        assert (nviol == 0) == ordered
        eu, einv = np.unique(v, axis=0, return_inverse=True)
        if ordered:
            assert np.array_equal(uniq[: int(totals[0].item())].cpu().numpy(), eu)
            assert np.array_equal(rank.cpu().numpy(), einv.reshape(-1))
        mesh = pipeline.RawMesh(kt, vt, ft)
        mesh._ny, mesh._nz = 1, (6 if ordered else 1)
        gv, gf = pipeline.ensure_manifold_mesh(mesh)
        ev, ef = O.ensure_manifold_mesh(v, f)
        assert np.array_equal(gv.cpu().numpy(), ev) and np.array_equal(gf.cpu().numpy(), ef)


@pytest.mark.parametrize("shape,pad", [((5, 7, 9), 1), ((5, 7, 9), 0), ((9, 33, 64), 1), ((4, 20, 1), 0), ((3, 3, 27), 1),
                                       ((6, 40, 59), 0), ((7, 18, 130), 1), ((3, 5, 2050), 0), ((2, 2, 2), 1), ((1, 1, 1), 0)])
def test_field_from_bits_equals_field_from_extended_volume(dev, shape, pad, monkeypatch):
    """tomo_field_fill_bits (border rules applied while staging) == tomo_extend_bits + tomo_field_fill, bit for bit,
    including the sign records and group classes -- and both equal the oracle."""
    rng = np.random.default_rng(shape[2] * 7 + pad)
    v = rng.random(shape) < 0.55
    vol = to_vol(v, dev)
    res = []
    for flag in (True, False):
        monkeypatch.setattr(pipeline, "FIELD_FROM_BITS", flag)
        f = pipeline.make_field(vol, True, bool(pad))
        res.append((f.dense().cpu().numpy().copy(), f.gcls.cpu().numpy().copy(), f))
    assert res[0][0].tobytes() == res[1][0].tobytes() == O.field(v, True, bool(pad)).tobytes()
    assert np.array_equal(res[0][1], res[1][1])
    stored = torch.from_numpy(res[0][1] == 2).to(dev)             # groups whose records exist: (Nz, G, S)
    a, b = res[0][2].signs, res[1][2].signs                         # (Nz, S, NyP, 4)
    m = stored.permute(0, 2, 1).repeat_interleave(16, dim=2)[:, :, :, None]
    Ny = res[0][2].Ny
    assert torch.equal((a * m)[:, :, :Ny], (b * m)[:, :, :Ny])


def test_whole_path_irregular_blobs_vs_oracle(dev):
    """A mid-sized irregular volume (overlapping balls, specks, a slab against the border, odd dimensions) through the
    whole path: exercises mixed tiles everywhere, duplicate vertices and degenerate faces (the fallbacks of the
    speculative unique / face kernels), the wave smoothing kernel across tile edges -- bit-exact against the oracle."""
    rng = np.random.default_rng(21)
    nz, ny, nx = 90, 150, 203
    zz, yy, xx = np.mgrid[0:nz, 0:ny, 0:nx]
    v = np.zeros((nz, ny, nx), bool)
    for _ in range(40):
        c = rng.random(3) * [nz, ny, nx]
        r = 4 + rng.random() * 18
        v |= ((zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2) <= r * r
    v ^= rng.random(v.shape) < 0.003
    v[:, :40, -3:] = True
    v[0] |= rng.random((ny, nx)) < 0.3
    depths = np.concatenate([np.full(20, 0.1), np.full(50, 0.3), np.full(20, 0.0)])     # zero depths: z values collapse
    vol = pipeline.smooth(pipeline.close_ends(to_vol(v, dev)), 3, True)
    osm = O.smooth(O.close_ends(v), 3, True)
    assert np.array_equal(to_np(vol), osm)
    before = dict(pipeline.COUNTERS)
    gv, gf = pipeline.extract_surface(vol, depths, 0.31, 0.47)
    ev, ef = O.SurfaceExtractor().extract_manifold_surface(osm, depths, 0.31, 0.47)
    assert gv.cpu().numpy().tobytes() == ev.tobytes() and np.array_equal(gf.cpu().numpy(), ef)
    assert len(ev) > 50000 and any(pipeline.COUNTERS[k] != before[k] for k in pipeline.COUNTERS)


# ------------------------------------------------------------------ sparse field
@pytest.mark.parametrize("case", ["ellipsoid", "blobs", "noise", "touching", "wide", "thin", "nopad", "nopad_solid"])
def test_sparse_field_same_mesh(dev, case):
    """tomo_field_fill_bits_sparse leaves constant tiles unwritten that no marching-cubes cell can touch.  The buffer is
    pre-filled with NaN here, so any read of an unwritten float would poison the mesh: mesh == the dense field's mesh,
    bit for bit, and what is written equals the dense field."""
    rng = np.random.default_rng(11)
    if case == "ellipsoid":
        v = np.asarray(O.ellipsoid_masks(96, 80, 40))
    elif case == "blobs":
        v = rng.random((30, 70, 150)) < 0.5
        for _ in range(3):
            v = O.smooth(v, 1, True)
    elif case == "noise":
        v = rng.random((12, 40, 90)) < 0.5
    elif case == "touching":          # solid up to every border: the pad ring is the only zero in reach
        v = np.ones((9, 37, 131), bool)
        v[4, 20, 70] = False
    elif case == "wide":              # several x chunks of the field kernel, tiles beyond the last column
        v = np.zeros((6, 20, 2100), bool)
        v[2:4, 5:15, 3:2098] = True
    elif case == "thin":
        v = np.zeros((5, 3, 33), bool)
        v[2, 1, 5:30] = True
    elif case == "nopad_solid":       # no pad ring and solid up to the array's edge: reflect makes the border tiles constant
        v = np.ones((37, 19, 95), bool)
        v[5, 1, 7] = False
    else:
        v = np.asarray(O.ellipsoid_masks(48, 40, 24))
    pad = case not in ("nopad", "nopad_solid")
    vol = to_vol(v, dev)
    L = pipeline._lib.lib()
    nz, ny, nx = v.shape
    p = 1 if pad else 0
    fd = pipeline.make_field(vol, True, pad)
    fs = pipeline.make_field(vol, True, pad, sparse=True)
    assert fs.sparse and not fd.sparse
    # the same call on an explicitly poisoned buffer
    poisoned = torch.full_like(fs.data, float("nan"))
    span = torch.empty(L.tomo_field_span_bytes(nz, ny, nx, p), dtype=torch.uint8, device=dev)
    sb = torch.empty_like(fs.signs)
    gc = torch.empty_like(fs.gcls)
    st = torch.cuda.current_stream().cuda_stream
    assert L.tomo_field_fill_bits_sparse(vol.bits.data_ptr(), poisoned.data_ptr(), nz, ny, nx, p, sb.data_ptr(), gc.data_ptr(),
                                         span.data_ptr(), st) == 0
    # group classes: identical, or constant (0 / 1, no records) where the dense kernel stored records -- that one classifies a
    # tile by a window that is not clipped to the array, so a tile at the array's edge can be "mixed" there and constant here
    assert bool(((gc == fd.gcls) | ((gc < 2) & (fd.gcls == 2))).all())
    written = ~torch.isnan(poisoned)
    assert torch.equal(poisoned[written], fd.data[written])          # what is written is the dense field
    fs.data = poisoned
    fs.signs, fs.gcls = sb, gc
    md, ms = pipeline.marching_cubes(fd, 0.5), pipeline.marching_cubes(fs, 0.5)
    if md is None:
        assert ms is None
        return
    assert torch.equal(md.vkey, ms.vkey) and torch.equal(md.faces32, ms.faces32)
    assert torch.equal(md.vpos.view(torch.int32), ms.vpos.view(torch.int32))      # bitwise: no NaN got in
    if case == "ellipsoid":
        assert float(written.float().mean()) < 0.9                                 # and something was actually left out
    with pytest.raises(Exception):
        pipeline.marching_cubes(fs, 0.3)                                           # other levels need the dense field


# ------------------------------------------------------------------ size hints
def test_marching_cubes_size_hints(dev):
    """A second field of the same geometry launches list / eval / scan ahead of the first count download, into buffers
    sized from the previous list: same mesh when the hint holds, when it is far too large, and when it is too small
    (the guarded kernels must not touch anything, the plain path redoes the stage)."""
    rng = np.random.default_rng(5)
    shape = (24, 60, 140)
    small = np.zeros(shape, bool); small[10:14, 20:30, 30:60] = True
    big = rng.random(shape) < 0.5                                   # noise: a far longer active-voxel list
    empty = np.zeros(shape, bool)
    assert pipeline.NA_HINTS

    def mesh_of(v, hints):
        pipeline.NA_HINTS = hints
        try:
            f = pipeline.make_field(to_vol(v, dev))
            m = pipeline.marching_cubes(f, 0.5)
        finally:
            pipeline.NA_HINTS = True
        return None if m is None else (m.vkey.clone(), m.vpos.clone(), m.faces32.clone())

    ref = {k: mesh_of(v, False) for k, v in (("small", small), ("big", big), ("empty", empty))}
    pipeline._NA_HINT.clear()
    c0 = dict(pipeline.COUNTERS)
    order = ["small", "small", "big", "big", "small", "empty", "small"]     # no hint, hit, miss (too short), hit, hit (too long), ...
    vols = {"small": small, "big": big, "empty": empty}
    for k in order:
        got = mesh_of(vols[k], True)
        if ref[k] is None:
            assert got is None
            continue
        for a, b in zip(got, ref[k]):
            assert torch.equal(a, b), k
    assert pipeline.COUNTERS.get("na_hint_miss", 0) - c0.get("na_hint_miss", 0) >= 1
    assert pipeline.COUNTERS.get("na_hint_hit", 0) - c0.get("na_hint_hit", 0) >= 3


def test_mc3_chain_size_hints(dev):
    """The mc3 chain is enqueued into buffers sized from the counts of the last surface of the same geometry (+ 25 %) before
    anything comes back from the device: same mesh as the oracle when the hint holds, when it is far too large, when it is
    too small in the list / the vertices / the triangles (flagged on the device, nothing written past a buffer, the pass is
    redone with exact sizes), without hints, and for an empty volume after a full one."""
    rng = np.random.default_rng(6)
    shape = (24, 60, 144)
    small = np.zeros(shape, bool); small[10:14, 20:30, 30:60] = True
    big = rng.random(shape) < 0.5                                   # noise: a far longer list, centre vertices, duplicate rows
    mid = np.stack(O.ellipsoid_masks(*shape))
    empty = np.zeros(shape, bool)
    depths = np.linspace(0.2, 0.6, shape[0])
    ref = {k: O.SurfaceExtractor().extract_manifold_surface(v, depths, 0.8, 1.1) for k, v in
           (("small", small), ("big", big), ("mid", mid), ("empty", empty))}
    vols = {"small": small, "big": big, "mid": mid, "empty": empty}
    if not pipeline.MC3:
        pytest.skip("TOMO_MC_PATH=old: the round-1 kernels are selected")
    assert pipeline.NA_HINTS
    pipeline._MC3_HINT.clear()
    c0 = dict(pipeline.COUNTERS)
    for k in ["small", "small", "big", "big", "mid", "small", "empty", "big", "empty", "mid"]:
        got = pipeline.extract_surface(to_vol(vols[k], dev), depths, 0.8, 1.1)
        if ref[k] is None:
            assert got is None, k
            continue
        assert got[0].cpu().numpy().tobytes() == ref[k][0].tobytes() and np.array_equal(got[1].cpu().numpy(), ref[k][1]), k
    assert pipeline.COUNTERS.get("mc3_hint_miss", 0) - c0.get("mc3_hint_miss", 0) >= 2
    assert pipeline.COUNTERS.get("mc3_hint_hit", 0) - c0.get("mc3_hint_hit", 0) >= 3
    # a hint that fits the list but not the vertices / not the triangles
    for which in (1, 2):
        pipeline.extract_surface(to_vol(mid, dev), depths, 0.8, 1.1)
        key = next(iter(k for k in pipeline._MC3_HINT if k[0] == shape[0] + 2))
        h = list(pipeline._MC3_HINT[key])
        h[which] = 8
        pipeline._MC3_HINT[key] = tuple(h)
        miss = pipeline.COUNTERS.get("mc3_hint_miss", 0)
        got = pipeline.extract_surface(to_vol(big, dev), depths, 0.8, 1.1)
        assert pipeline.COUNTERS.get("mc3_hint_miss", 0) == miss + 1
        assert got[0].cpu().numpy().tobytes() == ref["big"][0].tobytes() and np.array_equal(got[1].cpu().numpy(), ref["big"][1])
    with_hints = pipeline.extract_surface(to_vol(big, dev), depths, 0.8, 1.1)
    pipeline.NA_HINTS = False
    try:
        without = pipeline.extract_surface(to_vol(big, dev), depths, 0.8, 1.1)
    finally:
        pipeline.NA_HINTS = True
    assert torch.equal(with_hints[0], without[0]) and torch.equal(with_hints[1], without[1])


# ------------------------------------------------------------------ masks that touch the first / last slice
@pytest.mark.parametrize("kind", ["first", "both", "noise_first"])
def test_one_sort_unique_with_clamped_first_slice(dev, kind):
    """A mask in the first slice puts vertices on the z edges below it; the slice-depth map clamps their z to 0 -- the z of
    the in-plane vertices of that slice -- so np.unique interleaves two runs of the one-sort path.  The merge of those runs
    keeps the fast path exact: same mesh as the oracle WITHOUT falling back to the general sort."""
    rng = np.random.default_rng(3)
    shape = (12, 40, 90)
    if kind == "first":
        v = np.zeros(shape, bool); v[0:5, 8:30, 10:70] = True
    elif kind == "both":
        v = np.zeros(shape, bool); v[:, 5:35, 20:60] = True; v[4:8, 15:20, 30:40] = False
    else:
        v = np.zeros(shape, bool); v[0:3] = rng.random((3,) + shape[1:]) < 0.5
        v = O.smooth(v, 1, True); v[0] |= rng.random(shape[1:]) < 0.2
    depths = np.linspace(0.3, 0.9, shape[0])
    ref = O.SurfaceExtractor().extract_manifold_surface(v, depths, 0.7, 1.3)
    c0 = dict(pipeline.COUNTERS)
    got = pipeline.extract_surface(to_vol(v, dev), depths, 0.7, 1.3, True, True)
    assert ref is not None and got is not None
    gv, gf = got[0].cpu().numpy(), got[1].cpu().numpy()
    assert gv.shape == ref[0].shape and np.array_equal(gv.view(np.int32), np.ascontiguousarray(ref[0]).view(np.int32))
    assert np.array_equal(gf, ref[1])
    if kind != "noise_first":                     # (noise may break the one-sort order elsewhere: float32 ties between buckets)
        if pipeline.MC3:
            assert pipeline.COUNTERS.get("mc3_exact", 0) - c0.get("mc3_exact", 0) == 1
            assert pipeline.COUNTERS.get("mc3_general_unique", 0) - c0.get("mc3_general_unique", 0) == 0
        else:
            assert pipeline.COUNTERS["unique_one_sort"] - c0["unique_one_sort"] == 1
            assert pipeline.COUNTERS["unique_fallback"] - c0["unique_fallback"] == 0


@pytest.mark.parametrize("ny,kind", [(638, "body"), (639, "body"), (1400, "body"), (1400, "wide"), (1400, "first"), (660, "first_thin"), (1560, "noise")])
def test_planes_sorted_in_bands_of_owner_rows_vs_oracle(dev, ny, kind):
    """Planes of more than 640 field rows (ny + 2 with padding) are sorted in bands of 512 owner rows (tomo_common.h:
    tomo_sort_band; mc3_bands_kernel) -- the order inside a plane is local to a row.  Tall, thin stacks either side of the switch:
    a body whose surface crosses every band boundary (`body`: a narrow one, its flat faces fit the hand-written sort's LDS;
    `wide`: ~40 000 vertices between two planes, the library path), a mask in the first slice (the clamped run is merged with
    the first plane, now several segments: `first` takes the library path, `first_thin` -- a short line of pixels -- the kernel's
    two-pass (x', then y') order), noise (rows with hundreds of vertices, ties).  Same mesh as the oracle, and -- except for
    noise, which may tie across buckets -- without the general sort."""
    nz, nx = 5, 48
    rng = np.random.default_rng(ny)
    yy, xx = np.mgrid[0:ny, 0:nx]
    if kind == "noise":
        v = O.smooth(rng.random((nz, ny, nx)) < 0.5, 1, True)
    else:
        half = {"body": 0.03, "first_thin": 0.01}.get(kind, 0.4) * nx           # half width of the body along x
        disc = ((yy - ny / 2) / (0.47 * ny)) ** 2 + ((xx - nx / 2) / half) ** 2 <= 1.0
        if kind == "first_thin":                                                  # a short line of pixels across the band boundary at row 512:
            disc = ((yy - 0.75 * ny) / (0.15 * ny)) ** 2 + ((xx - nx / 2) / half) ** 2 <= 1.0    # its clamped run fits the kernel's 1 024
        v = np.zeros((nz, ny, nx), bool)
        v[(0 if kind.startswith("first") else 1):4] = disc
        v[2, ::37, 5:nx - 5:3] ^= True                                   # specks: vertices between the planes, in every band
    depths = np.linspace(0.4, 0.8, nz)
    ref = O.SurfaceExtractor().extract_manifold_surface(v, depths, 0.7, 1.1)
    c0 = dict(pipeline.COUNTERS)
    pipeline._MC3_LARGE.pop((nz + 2, ny + 2, nx + 2, 0), None)
    got = pipeline.extract_surface(to_vol(v, dev), depths, 0.7, 1.1, True, True)
    assert ref is not None and got is not None
    gv, gf = got[0].cpu().numpy(), got[1].cpu().numpy()
    assert gv.shape == ref[0].shape and np.array_equal(gv.view(np.int32), np.ascontiguousarray(ref[0]).view(np.int32))
    assert np.array_equal(gf, ref[1])
    L = _lib.lib()
    assert L.tomo_mc3_sort_segments(nz + 2, ny + 2) == (nz + 2) * ((1 if ny + 2 <= 640 else -(-(ny + 2) // 512)) + 1)
    if kind != "noise" and pipeline.MC3:
        assert pipeline.COUNTERS.get("mc3_general_unique", 0) - c0.get("mc3_general_unique", 0) == 0
    if pipeline.MC3 and pipeline.FUSED_SORT and kind != "noise":
        # which sort ran: the hand-written kernel, unless a segment (the flat faces of `wide` / `first`) is too long for its LDS
        # -- then the stage is repeated with the library's segmented sort, once and for good
        lib = pipeline.COUNTERS.get("mc3_sort_library", 0) - c0.get("mc3_sort_library", 0)
        assert (lib >= 1) == (kind in ("wide", "first")), (kind, pipeline.COUNTERS)
        assert bool(pipeline._MC3_LARGE.get((nz + 2, ny + 2, nx + 2, 0))) == (kind in ("wide", "first"))
    # once more from the size hints (everything enqueued before any count is known): identical
    again = pipeline.extract_surface(to_vol(v, dev), depths, 0.7, 1.1, True, True)
    assert torch.equal(again[0], got[0]) and torch.equal(again[1], got[1])
