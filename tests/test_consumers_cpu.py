"""CPU tier for the rows of SURVEY.md 8(f) either side of the hot path: the oracle's restatements of
volume_calculator.py / obj_exporter.py / image_loader.py against the fixtures generated from the reference
(tests/golden/consumers.npz), and the product's NATIVE OBJ writer (host code of libtomo_hip.so: runs without a GPU)
byte for byte against the reference's own output file."""
import contextlib
import io
import os

import numpy as np
import pytest

from oracle import oracle as O
from tomography_3d_reconstructor_amd.obj_exporter import OBJExporter

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "consumers.npz"))


def case(ci):
    k = "c%d_" % ci
    shape = tuple(int(s) for s in G[k + "shape"])
    vol = np.unpackbits(G[k + "vol"])[: int(np.prod(shape))].reshape(shape).astype(bool)
    return k, vol, G[k + "depths"], [float(x) for x in G[k + "mm"]]


def bbox_vec(b):
    return np.asarray([*b["x"], *b["y"], *b["z"], *b["dimensions"]], np.float64)


def check_volume_calculator(vc, ci):
    """Shared with the GPU tier: every number bit-identical to what the reference returned."""
    k, vol, depths, (mmx, mmy, mms) = case(ci)
    assert np.float64(vc.calculate_voxel_volume(vol, mmx, mmy, mms)).tobytes() == G[k + "voxel_volume"].tobytes()
    assert np.float64(vc.calculate_voxel_volume_variable_depth(vol, mmx, mmy, depths)).tobytes() == G[k + "voxel_volume_var"].tobytes()
    assert np.float64(vc.calculate_voxel_volume_variable_depth(vol, mmx, mmy, depths[:3])).tobytes() == G[k + "voxel_volume_var_short"].tobytes()
    assert vc.calculate_voxel_volume_variable_depth(vol, mmx, mmy, np.array([])) == 0.0
    if vol.any():
        assert bbox_vec(vc.calculate_bounding_box(vol, mmx, mmy, mms)).tobytes() == G[k + "bbox"].tobytes()
    else:
        with pytest.raises(ValueError):
            vc.calculate_bounding_box(vol, mmx, mmy, mms)
    assert bbox_vec(vc.calculate_bounding_box_variable_depth(vol, mmx, mmy, depths)).tobytes() == G[k + "bbox_var"].tobytes()
    assert np.float64(vc.calculate_density(12.5, 95.03, 143.1, _total_depth(ci))).tobytes() == G[k + "density"].tobytes()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        props = vc.analyze_object_properties(vol, 3.25, None if ci % 2 else 2.75, 0.0 if ci == 2 else 7.5, mmx, mmy, depths,
                                             95.03, 143.1, _total_depth(ci))
    assert buf.getvalue().encode("utf-8") == G[k + "analyze_stdout"].tobytes()
    nums = np.asarray([props["volume_mm3"], props["voxel_volume_mm3"], props["density"], *props["dimensions"]], np.float64)
    assert nums.tobytes() == G[k + "analyze_nums"].tobytes()


def _total_depth(ci):
    return [6.0, 3.3, 10.0, 1.0][ci]


@pytest.mark.parametrize("ci", range(int(G["n_cases"])))
def test_oracle_volume_calculator_matches_reference(ci):
    check_volume_calculator(O.VolumeCalculator(), ci)


@pytest.mark.parametrize("name", ["mesh", "nofaces"])
def test_oracle_obj_text_matches_reference(name):
    assert O.obj_text(G["obj_%s_verts" % name], G["obj_%s_faces" % name]).encode() == G["obj_%s_bytes" % name].tobytes()


@pytest.mark.parametrize("name", ["mesh", "nofaces"])
def test_native_obj_writer_matches_reference_bytes(name, tmp_path):
    """tomo_obj_write through the drop-in OBJExporter: same bytes, same return value, same print."""
    pth = str(tmp_path / (name + ".obj"))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ok = OBJExporter().export_to_obj(G["obj_%s_verts" % name], G["obj_%s_faces" % name], pth)
    assert ok == bool(G["obj_%s_ok" % name])
    assert open(pth, "rb").read() == G["obj_%s_bytes" % name].tobytes()
    assert buf.getvalue().replace(str(tmp_path), "<DIR>").encode() == G["obj_%s_stdout" % name].tobytes()


def test_native_obj_writer_formats_like_python(tmp_path):
    """%.6f edge cases: exact ties (2^-7), negative zero, tiny, large, float64 input, int32 faces, many threads."""
    rng = np.random.default_rng(5)
    v = np.concatenate([rng.standard_normal((70000, 3)).astype(np.float32) * 300,
                        np.array([[0.0078125, -0.0078125, 0.0234375], [1e-9, -1e-9, -0.0], [123456.789, 0.5e-6, 1.5e-6],
                                  [2.5e-6, 3.5e-6, 16777216.0], [3e38, -3e38, 1e12]], dtype=np.float32)])
    f = rng.integers(0, len(v), (50000, 3)).astype(np.int32)
    for arr in (v, v.astype(np.float64) * np.pi):
        pth = str(tmp_path / "t.obj")
        with contextlib.redirect_stdout(io.StringIO()):
            assert OBJExporter().export_to_obj(arr, f, pth)
        assert open(pth).read() == O.obj_text(arr, f)


def test_native_obj_writer_failure_is_reported_like_the_reference(tmp_path):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ok = OBJExporter().export_to_obj(np.zeros((1, 3), np.float32), np.zeros((0, 3), np.int64),
                                         str(tmp_path / "no_such_dir" / "x.obj"))
    assert ok is False and buf.getvalue().startswith("Export failed: ")


def test_oracle_loader_orders_and_thresholds(tmp_path):
    """image_loader.py:27-35,70,108 restated: numeric-suffix order (negative and dotted numbers), >= threshold,
    skipped folders / unreadable / wrong-shape files."""
    from PIL import Image
    rng = np.random.default_rng(2)
    imgs = {}
    for side, names in {"Section_0": ["Mask_P_-2.png", "Mask_P_-10.png", "Mask_P_3.png"],
                        "Section_1": ["Mask_P_10.png", "Mask_P_9.png", "Mask_P_9.5.png", "Mask_P_9.12.png"],
                        "Section_2": ["Mask_P_1.png"]}.items():
        os.makedirs(tmp_path / side)
        for n in names:
            a = rng.integers(0, 256, (12, 20), dtype=np.uint8)
            Image.fromarray(a, mode="L").save(tmp_path / side / n)
            imgs[n] = a
    Image.fromarray(np.zeros((5, 5), np.uint8), mode="L").save(tmp_path / "Section_1" / "Mask_P_11.png")   # wrong shape
    open(tmp_path / "Section_1" / "Mask_P_12.png", "wb").write(b"not a png")                                # unreadable
    masks, counts, files = O.load_masks(str(tmp_path), 200)
    order = ["Mask_P_-10.png", "Mask_P_-2.png", "Mask_P_3.png", "Mask_P_9.png", "Mask_P_9.5.png", "Mask_P_9.12.png",
             "Mask_P_10.png", "Mask_P_1.png"]
    assert [os.path.basename(f) for f in files][:7] == order[:7] and counts == (3, 6, 1)
    assert len(masks) == 8 and all(np.array_equal(m, imgs[n] >= 200) for m, n in zip(masks, order))
    masks2, counts2, _ = O.load_masks(str(tmp_path), 200, (True, False, True))
    assert counts2 == (3, 0, 1) and len(masks2) == 4


def test_slice_generator_half_ellipsoid_stack(tmp_path, capsys):
    """Row N3 (parity with OpenCV unpinned): naming, count, deletion of the extremes, shrinking areas, and the stack is
    what the loader's numeric ordering expects (ellipsoid_slice_generator.py:107-143, simple_generator.py:6-20)."""
    from PIL import Image
    from tomography_3d_reconstructor_amd.slice_generator import EllipsoidSliceGenerator, generate_slices_from_mask
    h, w = 96, 128
    yy, xx = np.mgrid[0:h, 0:w]
    base = ((((xx - 60.0) / 40.0) ** 2 + ((yy - 50.0) / 25.0) ** 2) <= 1.0)
    src = tmp_path / "Section_1"
    src.mkdir()
    Image.fromarray(np.where(base, 255, 0).astype(np.uint8), mode="L").save(src / "Mask_Patient_1.png")
    g = EllipsoidSliceGenerator(str(src / "Mask_Patient_1.png"))
    p = g.ellipse_params
    assert abs(p["center"][0] - 60) < 0.5 and abs(p["center"][1] - 50) < 0.5
    assert abs(p["semi_major_axis"] - 40) < 1.0 and abs(p["semi_minor_axis"] - 25) < 1.0
    assert np.array_equal(g._generate_slice_at_height(0.0, 25.0), g.middle_slice)
    assert not g._generate_slice_at_height(30.0, 25.0).any()
    out0 = tmp_path / "Section_0"
    generate_slices_from_mask(str(src / "Mask_Patient_1.png"), 6, str(out0), 1, False)       # numbers -6 .. 1, extremes removed
    assert "Generated 8 slices" in capsys.readouterr().out
    names = sorted(os.listdir(out0))
    assert names == sorted("Mask_Patient_%d.png" % n for n in range(-5, 1))
    areas = {n: int((np.asarray(Image.open(out0 / ("Mask_Patient_%d.png" % n))) > 127).sum()) for n in range(-5, 1)}
    assert all(areas[n] < areas[n + 1] for n in range(-5, -1)) and areas[-1] <= areas[0] <= int(base.sum())   # grows towards the base mask
    (tmp_path / "Section_2").mkdir()
    masks, counts, files = O.load_masks(str(tmp_path), 200)
    assert counts == (6, 1, 0) and [os.path.basename(f) for f in files][0] == "Mask_Patient_-5.png" and len(masks) == 7
    full = g.generate_slices(5, str(tmp_path / "full"))
    assert [os.path.basename(f) for f in full] == ["Mask_%03d.png" % i for i in range(1, 6)]
