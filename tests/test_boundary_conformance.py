"""CPU tier, build container only: the drop-in boundary against the reference's own caller and class definitions
(the reference has no FFI; its boundary for this path is a set of importable classes,
/root/reference/tomography_3d_reconstruction.py:13-17, 32-36).

  * the reference's orchestrator, imported UNCHANGED with dropin/ in front of the reference on sys.path, binds
    ImageLoader / VoxelProcessor / SurfaceExtractor / VolumeCalculator to this package and instantiates;
  * every public method of the drop-in classes has the signature of its namesake in the reference: names, order, kinds,
    defaults and annotations (/root/reference/voxel_processor.py:30-164, surface_extractor.py:31-149,
    volume_calculator.py:13-132, obj_exporter.py:14-38, image_loader.py:17-137, ellipsoid_slice_generator.py:9-143).

The reference's signatures are taken from its source TEXT (ast): a stub class with the same `def` lines and `pass` bodies is
compiled, so nothing of the reference runs and modules that need cv2 (absent here) are covered too.  /root/reference never
travels: on the GPU box these tests skip.
"""
import ast
import inspect
import os
import subprocess
import sys
import typing

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is only present in the build container")

PAIRS = [  # (reference file, class, drop-in module)
    ("voxel_processor.py", "VoxelProcessor", "voxel_processor"),
    ("surface_extractor.py", "SurfaceExtractor", "surface_extractor"),
    ("volume_calculator.py", "VolumeCalculator", "volume_calculator"),
    ("obj_exporter.py", "OBJExporter", "obj_exporter"),
    ("image_loader.py", "ImageLoader", "image_loader"),
]


def reference_signatures(path, cls):
    """{method name: inspect.Signature} of class `cls` in the reference file, from its source text."""
    tree = ast.parse(open(path).read())
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    stubs = [ast.FunctionDef(name=f.name, args=f.args, body=[ast.Pass()], decorator_list=[], returns=f.returns, type_comment=None)
             for f in node.body if isinstance(f, ast.FunctionDef)]
    mod = ast.fix_missing_locations(ast.Module(body=[ast.ClassDef(name=cls, bases=[], keywords=[], body=stubs, decorator_list=[])],
                                               type_ignores=[]))
    ns = {k: getattr(typing, k) for k in ("Optional", "Tuple", "List", "Dict", "Union", "Any")}
    ns["np"] = np
    exec(compile(mod, path, "exec"), ns)
    return {name: inspect.signature(fn) for name, fn in vars(ns[cls]).items() if inspect.isfunction(fn)}


def public(name):
    return name == "__init__" or not name.startswith("_")


@pytest.mark.parametrize("ref_file,cls,module", PAIRS)
def test_public_method_signatures_equal_the_reference(ref_file, cls, module):
    import importlib
    ours = getattr(importlib.import_module("tomography_3d_reconstructor_amd." + module), cls)
    ref = reference_signatures(os.path.join(REF, ref_file), cls)
    names = [n for n in ref if public(n)]
    assert names, "no public methods found in the reference class"
    for name in names:
        assert hasattr(ours, name), "%s.%s is missing" % (cls, name)
        mine = inspect.signature(getattr(ours, name))
        assert mine == ref[name], "%s.%s: %s here, %s in the reference" % (cls, name, mine, ref[name])
        assert str(mine) == str(ref[name])
    # nothing public here that the reference's callers could mistake for the reference's API under another meaning
    extra = [n for n, f in vars(ours).items() if inspect.isfunction(f) and public(n) and n not in ref]
    assert extra == [], "public methods the reference does not have: %r" % extra


def test_slice_generator_signatures_equal_the_reference():
    from tomography_3d_reconstructor_amd import slice_generator
    ref = reference_signatures(os.path.join(REF, "ellipsoid_slice_generator.py"), "EllipsoidSliceGenerator")
    ours = slice_generator.EllipsoidSliceGenerator
    for name in ("__init__", "generate_slices", "generate_slices_half_ellipsoid"):
        assert str(inspect.signature(getattr(ours, name))) == str(ref[name]), name


def test_reference_orchestrator_binds_to_the_drop_in_unchanged():
    """tomography_3d_reconstruction.py:13-17 -- `from voxel_processor import VoxelProcessor` etc. -- with dropin/ first."""
    code = r"""
import sys
sys.path[:0] = [%r, %r, %r]
import tomography_3d_reconstruction as T
pkg = "tomography_3d_reconstructor_amd."
for cls, mod in (("ImageLoader", "image_loader"), ("VoxelProcessor", "voxel_processor"), ("SurfaceExtractor", "surface_extractor"),
                 ("VolumeCalculator", "volume_calculator")):
    assert getattr(T, cls).__module__ == pkg + mod, (cls, getattr(T, cls).__module__)
assert T.__file__.startswith(%r), T.__file__
job = T.Tomography3DReconstruction(10.0, 12.0, 6.0)          # :24-36: instantiates every component
assert type(job.voxel_processor).__module__ == pkg + "voxel_processor"
assert type(job.surface_extractor).__module__ == pkg + "surface_extractor"
assert type(job.image_loader).__module__ == pkg + "image_loader"
assert type(job.volume_calculator).__module__ == pkg + "volume_calculator"
assert type(job.visualizer).__module__ == "visualizer"         # the reference's own: out of scope, untouched
print("BOUND")
""" % (os.path.join(ROOT, "dropin"), ROOT, REF, REF)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")          # /root/reference is read-only
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp", env=env, timeout=600)
    assert out.returncode == 0 and "BOUND" in out.stdout, out.stderr[-2000:]
