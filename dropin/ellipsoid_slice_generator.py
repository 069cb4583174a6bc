"""Put this directory first on sys.path and `from ellipsoid_slice_generator import EllipsoidSliceGenerator`
(simple_generator.py:4 of the reference) binds the cv2-free generator of this build."""
from tomography_3d_reconstructor_amd.slice_generator import EllipsoidSliceGenerator  # noqa: F401
