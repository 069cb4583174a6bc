"""Put this directory first on sys.path and the reference's `from voxel_processor import VoxelProcessor`
(tomography_3d_reconstruction.py:14) binds the MI355X implementation."""
from tomography_3d_reconstructor_amd.voxel_processor import VoxelProcessor  # noqa: F401
