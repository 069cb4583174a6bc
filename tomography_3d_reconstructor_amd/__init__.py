"""MI355X-native mask stack -> scalar field ("SDF") -> marching-cubes mesh path.

Drop-in for the reference's `voxel_processor.VoxelProcessor` and `surface_extractor.SurfaceExtractor`
(see INTEGRATION.md); `pipeline` is the device-resident API underneath.
"""
from . import _lib  # noqa: F401
from .voxel_processor import VoxelProcessor  # noqa: F401
from .surface_extractor import SurfaceExtractor  # noqa: F401

__all__ = ["VoxelProcessor", "SurfaceExtractor"]
