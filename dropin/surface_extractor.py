"""Put this directory first on sys.path and the reference's `from surface_extractor import SurfaceExtractor`
(tomography_3d_reconstruction.py:15) binds the MI355X implementation."""
from tomography_3d_reconstructor_amd.surface_extractor import SurfaceExtractor  # noqa: F401
