"""Put this directory first on sys.path and the reference's `from image_loader import ImageLoader`
(tomography_3d_reconstruction.py:13) binds the MI355X implementation."""
from tomography_3d_reconstructor_amd.image_loader import ImageLoader  # noqa: F401
