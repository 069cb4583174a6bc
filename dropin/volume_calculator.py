"""Put this directory first on sys.path and the reference's `from volume_calculator import VolumeCalculator`
(tomography_3d_reconstruction.py:17) binds the MI355X implementation."""
from tomography_3d_reconstructor_amd.volume_calculator import VolumeCalculator  # noqa: F401
