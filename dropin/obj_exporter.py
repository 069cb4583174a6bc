"""Put this directory first on sys.path and `from obj_exporter import OBJExporter` (the reference's stand-alone OBJ
writer, obj_exporter.py:11) binds the native writer of the MI355X build."""
from tomography_3d_reconstructor_amd.obj_exporter import OBJExporter  # noqa: F401
