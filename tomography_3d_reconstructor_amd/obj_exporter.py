"""Drop-in for the reference's obj_exporter.py (/root/reference/obj_exporter.py:11-37): same
class, same file bytes ("v %.6f %.6f %.6f" per vertex in the (z, y, x) column order the mesh has, "f a+1 b+1 c+1"
per face), written by the native formatter of libtomo_hip.so (tomo_obj_write, parallel on the host cores) instead
of a Python loop per vertex and face.  Host code: the mesh is already in host arrays at this point of the product.
"""
import os

import numpy as np

from . import _lib


class OBJExporter:
    """Handles exporting 3D models to OBJ file format (reference: obj_exporter.py:11)."""

    def __init__(self):
        pass

    def export_to_obj(self, vertices: np.ndarray, faces: np.ndarray,
                      filename: str = "tomography_model.obj") -> bool:
        """obj_exporter.py:17-37: True and "Model exported: ..." on success, False and "Export failed: ..." otherwise."""
        try:
            v = np.asarray(vertices)
            f = np.asarray(faces)
            if v.dtype not in (np.float32, np.float64):
                v = v.astype(np.float64)
            v = np.ascontiguousarray(v)
            nv = len(v)
            if nv and (v.ndim != 2 or v.shape[1] < 3):
                raise IndexError("vertices must have at least three columns")
            if nv and v.shape[1] > 3:
                v = np.ascontiguousarray(v[:, :3])
            nf = len(f)
            if nf and (f.ndim != 2 or f.shape[1] < 3):
                raise IndexError("faces must have at least three columns")
            f = np.ascontiguousarray(f[:, :3], dtype=np.int64) if nf else np.zeros((0, 3), dtype=np.int64)
            rc = _lib.lib().tomo_obj_write(os.fsencode(filename), v.ctypes.data if nv else None,
                                           1 if v.dtype == np.float64 else 0, nv, f.ctypes.data if nf else None, nf,
                                           min(16, os.cpu_count() or 1))
            if rc < -1:
                raise OSError(-rc, os.strerror(-rc), filename)
            if rc != 0:
                raise _lib.TomoError("tomo_obj_write: %s" % _lib.lib().tomo_error_string(rc).decode())
            print(f"Model exported: {filename}")
            return True
        except Exception as e:
            print(f"Export failed: {e}")
            return False
