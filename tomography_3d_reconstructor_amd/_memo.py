"""Device-side memo of RESULTS across the orchestrator's repeated calls.

The reference's orchestrator calls smooth_voxel_data 5-6 times and extract_manifold_surface 4-5 times with the same
arguments in one run (tomography_3d_reconstruction.py:108-134, 201-243).  What is remembered here is the DEVICE result
(the smoothed bit volume: 1 bit/voxel; the final vertices / faces tensors), keyed on the identity of the device volume
it was computed from plus the call's parameters.  Every call still hands out fresh host arrays (the reference's
ownership rule) -- measured at 1024^3 that download IS the cost of a call (extract: 5.5 ms of which the device pass is
1.2 ms; a host-side copy of the 212 MB mesh would take longer than computing and downloading it again), so only the
device work is skipped.  An entry dies with its source volume; at most `_MAX` entries per table are kept.
"""
import weakref
from collections import OrderedDict

_MAX = 2
ENABLED = True
STATS = {"hit": 0, "miss": 0}


class Table:
    def __init__(self):
        self._d = OrderedDict()

    def get(self, src, params):
        if not ENABLED:
            return None
        key = (id(src),) + tuple(params)
        ent = self._d.get(key)
        if ent is None or ent[0]() is not src:
            STATS["miss"] += 1
            self._d.pop(key, None)
            return None
        self._d.move_to_end(key)
        STATS["hit"] += 1
        return ent[1]

    def put(self, src, params, value):
        if not ENABLED:
            return
        key = (id(src),) + tuple(params)
        self._d[key] = (weakref.ref(src, lambda _r, k=key: self._d.pop(k, None)), value)
        self._d.move_to_end(key)
        while len(self._d) > _MAX:
            self._d.popitem(last=False)

    def clear(self):
        self._d.clear()


smoothed = Table()      # (source BitVolume, iterations, create_manifold) -> smoothed BitVolume
surfaces = Table()      # (source BitVolume, depths bytes, mm_y, mm_x, manifold, add_padding) -> (vertices, faces) tensors or None


def clear():
    smoothed.clear()
    surfaces.clear()
