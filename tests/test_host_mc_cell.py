"""CPU tier: the per-cell MC33 evaluator that the device runs (csrc/mc_cell.h, compiled for the host inside
libtomo_hip.so) against the golden cells produced by scikit-image through the reference's call."""
import ctypes
import os

import numpy as np

from tomography_3d_reconstructor_amd import _lib

G = os.path.join(os.path.dirname(__file__), "golden")
# edge -> (lower corner, upper corner, axis 0=x 1=y 2=z); corners in Lewiner order
EDGE = {0: (0, 1, 0), 1: (1, 2, 1), 2: (3, 2, 0), 3: (0, 3, 1), 4: (4, 5, 0), 5: (5, 6, 1), 6: (7, 6, 0), 7: (4, 7, 1),
        8: (0, 4, 2), 9: (1, 5, 2), 10: (2, 6, 2), 11: (3, 7, 2)}
CORNER_XYZ = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]


def eval_cell(L, row):
    v = np.ascontiguousarray(row, np.float32)
    tris = np.zeros(36, np.int8)
    centre = ctypes.c_int(0)
    nt = L.tomo_host_mc_cell(v.ctypes.data, 0.5, tris.ctypes.data, ctypes.addressof(centre))
    assert nt >= 0
    return tris[:3 * nt].astype(int), centre.value


def vertex_of(L, row, e):
    vd = row.astype(np.float64) - 0.5
    if e == 12:
        out = np.zeros(3)
        vv = np.ascontiguousarray(vd)
        L.tomo_host_mc_centre_offset(vv.ctypes.data, out.ctypes.data)
        x, y, z = out
    else:
        a, b, ax = EDGE[e]
        p = list(CORNER_XYZ[a])
        p[ax] = p[ax] + L.tomo_host_mc_edge_offset(vd[a], vd[b])
        x, y, z = p
    return np.array([z, y, x], np.float64).astype(np.float32)


def test_cells_match_skimage_bit_for_bit():
    L = _lib.lib()
    d = np.load(os.path.join(G, "mc_cells.npz"))
    vals, nv, nf, verts, faces = d["vals"], d["nv"], d["nf"], d["verts"], d["faces"]
    ov = of = 0
    for i, row in enumerate(vals):
        ev, ef = verts[ov:ov + nv[i]], faces[of:of + nf[i]]
        ov += nv[i]
        of += nf[i]
        tris, centre = eval_cell(L, row)
        assert len(tris) == 3 * nf[i], (i, len(tris) // 3, nf[i])
        # serial first-touch numbering of the reference, reproduced from the triangle list
        order = {}
        flat = []
        for e in tris:
            if e not in order:
                order[e] = len(order)
            flat.append(order[e])
        got_faces = np.asarray(flat, np.int32).reshape(-1, 3)[:, ::-1]
        assert np.array_equal(got_faces, ef), i
        assert len(order) == nv[i]
        got_verts = np.stack([vertex_of(L, row, e) for e in order]) if order else np.zeros((0, 3), np.float32)
        assert got_verts.tobytes() == ev.tobytes(), i
        assert centre == int(12 in order)
        # the design's vertex rule: edge vertices == bichromatic edges of the cube
        s = row.astype(np.float64) - 0.5 > 0
        bichromatic = {e for e, (a, b, _) in EDGE.items() if s[a] != s[b]}
        if nf[i]:
            assert {e for e in order if e != 12} == bichromatic, i
