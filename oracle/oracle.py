"""CPU parity oracle -- Python face of oracle/tomo_oracle.c.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never from the product package.

Parity status: PINNED by tests/golden/*.npz (generated from the reference itself,
tests/golden/make_golden.py) -- see tests/test_oracle_golden.py.

The two classes mirror the reference's call surface
(/root/reference/voxel_processor.py:27-164, /root/reference/surface_extractor.py:28-149)
so parity tests read like calls into the reference.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile oracle/tomo_oracle.c with gcc (make)."""
    so = os.path.join(_HERE, "_build", "libtomo_oracle.so")
    src = os.path.join(_HERE, "tomo_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(os.environ.get("TOMO_ORACLE_LIB") or build())      # TOMO_ORACLE_LIB: the sanitizer build (make asan)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        L.orc_fill_holes_2d.argtypes = [u8p, u8p, ctypes.c_int, ctypes.c_int]
        L.orc_close_ends.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_smooth.argtypes = [u8p, u8p] + [ctypes.c_int] * 5
        L.orc_field.argtypes = [u8p] + [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float)]
        L.orc_marching_cubes.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_double, ctypes.POINTER(ctypes.POINTER(ctypes.c_float)),
                                         ctypes.POINTER(ctypes.POINTER(ctypes.c_int)),
                                         ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        L.orc_marching_cubes_zoff.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_double, ctypes.c_int,
                                              ctypes.POINTER(ctypes.POINTER(ctypes.c_float)),
                                              ctypes.POINTER(ctypes.POINTER(ctypes.c_int)),
                                              ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        L.orc_free.argtypes = [ctypes.c_void_p]
        L.orc_finalize_vertices.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int64, ctypes.c_int,
                                            ctypes.POINTER(ctypes.c_double), ctypes.c_int64,
                                            ctypes.POINTER(ctypes.c_double), ctypes.c_int64,
                                            ctypes.c_float, ctypes.c_float]
        L.orc_mesh_volume.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64), ctypes.c_int64]
        L.orc_mesh_volume.restype = ctypes.c_double
        _LIB = L
    return _LIB


def _u8(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _f32(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


# ----------------------------------------------------------------------------- stage functions
def fill_holes_2d(sl):
    a = np.ascontiguousarray(sl, dtype=np.uint8)
    out = np.empty_like(a)
    if lib().orc_fill_holes_2d(_u8(a), _u8(out), a.shape[0], a.shape[1]):
        raise MemoryError
    return out.astype(bool)


def close_ends(vol):
    a = np.ascontiguousarray(vol, dtype=np.uint8).copy()
    nz, ny, nx = a.shape
    if lib().orc_close_ends(_u8(a), nz, ny, nx):
        raise MemoryError
    return a.astype(bool)


def smooth(vol, iterations=3, create_manifold=True):
    a = np.ascontiguousarray(vol, dtype=np.uint8)
    out = np.empty_like(a)
    nz, ny, nx = a.shape
    if lib().orc_smooth(_u8(a), _u8(out), nz, ny, nx, int(iterations), int(bool(create_manifold))):
        raise MemoryError
    return out.astype(bool)


def field(vol, manifold=True, add_padding=True):
    """float32 field marching cubes sees (surface_extractor.py:43-55)."""
    a = np.ascontiguousarray(vol, dtype=np.uint8)
    nz, ny, nx = a.shape
    p = 1 if (manifold and add_padding) else 0
    out = np.empty((nz + 2 * p, ny + 2 * p, nx + 2 * p), np.float32)
    if lib().orc_field(_u8(a), nz, ny, nx, p, int(bool(manifold)), _f32(out)):
        raise MemoryError
    return out


def marching_cubes(vol_f32, level=0.5, z_offset=0):
    """skimage.measure.marching_cubes(volume, level) -> (verts (V,3) f32 zyx, faces (F,3) i32).

    Raises ValueError / RuntimeError in the situations the wrapper does
    (skimage/measure/_marching_cubes_lewiner.py:288-302,326-327).
    """
    v = np.ascontiguousarray(vol_f32, np.float32)
    if v.ndim != 3:
        raise ValueError("Input volume should be a 3D numpy array.")
    if min(v.shape) < 2:
        raise ValueError("Input array must be at least 2x2x2.")
    level = float(level)
    if level < v.min() or level > v.max():
        raise ValueError("Surface level must be within volume data range.")
    pv = ctypes.POINTER(ctypes.c_float)()
    pf = ctypes.POINTER(ctypes.c_int)()
    nv = ctypes.c_int64()
    nf = ctypes.c_int64()
    rc = lib().orc_marching_cubes_zoff(_f32(v), v.shape[0], v.shape[1], v.shape[2], level, int(z_offset),
                                       ctypes.byref(pv), ctypes.byref(pf), ctypes.byref(nv), ctypes.byref(nf))
    if rc:
        raise MemoryError
    try:
        if nv.value == 0:
            raise RuntimeError("No surface found at the given iso value.")
        verts = np.ctypeslib.as_array(pv, shape=(nv.value, 3)).copy()
        faces = np.ctypeslib.as_array(pf, shape=(nf.value, 3)).copy() if nf.value else np.zeros((0, 3), np.int32)
    finally:
        lib().orc_free(pv)
        lib().orc_free(pf)
    return verts, faces


def finalize_vertices(verts, slice_depths, mm_per_pixel_y, mm_per_pixel_x, manifold=True, add_padding=True):
    """surface_extractor.py:57-65,82-113 on a float32 (V,3) array (returns a new array)."""
    v = np.ascontiguousarray(verts, np.float32).copy()
    d = np.asarray(slice_depths, np.float64)
    if len(d):
        adj = np.concatenate([[d[0]], d, [d[-1]]]) if add_padding else d
        cum = np.cumsum(np.concatenate([[0], adj]))
    else:
        adj = np.zeros(0)
        cum = np.zeros(1)
    adj = np.ascontiguousarray(adj, np.float64)
    cum = np.ascontiguousarray(cum, np.float64)
    dp = ctypes.POINTER(ctypes.c_double)
    lib().orc_finalize_vertices(_f32(v), len(v), int(bool(manifold)), cum.ctypes.data_as(dp), len(cum),
                                adj.ctypes.data_as(dp), len(adj),
                                np.float32(mm_per_pixel_y), np.float32(mm_per_pixel_x))
    return v


def ensure_manifold_mesh(vertices, faces):
    """surface_extractor.py:115-126 (np.unique rows + drop faces with <3 distinct indices)."""
    uniq, inv = np.unique(vertices, axis=0, return_inverse=True)
    inv = np.asarray(inv).reshape(-1)
    nf = inv[faces]
    if len(nf):
        keep = (nf[:, 0] != nf[:, 1]) & (nf[:, 1] != nf[:, 2]) & (nf[:, 0] != nf[:, 2])
        nf = nf[keep]
    if len(nf) == 0:
        nf = np.array([])
    else:
        nf = nf.astype(np.int64)
    return uniq, nf


# ----------------------------------------------------------------------------- class mirror
class VoxelProcessor:
    """Mirror of the reference VoxelProcessor (voxel_processor.py:27-164)."""

    def __init__(self):
        self.voxel_data = None
        self.side_0_count = 0
        self.side_1_count = 0
        self.side_2_count = 0

    def create_voxel_data(self, mask_images, close_ends_flag=True, side_0_count=0, side_1_count=0, side_2_count=0):
        if not mask_images:
            raise ValueError("Load masks first, hmm.")
        self.side_0_count, self.side_1_count, self.side_2_count = side_0_count, side_1_count, side_2_count
        self.voxel_data = np.stack(mask_images, axis=0)
        if close_ends_flag:
            self.voxel_data = close_ends(self.voxel_data)
        return self.voxel_data

    def smooth_voxel_data(self, voxel_data, iterations=3, create_manifold=True):
        return smooth(voxel_data, iterations, create_manifold)

    def calculate_slice_depths(self, total_depth_mm):
        s0, s1, s2 = self.side_0_count, self.side_1_count, self.side_2_count
        total = s0 + s1 + s2
        if s1 == 0 or total == 0:
            if total == 0:
                return np.array([])
            return np.full(total, total_depth_mm / total)
        d1 = total_depth_mm / s1
        d02 = 2 * d1
        d0 = d02 / s0 if s0 > 0 else 0
        d2 = d02 / s2 if s2 > 0 else 0
        return np.array([d0] * s0 + [d1] * s1 + [d2] * s2)

    def generate_point_cloud(self, voxel_data, mm_per_pixel_x, mm_per_pixel_y, slice_depths, subsample_factor=1):
        z, y, x = np.where(voxel_data)
        if subsample_factor > 1:
            idx = np.arange(0, len(z), subsample_factor)
            z, y, x = z[idx], y[idx], x[idx]
        d = np.asarray(slice_depths, np.float64)
        cum = np.cumsum(np.concatenate([[0], d]))
        inside = z < len(d)
        zc = np.minimum(z, max(len(d) - 1, 0))
        zmm = np.where(inside, cum[zc] + (d[zc] / 2 if len(d) else 0.0), cum[-1])
        return np.column_stack([zmm, y * mm_per_pixel_y, x * mm_per_pixel_x])


class SurfaceExtractor:
    """Mirror of the reference SurfaceExtractor (surface_extractor.py:28-149)."""

    def extract_manifold_surface(self, volume_data, slice_depths, mm_per_pixel_y, mm_per_pixel_x,
                                 smooth=True, manifold=True, add_padding=True):
        try:
            f = field(volume_data, manifold, add_padding)
            verts, faces = marching_cubes(f, 0.5)
            verts = finalize_vertices(verts, slice_depths, mm_per_pixel_y, mm_per_pixel_x, manifold, add_padding)
            if manifold:
                verts, faces = ensure_manifold_mesh(verts, faces)
            return verts, faces
        except Exception:
            return None

    def calculate_mesh_volume(self, vertices, faces):
        v = np.ascontiguousarray(vertices, np.float32)
        f = np.ascontiguousarray(faces, np.int64)
        return float(lib().orc_mesh_volume(_f32(v), f.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), len(f)))

    def calculate_surface_area(self, vertices, faces):
        v0 = vertices[faces[:, 0]]
        v1 = vertices[faces[:, 1]]
        v2 = vertices[faces[:, 2]]
        cp = np.cross(v1 - v0, v2 - v0)
        return np.sum(0.5 * np.linalg.norm(cp, axis=1))


def ellipsoid_masks(nz, ny, nx):
    """Synthetic ellipsoid stack of SURVEY.md section 8(d) (same formula as tests/golden/make_golden.py)."""
    cx, cy, cz = (nx - 1) / 2.0, (ny - 1) / 2.0, (nz - 1) / 2.0
    ax, ay, az = 0.42 * nx, 0.40 * ny, 0.45 * nz
    x = np.arange(nx, dtype=np.float64)[None, :]
    y = np.arange(ny, dtype=np.float64)[:, None]
    ex = ((x - cx) / ax) ** 2
    ey = ((y - cy) / ay) ** 2
    return [(ex + ey) + ((float(z) - cz) / az) ** 2 <= 1.0 for z in range(nz)]


# ----------------------------------------------------------------------------- the callers either side
class VolumeCalculator:
    """NumPy restatement of /root/reference/volume_calculator.py:10-132 (row N1), pinned by tests/golden/consumers.npz."""

    def calculate_voxel_volume(self, voxel_data, mm_per_pixel_x, mm_per_pixel_y, mm_per_slice):
        # volume_calculator.py:19-21
        return np.sum(voxel_data) * (mm_per_pixel_x * mm_per_pixel_y * mm_per_slice)

    def calculate_voxel_volume_variable_depth(self, voxel_data, mm_per_pixel_x, mm_per_pixel_y, slice_depths):
        # volume_calculator.py:26-35: sequential accumulation over z
        if len(slice_depths) == 0:
            return 0.0
        total = 0.0
        for z in range(min(voxel_data.shape[0], len(slice_depths))):
            total += np.sum(voxel_data[z]) * (mm_per_pixel_x * mm_per_pixel_y * slice_depths[z])
        return total

    def calculate_bounding_box(self, voxel_data, mm_per_pixel_x, mm_per_pixel_y, mm_per_slice):
        # volume_calculator.py:40-57
        z, y, x = np.where(voxel_data)
        bx = (x.min() * mm_per_pixel_x, x.max() * mm_per_pixel_x)
        by = (y.min() * mm_per_pixel_y, y.max() * mm_per_pixel_y)
        bz = (z.min() * mm_per_slice, z.max() * mm_per_slice)
        return {'x': bx, 'y': by, 'z': bz, 'dimensions': (bx[1] - bx[0], by[1] - by[0], bz[1] - bz[0])}

    def calculate_bounding_box_variable_depth(self, voxel_data, mm_per_pixel_x, mm_per_pixel_y, slice_depths):
        # volume_calculator.py:62-94
        z, y, x = np.where(voxel_data)
        if len(z) == 0 or len(slice_depths) == 0:
            return {'x': (0, 0), 'y': (0, 0), 'z': (0, 0), 'dimensions': (0, 0, 0)}
        bx = (x.min() * mm_per_pixel_x, x.max() * mm_per_pixel_x)
        by = (y.min() * mm_per_pixel_y, y.max() * mm_per_pixel_y)
        cum = np.cumsum(np.concatenate([[0], slice_depths]))
        bz = (cum[z.min()], cum[min(z.max() + 1, len(cum) - 1)])
        return {'x': bx, 'y': by, 'z': bz, 'dimensions': (bx[1] - bx[0], by[1] - by[0], bz[1] - bz[0])}

    def calculate_density(self, volume, x_length_mm, y_length_mm, total_depth_mm):
        return volume / (x_length_mm * y_length_mm * total_depth_mm)       # volume_calculator.py:99-100

    def analyze_object_properties(self, voxel_data, processed_volume, mesh_volume, surface_area, mm_per_pixel_x,
                                  mm_per_pixel_y, slice_depths, x_length_mm, y_length_mm, total_depth_mm):
        # volume_calculator.py:107-132
        vv = self.calculate_voxel_volume_variable_depth(voxel_data, mm_per_pixel_x, mm_per_pixel_y, slice_depths)
        bb = self.calculate_bounding_box_variable_depth(voxel_data, mm_per_pixel_x, mm_per_pixel_y, slice_depths)
        primary = mesh_volume if mesh_volume is not None else processed_volume
        density = self.calculate_density(primary, x_length_mm, y_length_mm, np.sum(slice_depths))
        print(f"Volume: {primary:.4f} mm³")
        print(f"Dimensions: {bb['dimensions'][0]:.2f} x {bb['dimensions'][1]:.2f} x {bb['dimensions'][2]:.2f} mm")
        if surface_area:
            print(f"Surface Area: {surface_area:.4f} mm²")
        print(f"Density: {100*density:.1f}% of total space")
        return {'volume_mm3': primary, 'voxel_volume_mm3': vv, 'processed_voxel_volume_mm3': processed_volume,
                'mesh_volume_mm3': mesh_volume, 'bounding_box': {'x': bb['x'], 'y': bb['y'], 'z': bb['z']},
                'dimensions': bb['dimensions'], 'surface_area_mm2': surface_area, 'density': density}


def obj_text(vertices, faces):
    """The bytes /root/reference/obj_exporter.py:19-31 writes (row N4), as one str."""
    out = ["# Tomography reconstruction model\n", f"# {len(vertices)} vertices, {len(faces)} faces\n\n"]
    for v in vertices:
        out.append(f"v {v[0]:.6f} {v[1]:.6f} {v[2]:.6f}\n")
    out.append("\n")
    for f in faces:
        out.append(f"f {f[0]+1} {f[1]+1} {f[2]+1}\n")
    return "".join(out)


def load_masks(directory, threshold=200, load_sides=(True, True, True), read_grey=None):
    """Host restatement of /root/reference/image_loader.py:37-120 (row N2) for the loader tests: returns
    (masks list | None, side counts, file list).  `read_grey(path)` stands for cv2.imread(path, IMREAD_GRAYSCALE)
    (Pillow here: cv2 is absent; identical for 8-bit grey PNGs).  Parity with cv2 on colour PNGs: unpinned."""
    import glob
    import re
    if read_grey is None:
        from PIL import Image

        def read_grey(p):
            try:
                return np.asarray(Image.open(p).convert("L"), dtype=np.uint8)
            except Exception:
                return None

    def key(fn):
        m = re.search(r'_(-?\d+)(?:\.(\d+))?\.png$', fn, re.IGNORECASE)
        return (int(m.group(1)), int(m.group(2)) if m.group(2) else 0) if m else (0, 0)
    counts, files = [0, 0, 0], []
    for idx, side in enumerate(['Section_0', 'Section_1', 'Section_2']):
        if not load_sides[idx]:
            continue
        sp = os.path.join(directory, side)
        if not os.path.exists(sp):
            return None, tuple(counts), files
        fs = sorted(glob.glob(os.path.join(sp, "Mask_*.png")), key=key)
        files.extend(fs)
        counts[idx] = len(fs)
    masks, first = [], None
    for p in files:
        img = read_grey(p)
        if img is None:
            continue
        if first is None:
            first = img
        elif img.shape != first.shape:
            continue
        masks.append(img >= threshold)
    return (masks if masks else None), tuple(counts), files
