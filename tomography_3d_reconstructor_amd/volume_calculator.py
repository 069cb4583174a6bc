"""Drop-in for the reference's volume_calculator.py (/root/reference/volume_calculator.py:10-132; SURVEY.md 8f row
N1), written against its contract: class and method names, argument order, the keys of the returned dicts and the four
report lines.  The two reductions over the volume -- voxel counts per slice and the index bounding box -- run on the
bit-packed copy that is already resident on the MI355X (tomo_slice_popcounts / tomo_bbox: 1 bit per voxel read once,
where np.sum per slice + np.where move 1 B per voxel and 24 B per set voxel); everything after them is a handful of
float64 operations on the host, done in the order that makes every returned number bit-identical to the reference's
(tests/golden/consumers.npz).

The host arithmetic is exposed as plain functions of (slice counts, index box) so that a Z-slab job can feed it the
all-gathered per-rank reductions (slab.SlabJob.voxel_volume / bounding_box) and get the very same numbers.
"""
import numpy as np

from . import pipeline
from .voxel_processor import to_device_volume

_EMPTY_BOX = {'x': (0, 0), 'y': (0, 0), 'z': (0, 0), 'dimensions': (0, 0, 0)}


# ----------------------------------------------------------------------------- host arithmetic on the reductions
def volume_from_slice_counts(counts, mm_per_pixel_x, mm_per_pixel_y, slice_depths):
    """Volume in mm^3 of a stack whose slice z holds counts[z] set voxels of depth slice_depths[z] (:23-35).
    The reference adds count * (mm_x * mm_y * depth) slice by slice into a float; a cumulative sum performs the same
    additions in the same order, so the last partial sum is that float."""
    n = min(len(counts), len(slice_depths))
    if n == 0:
        return 0.0
    depth = np.asarray(slice_depths, dtype=np.float64)[:n]
    per_slice = np.asarray(counts[:n], dtype=np.int64) * (mm_per_pixel_x * mm_per_pixel_y * depth)
    return np.cumsum(per_slice)[-1]


def _span(lo, hi):
    return (lo, hi), hi - lo


def box_record(extent_x, extent_y, extent_z):
    """The dict both bounding-box methods return, from the three (low, high) extents in mm."""
    (x, dx), (y, dy), (z, dz) = _span(*extent_x), _span(*extent_y), _span(*extent_z)
    return {'x': x, 'y': y, 'z': z, 'dimensions': (dx, dy, dz)}


def box_uniform_depth(box, mm_per_pixel_x, mm_per_pixel_y, mm_per_slice):
    """:37-57 for an index box (zmin, zmax, ymin, ymax, xmin, xmax) of np.int64."""
    z0, z1, y0, y1, x0, x1 = box
    return box_record((x0 * mm_per_pixel_x, x1 * mm_per_pixel_x), (y0 * mm_per_pixel_y, y1 * mm_per_pixel_y),
                      (z0 * mm_per_slice, z1 * mm_per_slice))


def box_variable_depth(box, mm_per_pixel_x, mm_per_pixel_y, slice_depths):
    """:59-94: x / y as above, z from the running sum of the slice depths -- the lower face of the first occupied slice
    to the upper face of the last one (clamped to the table)."""
    if box is None or len(slice_depths) == 0:
        return dict(_EMPTY_BOX)
    z0, z1, y0, y1, x0, x1 = box
    edges = np.cumsum(np.concatenate([[0], slice_depths]))
    top = min(z1 + 1, len(edges) - 1)
    return box_record((x0 * mm_per_pixel_x, x1 * mm_per_pixel_x), (y0 * mm_per_pixel_y, y1 * mm_per_pixel_y),
                      (edges[z0], edges[top]))


# ----------------------------------------------------------------------------- the two reductions
def _on_device(a):
    return isinstance(a, np.ndarray) and a.dtype == np.bool_ and a.ndim == 3


def slice_counts(voxel_data):
    """np.sum(voxel_data[z]) for every z -> int64 (nz,)."""
    if _on_device(voxel_data):
        return pipeline.slice_counts(to_device_volume(voxel_data)).cpu().numpy()
    return np.asarray([np.sum(voxel_data[z]) for z in range(voxel_data.shape[0])])      # other dtypes: plain host sums


def index_box(voxel_data):
    """min / max index of the set voxels per axis as (zmin, zmax, ymin, ymax, xmin, xmax) of np.int64; None if empty."""
    if _on_device(voxel_data):
        b = pipeline.bounding_box(to_device_volume(voxel_data))
        return None if b is None else tuple(np.int64(i) for i in b)
    idx = np.nonzero(voxel_data)
    if len(idx[0]) == 0:
        return None
    return tuple(f(axis) for axis in idx for f in (np.min, np.max))


class VolumeCalculator:
    """Handles volume calculations and object property analysis (reference: volume_calculator.py:10)."""

    def __init__(self):
        pass

    def calculate_voxel_volume(self, voxel_data: np.ndarray, mm_per_pixel_x: float,
                               mm_per_pixel_y: float, mm_per_slice: float) -> float:
        """:16-21: number of set voxels times the volume of one."""
        if _on_device(voxel_data):
            n_set = np.int64(int(pipeline.popcount_async(to_device_volume(voxel_data)).item()))
        else:
            n_set = np.sum(voxel_data)
        return n_set * (mm_per_pixel_x * mm_per_pixel_y * mm_per_slice)

    def calculate_voxel_volume_variable_depth(self, voxel_data: np.ndarray, mm_per_pixel_x: float,
                                              mm_per_pixel_y: float, slice_depths: np.ndarray) -> float:
        """:23-35."""
        if len(slice_depths) == 0:
            return 0.0
        return volume_from_slice_counts(slice_counts(voxel_data), mm_per_pixel_x, mm_per_pixel_y, slice_depths)

    def calculate_bounding_box(self, voxel_data: np.ndarray, mm_per_pixel_x: float,
                               mm_per_pixel_y: float, mm_per_slice: float) -> dict:
        """:37-57.  An empty volume is an error there too (the minimum of an empty index array)."""
        box = index_box(voxel_data)
        if box is None:
            raise ValueError("zero-size array to reduction operation minimum which has no identity")
        return box_uniform_depth(box, mm_per_pixel_x, mm_per_pixel_y, mm_per_slice)

    def calculate_bounding_box_variable_depth(self, voxel_data: np.ndarray, mm_per_pixel_x: float,
                                              mm_per_pixel_y: float, slice_depths: np.ndarray) -> dict:
        """:59-94: all zeros for an empty volume or an empty depth table."""
        return box_variable_depth(index_box(voxel_data), mm_per_pixel_x, mm_per_pixel_y, slice_depths)

    def calculate_density(self, volume: float, x_length_mm: float,
                          y_length_mm: float, total_depth_mm: float) -> float:
        """:96-100: the share of the x * y * depth block the volume fills."""
        return volume / (x_length_mm * y_length_mm * total_depth_mm)

    def analyze_object_properties(self, voxel_data: np.ndarray, processed_volume: float,
                                  mesh_volume: float, surface_area: float,
                                  mm_per_pixel_x: float, mm_per_pixel_y: float,
                                  slice_depths: np.ndarray, x_length_mm: float,
                                  y_length_mm: float, total_depth_mm: float) -> dict:
        """:102-132: the report (four lines, the surface-area line only for a non-zero area) and the property dict.
        The density is taken over the summed slice depths, not over `total_depth_mm`; the mesh volume, when there is
        one, is the headline volume."""
        box = self.calculate_bounding_box_variable_depth(voxel_data, mm_per_pixel_x, mm_per_pixel_y, slice_depths)
        dims = box['dimensions']
        headline = processed_volume if mesh_volume is None else mesh_volume
        fill = self.calculate_density(headline, x_length_mm, y_length_mm, np.sum(slice_depths))
        report = [f"Volume: {headline:.4f} mm³", f"Dimensions: {dims[0]:.2f} x {dims[1]:.2f} x {dims[2]:.2f} mm"]
        if surface_area:
            report.append(f"Surface Area: {surface_area:.4f} mm²")
        report.append(f"Density: {100*fill:.1f}% of total space")
        print("\n".join(report))
        return {
            'volume_mm3': headline,
            'voxel_volume_mm3': self.calculate_voxel_volume_variable_depth(voxel_data, mm_per_pixel_x, mm_per_pixel_y,
                                                                          slice_depths),
            'processed_voxel_volume_mm3': processed_volume,
            'mesh_volume_mm3': mesh_volume,
            'bounding_box': {axis: box[axis] for axis in ('x', 'y', 'z')},
            'dimensions': dims,
            'surface_area_mm2': surface_area,
            'density': fill,
        }
