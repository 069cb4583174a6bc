"""Drop-in for the reference's volume_calculator.py (/root/reference/volume_calculator.py:10-132; SURVEY.md 8f row
N1): same class, same methods, same prints.  The two reductions over the volume -- per-slice voxel counts and the
bounding box -- run on the bit-packed copy that is already resident on the MI355X (1 bit/voxel read once) instead of
np.sum per slice and np.where (24 B per set voxel); the float arithmetic on top of them is done on the host in the
reference's own order, so every returned number is bit-identical.
"""
import numpy as np

from . import pipeline
from .voxel_processor import to_device_volume


def _is_bool_volume(a):
    return isinstance(a, np.ndarray) and a.dtype == np.bool_ and a.ndim == 3


class VolumeCalculator:
    """Handles volume calculations and object property analysis (reference: volume_calculator.py:10)."""

    def __init__(self):
        pass

    def calculate_voxel_volume(self, voxel_data: np.ndarray, mm_per_pixel_x: float,
                               mm_per_pixel_y: float, mm_per_slice: float) -> float:
        """volume_calculator.py:16-21."""
        voxel_volume = mm_per_pixel_x * mm_per_pixel_y * mm_per_slice
        if _is_bool_volume(voxel_data):
            total = np.int64(int(pipeline.popcount_async(to_device_volume(voxel_data)).item()))
        else:
            total = np.sum(voxel_data)              # values other than 0/1: plain host sum, not the device path
        return total * voxel_volume

    def calculate_voxel_volume_variable_depth(self, voxel_data: np.ndarray, mm_per_pixel_x: float,
                                              mm_per_pixel_y: float, slice_depths: np.ndarray) -> float:
        """volume_calculator.py:23-35: sum over z of count(z) * (mm_x * mm_y * depth[z]), accumulated in z order."""
        if len(slice_depths) == 0:
            return 0.0
        n = min(voxel_data.shape[0], len(slice_depths))
        if _is_bool_volume(voxel_data):
            counts = pipeline.slice_counts(to_device_volume(voxel_data)).cpu().numpy()
        else:
            counts = [np.sum(voxel_data[z]) for z in range(n)]
        total_volume = 0.0
        for z in range(n):
            slice_volume = mm_per_pixel_x * mm_per_pixel_y * slice_depths[z]
            total_volume += counts[z] * slice_volume
        return total_volume

    @staticmethod
    def _box(voxel_data):
        """(zmin, zmax, ymin, ymax, xmin, xmax) as np.int64, or None for an empty volume."""
        if _is_bool_volume(voxel_data):
            b = pipeline.bounding_box(to_device_volume(voxel_data))
            return None if b is None else tuple(np.int64(v) for v in b)
        z, y, x = np.where(voxel_data)
        if len(z) == 0:
            return None
        return z.min(), z.max(), y.min(), y.max(), x.min(), x.max()

    def calculate_bounding_box(self, voxel_data: np.ndarray, mm_per_pixel_x: float,
                               mm_per_pixel_y: float, mm_per_slice: float) -> dict:
        """volume_calculator.py:37-57 (an empty volume raises ValueError there as well: min of an empty array)."""
        b = self._box(voxel_data)
        if b is None:
            raise ValueError("zero-size array to reduction operation minimum which has no identity")
        zmin, zmax, ymin, ymax, xmin, xmax = b
        bbox_x = (xmin * mm_per_pixel_x, xmax * mm_per_pixel_x)
        bbox_y = (ymin * mm_per_pixel_y, ymax * mm_per_pixel_y)
        bbox_z = (zmin * mm_per_slice, zmax * mm_per_slice)
        bbox_dimensions = (bbox_x[1] - bbox_x[0], bbox_y[1] - bbox_y[0], bbox_z[1] - bbox_z[0])
        return {'x': bbox_x, 'y': bbox_y, 'z': bbox_z, 'dimensions': bbox_dimensions}

    def calculate_bounding_box_variable_depth(self, voxel_data: np.ndarray, mm_per_pixel_x: float,
                                              mm_per_pixel_y: float, slice_depths: np.ndarray) -> dict:
        """volume_calculator.py:59-94."""
        b = self._box(voxel_data)
        if b is None or len(slice_depths) == 0:
            return {'x': (0, 0), 'y': (0, 0), 'z': (0, 0), 'dimensions': (0, 0, 0)}
        zmin, zmax, ymin, ymax, xmin, xmax = b
        bbox_x = (xmin * mm_per_pixel_x, xmax * mm_per_pixel_x)
        bbox_y = (ymin * mm_per_pixel_y, ymax * mm_per_pixel_y)
        cumulative_depths = np.cumsum(np.concatenate([[0], slice_depths]))
        z_min = cumulative_depths[zmin]
        z_max = cumulative_depths[min(zmax + 1, len(cumulative_depths) - 1)]
        bbox_z = (z_min, z_max)
        bbox_dimensions = (bbox_x[1] - bbox_x[0], bbox_y[1] - bbox_y[0], bbox_z[1] - bbox_z[0])
        return {'x': bbox_x, 'y': bbox_y, 'z': bbox_z, 'dimensions': bbox_dimensions}

    def calculate_density(self, volume: float, x_length_mm: float,
                          y_length_mm: float, total_depth_mm: float) -> float:
        """volume_calculator.py:96-100."""
        total_possible_volume = x_length_mm * y_length_mm * total_depth_mm
        return volume / total_possible_volume

    def analyze_object_properties(self, voxel_data: np.ndarray, processed_volume: float,
                                  mesh_volume: float, surface_area: float,
                                  mm_per_pixel_x: float, mm_per_pixel_y: float,
                                  slice_depths: np.ndarray, x_length_mm: float,
                                  y_length_mm: float, total_depth_mm: float) -> dict:
        """volume_calculator.py:102-132."""
        voxel_volume = self.calculate_voxel_volume_variable_depth(voxel_data, mm_per_pixel_x, mm_per_pixel_y, slice_depths)
        bbox_info = self.calculate_bounding_box_variable_depth(voxel_data, mm_per_pixel_x, mm_per_pixel_y, slice_depths)
        primary_volume = mesh_volume if mesh_volume is not None else processed_volume
        total_actual_depth = np.sum(slice_depths)
        density = self.calculate_density(primary_volume, x_length_mm, y_length_mm, total_actual_depth)
        print(f"Volume: {primary_volume:.4f} mm³")
        print(f"Dimensions: {bbox_info['dimensions'][0]:.2f} x {bbox_info['dimensions'][1]:.2f} x {bbox_info['dimensions'][2]:.2f} mm")
        if surface_area:
            print(f"Surface Area: {surface_area:.4f} mm²")
        print(f"Density: {100*density:.1f}% of total space")
        return {
            'volume_mm3': primary_volume,
            'voxel_volume_mm3': voxel_volume,
            'processed_voxel_volume_mm3': processed_volume,
            'mesh_volume_mm3': mesh_volume,
            'bounding_box': {'x': bbox_info['x'], 'y': bbox_info['y'], 'z': bbox_info['z']},
            'dimensions': bbox_info['dimensions'],
            'surface_area_mm2': surface_area,
            'density': density
        }
