"""cv2-free counterpart of the reference's offline slice generators (/root/reference/ellipsoid_slice_generator.py:8-167,
/root/reference/simple_generator.py:6-31): from ONE mask image, synthesise the flanking slices of a
(half-)ellipsoid by scaling the mask about the fitted ellipse centre with factor sqrt(1 - (z/c)^2), and write them as
Mask_*.png files the ImageLoader reads.  Host-side data preparation -- nothing here runs on the GPU.

Same class / method names, arguments, file names and slice ordering as the reference.  PARITY UNPINNED: the reference
uses cv2.fitEllipse, cv2.warpAffine (fixed-point bilinear) and cv2.imwrite; OpenCV is not part of this build, so the
ellipse comes from second-order image moments and the scaling from Pillow's bilinear affine transform -- the masks agree
with OpenCV's up to boundary pixels, which is all the reconstruction benchmark needs.
"""
import os
import shutil
from typing import List

import numpy as np


class EllipsoidSliceGenerator:
    def __init__(self, image_path: str):
        """ellipsoid_slice_generator.py:9-13."""
        self.image_path = image_path
        self.middle_slice = self._load_and_preprocess_image()
        self.ellipse_params = self._extract_ellipse_parameters()

    def _load_and_preprocess_image(self) -> np.ndarray:
        """:15-22: grey image, threshold 127 -> 0 / 255 uint8."""
        from PIL import Image
        try:
            with Image.open(self.image_path) as im:
                img = np.asarray(im.convert("L"), dtype=np.uint8)
        except Exception:
            raise ValueError(f"Could not load image from {self.image_path}")
        return np.where(img > 127, 255, 0).astype(np.uint8)

    def _extract_ellipse_parameters(self) -> dict:
        """:24-49 with the moment ellipse of the mask's largest 4-connected component instead of cv2.fitEllipse."""
        from scipy import ndimage
        lab, n = ndimage.label(self.middle_slice > 0)
        if n == 0:
            raise ValueError("No contours found in the image")
        sizes = ndimage.sum(np.ones_like(lab), lab, index=np.arange(1, n + 1))
        comp = lab == (1 + int(np.argmax(sizes)))
        ys, xs = np.nonzero(comp)
        if len(xs) < 5:
            raise ValueError("Could not fit ellipse to the contour")
        cx, cy = xs.mean(), ys.mean()
        cov = np.cov(np.stack([xs - cx, ys - cy]))
        ev, evec = np.linalg.eigh(cov)
        a, b = 2.0 * np.sqrt(ev[1]), 2.0 * np.sqrt(ev[0])        # a uniform ellipse has variance (semi-axis)^2 / 4
        angle = float(np.degrees(np.arctan2(evec[1, 1], evec[0, 1])))
        return {'center': (float(cx), float(cy)), 'semi_major_axis': float(a), 'semi_minor_axis': float(b),
                'angle': angle, 'area': float(comp.sum())}

    def _calculate_ellipse_area_at_height(self, z: float, c: float) -> float:
        """:51-60."""
        if abs(z) > c:
            return 0.0
        factor = np.sqrt(1 - (z / c) ** 2)
        return np.pi * self.ellipse_params['semi_major_axis'] * factor * self.ellipse_params['semi_minor_axis'] * factor

    def _generate_slice_at_height(self, z: float, c: float) -> np.ndarray:
        """:62-77: the mask scaled about the ellipse centre by sqrt(1 - (z/c)^2) (z = 0: the mask itself)."""
        from PIL import Image
        if z < 0 or z > c:
            return np.zeros_like(self.middle_slice)
        factor = np.sqrt(1 - (z / c) ** 2) if c > 0 else 0
        if factor <= 0:
            return np.zeros_like(self.middle_slice)
        cx, cy = self.ellipse_params['center']
        h, w = self.middle_slice.shape
        inv = 1.0 / factor                                           # output (x, y) samples input at centre + (p - centre) / factor
        coeffs = (inv, 0.0, cx - inv * cx, 0.0, inv, cy - inv * cy)
        out = Image.fromarray(self.middle_slice, mode="L").transform((w, h), Image.AFFINE, coeffs, resample=Image.BILINEAR)
        return np.asarray(out, dtype=np.uint8)

    def generate_slices(self, num_slices: int, output_dir: str = "slices") -> List[str]:
        """:79-105: n slices over z in [-c, c], saved as Mask_001.png ... in order of increasing area."""
        from PIL import Image
        os.makedirs(output_dir, exist_ok=True)
        c = min(self.ellipse_params['semi_major_axis'], self.ellipse_params['semi_minor_axis'])
        data = []
        for i, z in enumerate(np.linspace(-c, c, num_slices)):
            img = self._generate_slice_at_height(z, c)
            data.append((i, z, img, np.sum(img > 0)))
        data.sort(key=lambda x: x[3])
        saved = []
        for number, (_, _, img, _) in enumerate(data, 1):
            path = os.path.join(output_dir, f"Mask_{number:03d}.png")
            Image.fromarray(img, mode="L").save(path)
            saved.append(path)
        return saved

    def generate_slices_half_ellipsoid(self, num_slices: int, output_dir: str = "slices",
                                       num_start: int = 28, increase: bool = True) -> List[str]:
        """:107-143: num_slices + 2 slices over z in [0, c] named Mask_Patient_<n>.png, then the two extreme files
        (the mask itself and the vanishing slice) are deleted again; the returned list still names all of them."""
        from PIL import Image
        c = min(self.ellipse_params['semi_major_axis'], self.ellipse_params['semi_minor_axis'])
        z_positions = np.linspace(0, c, num_slices + 2)
        if increase:
            num_end = num_start + 1 + num_slices
        else:
            num_end = num_start - num_slices - 1
            num_start, num_end = num_end, num_start
        saved = []
        number_range = list(range(num_start, num_end + 1))
        for i, number in enumerate(number_range):
            z_index = i if increase else len(number_range) - 1 - i
            z = z_positions[z_index] if z_index < len(z_positions) else c
            path = os.path.join(output_dir, f"Mask_Patient_{number}.png")
            Image.fromarray(self._generate_slice_at_height(z, c), mode="L").save(path)
            saved.append(path)
        os.remove(saved[0])
        os.remove(saved[-1])
        return saved


def generate_slices_from_mask(mask_path, n_slices, output_directory, num_start, increase):
    """simple_generator.py:6-20."""
    if os.path.exists(output_directory):
        shutil.rmtree(output_directory)
    os.makedirs(output_directory, exist_ok=True)
    if not os.path.exists(mask_path):
        print(f"Error: Image '{mask_path}' not found.")
        return
    try:
        generator = EllipsoidSliceGenerator(mask_path)
        slice_files = generator.generate_slices_half_ellipsoid(n_slices, output_directory, num_start, increase)
        print(f"Generated {len(slice_files)} slices in '{output_directory}'")
    except Exception as e:
        print(f"Error: {e}")
