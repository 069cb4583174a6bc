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

    def _depth_scale(self):
        """The ellipsoid's third semi-axis: the smaller of the two fitted ones (:83, :110)."""
        return min(self.ellipse_params['semi_major_axis'], self.ellipse_params['semi_minor_axis'])

    @staticmethod
    def _save(img, path):
        from PIL import Image
        Image.fromarray(img, mode="L").save(path)
        return path

    def generate_slices(self, num_slices: int, output_dir: str = "slices") -> List[str]:
        """(:79-105) `num_slices` heights evenly spaced over [-c, c]; the files are numbered Mask_001.png ... by
        ascending mask area (ties keep their height order), so the list runs from the empty slices to the mask itself."""
        os.makedirs(output_dir, exist_ok=True)
        c = self._depth_scale()
        images = [self._generate_slice_at_height(z, c) for z in np.linspace(-c, c, num_slices)]
        by_area = np.argsort([int(np.count_nonzero(img)) for img in images], kind="stable")
        return [self._save(images[k], os.path.join(output_dir, "Mask_%03d.png" % (pos + 1))) for pos, k in enumerate(by_area)]

    def generate_slices_half_ellipsoid(self, num_slices: int, output_dir: str = "slices",
                                       num_start: int = 28, increase: bool = True) -> List[str]:
        """(:107-143) One flank of the stack: num_slices + 2 heights evenly spaced from 0 (the mask itself) to c (the
        vanishing slice).  File Mask_Patient_<num_start> is height 0; the numbers run away from it upwards
        (increase=True) or downwards, one per height.  Files are written in ascending number order, then the two ends of
        the run -- the copy of the mask and the empty slice -- are deleted again.  The returned list names every file
        that was written, the two deleted ones included."""
        c = self._depth_scale()
        heights = np.linspace(0, c, num_slices + 2)
        step = 1 if increase else -1
        numbered = sorted((num_start + step * k, z) for k, z in enumerate(heights))
        written = [self._save(self._generate_slice_at_height(z, c), os.path.join(output_dir, f"Mask_Patient_{number}.png"))
                   for number, z in numbered]
        for path in (written[0], written[-1]):
            os.remove(path)
        return written


def generate_slices_from_mask(mask_path, n_slices, output_directory, num_start, increase):
    """simple_generator.py:6-20: start from an EMPTY `output_directory` (an existing one is removed), then write the
    half-ellipsoid flank of `mask_path` into it.  Problems are reported on the console, never raised."""
    shutil.rmtree(output_directory, ignore_errors=False) if os.path.exists(output_directory) else None
    os.makedirs(output_directory, exist_ok=True)
    if not os.path.exists(mask_path):
        print(f"Error: Image '{mask_path}' not found.")
        return
    try:
        files = EllipsoidSliceGenerator(mask_path).generate_slices_half_ellipsoid(n_slices, output_directory, num_start, increase)
    except Exception as e:                       # noqa: BLE001 -- reported, as the reference does
        print(f"Error: {e}")
        return
    print(f"Generated {len(files)} slices in '{output_directory}'")
