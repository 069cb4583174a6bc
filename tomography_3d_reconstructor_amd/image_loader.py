"""Drop-in for the reference's image_loader.py (/root/reference/image_loader.py:14-136): same
class and accessors, same folder / file-name rules (Section_0/1/2, Mask_*.png, numeric-suffix order), same prints and
return values.  Differences, on purpose:

  * PNGs are decoded with Pillow instead of OpenCV (cv2 is not part of this build).  For 8-bit grey PNGs -- what mask
    files are -- the pixels are identical; colour / 16-bit PNGs go through Pillow's "L" conversion, which may differ
    from cv2.IMREAD_GRAYSCALE by one grey level away from pure black / white (parity unpinned there: no cv2 here to
    generate fixtures from; binary 0/255 masks are unaffected);
  * the grey stack goes to the MI355X as it is decoded and `img >= threshold` (image_loader.py:108) is fused with the
    bit packing on the device (tomo_pack_threshold); the packed volume is remembered, so
    VoxelProcessor.create_voxel_data(loader.get_mask_images(), ...) neither stacks nor uploads anything again.
    The host list of boolean masks the reference's callers expect is still produced (views of one contiguous stack).
"""
import glob
import os
import re

import numpy as np
import torch

from . import _devcache, pipeline

_SUFFIX = re.compile(r'_(-?\d+)(?:\.(\d+))?\.png$', re.IGNORECASE)


def _read_grey(path):
    """cv2.imread(path, cv2.IMREAD_GRAYSCALE) counterpart: uint8 (H, W) array, or None if the file cannot be read."""
    try:
        from PIL import Image
        with Image.open(path) as im:
            if im.mode != "L":
                im = im.convert("L")
            a = np.asarray(im, dtype=np.uint8)
        return a if a.ndim == 2 else None
    except Exception:
        return None


class ImageLoader:
    """Handles loading and preprocessing of mask images (reference: image_loader.py:14)."""

    def __init__(self):
        self.mask_files = []
        self.mask_images = []
        self.image_width = None
        self.image_height = None
        self.num_slices = 0
        self.side_0_count = 0
        self.side_1_count = 0
        self.side_2_count = 0

    def _extract_numeric_suffix(self, filename: str) -> tuple:
        """image_loader.py:27-35: (main number, interpolation index) of "..._<int>[.<int>].png", (0, 0) otherwise."""
        m = _SUFFIX.search(filename)
        if m:
            return (int(m.group(1)), int(m.group(2)) if m.group(2) else 0)
        return (0, 0)

    def load_mask_images(self, directory: str = ".", threshold: int = 200, load_sides: list = [True, True, True]) -> bool:
        """image_loader.py:37-120."""
        try:
            side_folders = ['Section_0', 'Section_1', 'Section_2']
            all_mask_files = []
            self.side_0_count = 0
            self.side_1_count = 0
            self.side_2_count = 0
            for idx, side_folder in enumerate(side_folders):
                if not load_sides[idx]:
                    print(f"Skipping {side_folder} (disabled)")
                    continue
                side_path = os.path.join(directory, side_folder)
                if not os.path.exists(side_path):
                    print(f"Folder {side_folder} not found in {directory}")
                    return False
                side_files = glob.glob(os.path.join(side_path, "Mask_*.png"))
                if not side_files:
                    print(f"No mask images found in {side_folder}")
                    continue
                side_files = sorted(side_files, key=self._extract_numeric_suffix)
                print(f"Loading {len(side_files)} images from {side_folder} in numeric order")
                first_nums = self._extract_numeric_suffix(side_files[0])
                last_nums = self._extract_numeric_suffix(side_files[-1])
                print(f"  Range: {os.path.basename(side_files[0])} ({first_nums[0]}.{first_nums[1]}) → "
                      f"{os.path.basename(side_files[-1])} ({last_nums[0]}.{last_nums[1]})")
                all_mask_files.extend(side_files)
                if idx == 0:
                    self.side_0_count = len(side_files)
                elif idx == 1:
                    self.side_1_count = len(side_files)
                else:
                    self.side_2_count = len(side_files)
            self.mask_files = all_mask_files
            print(f"Found masks - Side_0: {self.side_0_count}, Side_1: {self.side_1_count}, Side_2: {self.side_2_count}")

            self.mask_images = []
            grey = None        # (n, H, W) uint8 staging stack (pinned when a GPU is present)
            n = 0
            for file_path in self.mask_files:
                img = _read_grey(file_path)
                if img is None:
                    continue
                if grey is None:
                    self.image_height, self.image_width = img.shape
                    shape = (len(self.mask_files),) + img.shape
                    if torch.cuda.is_available():
                        grey_t = torch.empty(shape, dtype=torch.uint8, pin_memory=True)
                        grey = grey_t.numpy()
                    else:
                        grey_t = None
                        grey = np.empty(shape, dtype=np.uint8)
                elif img.shape != grey.shape[1:]:
                    continue
                grey[n] = img
                n += 1
            self.num_slices = n
            if n == 0:
                return False
            stack = grey[:n] >= threshold                       # one contiguous bool (n, H, W) array
            if torch.cuda.is_available():
                dev = torch.device("cuda", torch.cuda.current_device())
                vol = pipeline.pack_threshold(grey_t[:n].to(dev, non_blocking=True), threshold)
                torch.cuda.current_stream().synchronize()       # the pinned staging buffer is released on return
                # create_voxel_data finds the volume through the views' base; the stack is write-protected BEFORE the
                # views are made, so they are read-only too and the device copy cannot go stale
                _devcache.put(stack, vol)
            self.mask_images = [stack[i] for i in range(n)]
            return True
        except Exception as e:
            print(f"Loading failed: {e}")
            return False

    def get_mask_images(self) -> list:
        """Get loaded mask images."""
        return self.mask_images

    def get_image_dimensions(self) -> tuple:
        """Get image dimensions (width, height)."""
        return self.image_width, self.image_height

    def get_num_slices(self) -> int:
        """Get number of loaded slices."""
        return self.num_slices

    def get_side_counts(self) -> tuple:
        """Get counts for each side (Side_0, Side_1, Side_2)."""
        return self.side_0_count, self.side_1_count, self.side_2_count
