"""Drop-in for the reference's image_loader.py (/root/reference/image_loader.py:14-136), written
against its contract: the class with its eight attributes and four accessors, the folder rule (Section_0 / 1 / 2 under
`directory`, each optional through `load_sides`), the file rule (Mask_*.png ordered by the numeric suffix
"_<int>[.<int>].png"), the console lines, `img >= threshold`, and False for "nothing usable" or any failure.

What is different, on purpose:
  * PNGs are decoded with Pillow (OpenCV is not part of this build).  8-bit grey PNGs -- what mask files are -- decode
    to the same pixels.  Anything else (colour, palette with non-grey entries, 16-bit) goes through Pillow's "L"
    conversion, which is NOT cv2.IMREAD_GRAYSCALE (other luma weights; 16-bit is clamped, cv2 scales by 1/256):
    parity is unpinned there (no cv2 in the build container to make fixtures from) and the device copy is not cached
    for such stacks, so downstream results at least follow the host masks this loader returns.
  * the grey stack is decoded into one page-locked block, goes to the MI355X in one transfer, and `>= threshold` is
    fused with the bit packing there (tomo_pack_threshold).  The packed volume is remembered against the host stack,
    so VoxelProcessor.create_voxel_data(loader.get_mask_images(), ...) neither stacks nor uploads anything again.
    The list of 2-D boolean masks the reference's callers expect is still produced: read-only views of that one stack
    (see _devcache: write-protected arrays are what makes the remembered device copy exact).
"""
import glob
import math
import os
import re
import sys

import numpy as np
import torch

from . import _devcache, pipeline

SECTIONS = ("Section_0", "Section_1", "Section_2")
_TAIL = re.compile(r'_(-?\d+)(?:\.(\d+))?\.png$', re.IGNORECASE)
_EXACT_MODES = ("L", "1")       # Pillow modes whose "L" pixels equal cv2.IMREAD_GRAYSCALE's


def _decode(path):
    """-> (uint8 (H, W) grey image, True if the decode is pixel-exact w.r.t. OpenCV), or (None, False) if unreadable."""
    try:
        from PIL import Image
        with Image.open(path) as im:
            exact = im.mode in _EXACT_MODES
            grey = np.asarray(im if im.mode == "L" else im.convert("L"), dtype=np.uint8)
    except Exception:                       # noqa: BLE001 -- cv2.imread returns None for anything it cannot read
        return None, False
    return (grey, exact) if grey.ndim == 2 else (None, False)


class ImageLoader:
    """Handles loading and preprocessing of mask images (reference: image_loader.py:14)."""

    def __init__(self):
        self.mask_files = []
        self.mask_images = []
        self.image_width = None
        self.image_height = None
        self.num_slices = 0
        self.side_0_count = self.side_1_count = self.side_2_count = 0

    def _extract_numeric_suffix(self, filename: str) -> tuple:
        """(:27-35) sort key of a mask file: (number, interpolation index) from "..._<int>[.<int>].png"; (0, 0) for
        names without such a tail.  The part after the dot is an INTEGER index, so 9.5 sorts before 9.12."""
        hit = _TAIL.search(filename)
        if hit is None:
            return (0, 0)
        return (int(hit.group(1)), int(hit.group(2) or 0))

    # -- step 1: which files, in which order (console lines as the reference prints them)
    def _scan(self, directory, load_sides):
        """-> ordered file list, or None when an enabled section folder is missing.  The per-section counts are set as
        the scan proceeds (a failed scan leaves the counts of the sections before the missing one, as the reference does)."""
        files = []
        for k, section in enumerate(SECTIONS):
            if not load_sides[k]:
                print(f"Skipping {section} (disabled)")
                continue
            folder = os.path.join(directory, section)
            if not os.path.exists(folder):
                print(f"Folder {section} not found in {directory}")
                return None
            found = sorted(glob.glob(os.path.join(folder, "Mask_*.png")), key=self._extract_numeric_suffix)
            if not found:
                print(f"No mask images found in {section}")
                continue
            print(f"Loading {len(found)} images from {section} in numeric order")
            (a0, a1), (b0, b1) = (self._extract_numeric_suffix(found[i]) for i in (0, -1))
            print(f"  Range: {os.path.basename(found[0])} ({a0}.{a1}) → {os.path.basename(found[-1])} ({b0}.{b1})")
            files += found
            setattr(self, "side_%d_count" % k, len(found))
        return files

    # -- step 2: decode into ONE stack
    def _decode_stack(self, files, on_gpu):
        """-> (grey (n, H, W) uint8 ndarray, the same as a torch tensor when it is going to the GPU else None, all decodes
        exact?) ; n may be 0.  (Ordinary memory: on this platform the upload is as fast from pageable memory as from
        page-locked, and page-locking a stack costs ~65 ms per GiB -- _hostbuf.py.)"""
        stack = None
        n, exact = 0, True
        for path in files:
            img, ok = _decode(path)
            if img is None:
                continue                                    # unreadable: skipped
            if stack is None:                               # the first readable image fixes the slice shape
                self.image_height, self.image_width = img.shape
                shape = (len(files),) + img.shape
                stack = np.empty(shape, dtype=np.uint8)
            elif img.shape != stack.shape[1:]:
                continue                                    # other shape: skipped
            stack[n] = img
            exact = exact and ok
            n += 1
        if stack is None:
            return np.empty((0, 0, 0), np.uint8), None, True
        return stack[:n], (torch.from_numpy(stack[:n]) if on_gpu else None), exact

    def load_mask_images(self, directory: str = ".", threshold: int = 200, load_sides: list = [True, True, True]) -> bool:
        """(:37-120) True when at least one mask was loaded."""
        try:
            self.side_0_count = self.side_1_count = self.side_2_count = 0
            scanned = self._scan(directory, load_sides)
            if scanned is None:
                return False
            self.mask_files = scanned
            print(f"Found masks - Side_0: {self.side_0_count}, Side_1: {self.side_1_count}, Side_2: {self.side_2_count}")
            self.mask_images = []
            on_gpu = torch.cuda.is_available()
            grey, backing, exact = self._decode_stack(self.mask_files, on_gpu)
            self.num_slices = len(grey)
            if self.num_slices == 0:
                return False
            masks = grey >= threshold                               # one contiguous bool (n, H, W) array
            if on_gpu and exact:
                # the device copy is an optimisation of the NEXT step (create_voxel_data need not upload): if it cannot be
                # made -- out of device memory, say -- the masks are still loaded, exactly as the host-only reference's are;
                # only "no usable GPU / library" is reported, as everywhere in this package
                try:
                    dev = torch.device("cuda", torch.cuda.current_device())
                    # integer grey levels: g >= t  <=>  g >= ceil(t)
                    vol = pipeline.pack_threshold(backing.to(dev), math.ceil(threshold))
                    torch.cuda.current_stream().synchronize()
                    # write-protect the stack BEFORE the per-slice views exist: they inherit the flag, and the remembered
                    # device copy can then never differ from what the caller sees
                    _devcache.put(masks, vol)
                except pipeline._lib.TomoUnavailable:
                    raise
                except Exception as e:                              # noqa: BLE001
                    print(f"tomography_3d_reconstructor_amd: masks kept on the host only ({e})", file=sys.stderr)
            self.mask_images = [masks[i] for i in range(self.num_slices)]
            return True
        except pipeline._lib.TomoUnavailable:
            raise
        except Exception as e:                                      # noqa: BLE001 -- the reference reports and returns False
            print(f"Loading failed: {e}")
            return False

    def get_mask_images(self) -> list:
        """The loaded masks, one 2-D boolean array per slice, in stack order."""
        return self.mask_images

    def get_image_dimensions(self) -> tuple:
        """(width, height) of a slice; (None, None) before anything was loaded."""
        return self.image_width, self.image_height

    def get_num_slices(self) -> int:
        """How many slices were loaded."""
        return self.num_slices

    def get_side_counts(self) -> tuple:
        """Mask files found per section folder (Section_0, Section_1, Section_2)."""
        return self.side_0_count, self.side_1_count, self.side_2_count
