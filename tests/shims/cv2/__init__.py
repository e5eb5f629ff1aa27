"""Minimal stand-in for ``cv2`` (absent from this image) -- TEST FIXTURE, not product code.  The reference only needs
the import to succeed plus, when ``-d/--depths`` is given, ``cv2.imread(path, -1)`` and ``cv2.resize(arr, (w, h))``
(scene/cameras.py:16,73; utils/camera_utils.py:18,42).  Both are implemented over PIL."""
import numpy as np
from PIL import Image

IMREAD_UNCHANGED = -1
INTER_LINEAR = 1
INTER_NEAREST = 0
INTER_AREA = 3


def imread(path, flags=1):
    try:
        img = Image.open(path)
    except (FileNotFoundError, OSError):
        return None
    arr = np.array(img)
    if arr.ndim == 3 and arr.shape[2] >= 3:        # OpenCV's channel order
        arr = arr[..., [2, 1, 0] + list(range(3, arr.shape[2]))]
    return arr


def resize(src, dsize, interpolation=INTER_LINEAR):
    w, h = int(dsize[0]), int(dsize[1])
    mode = Image.NEAREST if interpolation == INTER_NEAREST else Image.BILINEAR
    if src.ndim == 2:
        return np.array(Image.fromarray(src.astype(np.float32), mode="F").resize((w, h), mode)).astype(src.dtype)
    chans = [np.array(Image.fromarray(src[..., c].astype(np.float32), mode="F").resize((w, h), mode))
             for c in range(src.shape[2])]
    return np.stack(chans, -1).astype(src.dtype)
