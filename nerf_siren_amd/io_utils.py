"""On-disk formats next to the path: PFM depth maps as eval.py:176-180 writes them
(datasets/depth_utils.py:5-69).  Host-side byte work; PNG/GIF (imageio, cv2) stay out of scope."""
from __future__ import annotations

import sys

import numpy as np


def save_pfm(filename, image, scale=1):
    """depth_utils.py:43-69: float32 (H,W), (H,W,1) or (H,W,3); rows stored bottom-up; the sign of the scale line
    carries the byte order (negative = little-endian)."""
    image = np.asarray(image)
    if image.dtype != np.float32:
        raise Exception('Image dtype must be float32.')
    if image.ndim == 3 and image.shape[2] == 3:
        tag = b'PF\n'
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        tag = b'Pf\n'
    else:
        raise Exception('Image must have H x W x 3, H x W x 1 or H x W dimensions.')
    order = image.dtype.byteorder
    little = order == '<' or (order == '=' and sys.byteorder == 'little')
    header = tag + f'{image.shape[1]} {image.shape[0]}\n'.encode() + (b'%f\n' % (-scale if little else scale))
    with open(filename, 'wb') as f:
        f.write(header)
        f.write(np.ascontiguousarray(image[::-1]).tobytes())


def read_pfm(filename):
    """depth_utils.py:5-40 -> (data (H,W[,3]) float32 top-down, scale)."""
    with open(filename, 'rb') as f:
        tag = f.readline().decode('utf-8').rstrip()
        if tag not in ('PF', 'Pf'):
            raise Exception('Not a PFM file.')
        dims = f.readline().decode('utf-8').split()
        if len(dims) != 2 or not all(d.isdigit() for d in dims):
            raise Exception('Malformed PFM header.')
        width, height = int(dims[0]), int(dims[1])
        scale = float(f.readline().rstrip())
        endian = '<' if scale < 0 else '>'
        data = np.frombuffer(f.read(), dtype=endian + 'f4')
    shape = (height, width, 3) if tag == 'PF' else (height, width)
    return np.flipud(data.reshape(shape)), abs(scale)
