"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's multi-resolution merge (multires.py:16-40) in numpy.

PARITY UNPINNED for the resize: the reference calls cv2.resize (INTER_LINEAR) and OpenCV is absent from this image, so its float32
path is restated from its published algorithm (half-pixel centres, source index clamped at both ends with the second tap folded
onto the last pixel, horizontal then vertical pass in float32) and cross-checked against torch's independent
``F.interpolate(mode="bilinear", align_corners=False)`` for up-sampling (tests/test_multires.py).  The select step
``where(|im1 - im2| < th * im1, im2, im1)`` is the reference's own numpy expression."""
import numpy as np


def _coords(dst, src):
    """cv2 INTER_LINEAR source taps of ``dst`` output positions over ``src`` input positions -> (i0, i1, w1 as float32)."""
    scale = src / dst
    fx = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    fx = (fx - sx.astype(np.float32)).astype(np.float32)
    lo = sx < 0
    sx[lo] = 0
    fx[lo] = 0
    hi = sx >= src - 1
    sx[hi] = src - 1
    fx[hi] = 0
    i1 = np.where(hi, sx, sx + 1)
    return sx, i1, fx


def resize_linear(im, shape):
    """float32 [h,w] -> [shape[0], shape[1]] like cv2.resize(im, shape[::-1]) (INTER_LINEAR)."""
    im = np.asarray(im, dtype=np.float32)
    ho, wo = int(shape[0]), int(shape[1])
    if im.shape == (ho, wo):
        return im.copy()
    x0, x1, fx = _coords(wo, im.shape[1])
    y0, y1, fy = _coords(ho, im.shape[0])
    a0, a1 = (np.float32(1) - fx)[None, :], fx[None, :]
    rows = im[:, x0] * a0 + im[:, x1] * a1                                      # horizontal pass, float32
    b0, b1 = (np.float32(1) - fy)[:, None], fy[:, None]
    return (rows[y0] * b0 + rows[y1] * b1).astype(np.float32)


def merge(im1, im2, th=0.02, down_sample=1):
    """multires.py:26-31."""
    im2 = np.asarray(im2, dtype=np.float32)
    im1 = resize_linear(im1, im2.shape)
    mask = np.abs(im1 - im2) < np.float32(th) * im1
    im = np.where(mask, im2, im1)
    if down_sample != 1:
        im = resize_linear(im, (im.shape[0] // down_sample, im.shape[1] // down_sample))
    return im.astype(np.float32)
