"""ORACLE (test infrastructure, not product code): numpy restatement of the image resize the reference performs
before the encoder -- `transforms.Resize((r, r))` on a PIL image (modules.py:135-140; torchvision hands a PIL image to
`Image.resize(..., BILINEAR)`) and `SmartResize` = centre crop + `Image.resize(..., Image.LANCZOS)` (modules.py:142-178).

The arithmetic lives in Pillow (third-party; requirements.txt pins none, 12.2.0 is installed here): `ImagingResample`
(libImaging/Resample.c) -- per axis, per output sample, the filter is evaluated in double precision over the support
window, normalised, converted to 22-bit fixed point, and applied in int32 with a rounding half and a clip to uint8; the
horizontal pass runs first and its uint8 result feeds the vertical pass.  PINNED: tests/test_oracle.py compares this
restatement bit for bit with Pillow itself on random images and sizes.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
BILINEAR, LANCZOS = 0, 1


def _filter(kind, x):
    if kind == BILINEAR:
        x = abs(x)
        return 1.0 - x if x < 1.0 else 0.0
    if -3.0 <= x < 3.0:
        def sinc(v):
            if v == 0.0:
                return 1.0
            v *= math.pi
            return math.sin(v) / v
        return sinc(x) * sinc(x / 3)
    return 0.0


def coefficients(in_size, out_size, kind):
    """-> (bounds [out,2] int (first sample, count), coeffs [out, ksize] int32), Pillow's precompute_coeffs +
    normalize_coeffs_8bpc for the whole axis (box = (0, in_size))."""
    support0 = 1.0 if kind == BILINEAR else 3.0
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_filter(kind, (x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)                                          # left-to-right double sum, as in C
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + img.shape[1:], dtype=np.uint8)
    for xx in range(bounds.shape[0]):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[xx, :n].astype(np.int64), img[x0:x0 + n], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize(img_u8_hwc, out_w, out_h, kind):
    """uint8 [H,W,C] -> uint8 [out_h,out_w,C]; horizontal pass first (Pillow skips a pass whose size is unchanged)."""
    h, w, _ = img_u8_hwc.shape
    x = img_u8_hwc
    if out_w != w:
        x = _pass(x, *coefficients(w, out_w, kind), axis=1)
    if out_h != h:
        x = _pass(x, *coefficients(h, out_h, kind), axis=0)
    return x


def smart_crop_box(width, height, target_w, target_h):
    """SmartResize's centre crop (modules.py:150-175): (left, top, crop_w, crop_h)."""
    tr, r = target_w / target_h, width / height
    if r > tr:
        nw = int(height * tr)
        return (width - nw) // 2, 0, nw, height
    if r < tr:
        nh = int(width / tr)
        return 0, (height - nh) // 2, width, nh
    return 0, 0, width, height
