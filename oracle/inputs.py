"""Oracle (test infrastructure): the harness input path in numpy.

Restates reference train_utils/kaist_train_eval_utils.py:54-55 (`imgs.float() / 255.0`) and :59-71 (multi-scale:
`sf = img_size / max(H, W)`, `ns = ceil(x * sf / gs) * gs`, `F.interpolate(size=ns, mode='bilinear',
align_corners=False)`).  The interpolation arithmetic lives in the third-party dependency torch (unpinned in
requirements.txt; 2.10.0 installed here): ATen `area_pixel_compute_scale` (scale = in / out in float32),
`area_pixel_compute_source_index` (src = scale * (dst + 0.5) - 0.5, negative -> 0) and `guard_index_and_lambda`.
Pinned by tests/golden/inputs.npz (generated with torch in the build container by tests/golden/make_golden_inputs.py).
"""
import math

import numpy as np


def multi_scale_size(shape_hw, img_size, gs=32):
    """kaist_train_eval_utils.py:63-66 -> None when no resize happens"""
    sf = img_size / max(shape_hw)
    if sf == 1:
        return None
    return [math.ceil(x * sf / gs) * gs for x in shape_hw]


def _axis(n_in, n_out):
    f32 = np.float32
    scale = f32(n_in) / f32(n_out)
    # torch's build contracts scale*(dst+0.5)-0.5 into one fused multiply-add: a single rounding (done here in float64,
    # where the product of two float32 is exact).  The unfused form moves lambda by ~2e-6 at 32 <= src < 64.
    src = (np.float64(scale) * (np.arange(n_out, dtype=f32) + f32(0.5)).astype(np.float64) - 0.5).astype(f32)
    src = np.maximum(src, f32(0)).astype(f32)
    i0 = np.minimum(src.astype(np.int64), n_in - 1)
    i1 = i0 + (i0 < n_in - 1)
    lam = np.clip(src - i0.astype(f32), f32(0), f32(1)).astype(f32)
    return i0, i1, lam


def prepare_images(imgs, size=None):
    """uint8 [B,C,H,W] -> float32 in 0..1 (float32 input: taken as is), bilinear-resized to `size` if given"""
    x = imgs.astype(np.float32) / np.float32(255.0) if imgs.dtype == np.uint8 else imgs.astype(np.float32)
    if size is None or (size[0] == x.shape[2] and size[1] == x.shape[3]):
        return x
    y0, y1, ly = _axis(x.shape[2], size[0])
    x0, x1, lx = _axis(x.shape[3], size[1])
    one = np.float32(1)
    top = x[:, :, y0][:, :, :, x0] * (one - lx) + x[:, :, y0][:, :, :, x1] * lx
    bot = x[:, :, y1][:, :, :, x0] * (one - lx) + x[:, :, y1][:, :, :, x1] * lx
    return (top * (one - ly)[:, None] + bot * ly[:, None]).astype(np.float32)
