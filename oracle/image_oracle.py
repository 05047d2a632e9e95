"""TEST INFRASTRUCTURE ONLY (see oracle/feddat_oracle.py): numpy restatement of the image half of the reference's
input pipeline -- `ViltEncoderWrapper.process_inputs` (src/modeling/vilt.py:87-100) -> HF `ViltProcessor` ->
`ViltImageProcessor` (transformers, not vendored by the reference): resize so that the shorter edge is 384 and the
longer at most 640, floor both to multiples of 32, PIL BICUBIC on uint8, rescale 1/255, normalise with mean = std =
0.5, zero-pad to the batch maximum, pixel_mask.

The resize restates Pillow's `ImagingResample` for 8-bit images (src/libImaging/Resample.c, Pillow 12.2 in this
container): separable convolution, horizontal pass first, bicubic kernel (a = -0.5) whose support is widened by the
down-scaling factor (antialiasing), coefficients normalised in double precision and quantised to 22 fractional bits,
accumulation in int32 starting from 1 << 21, arithmetic shift, clip to [0, 255] after EACH pass.

Pinned: tests/test_image_oracle.py checks this file bit-for-bit against Pillow itself and against the HF processor on
the committed fixture tests/golden/g7_images.npz (written by oracle/make_image_golden.py)."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
MAX_SHORTER, MAX_LONGER = 800, 1333      # transformers vilt image processor constants


def resize_output_size(h: int, w: int, shorter: int = 384, size_divisor: int = 32):
    """transformers get_resize_output_image_size (ViLT): python float arithmetic, int(x + 0.5), floor to divisor."""
    longer = int(MAX_LONGER / MAX_SHORTER * shorter)
    scale = shorter / min(h, w)
    if h < w:
        nh, nw = shorter, scale * w
    else:
        nh, nw = scale * h, shorter
    if max(nh, nw) > longer:
        scale = longer / max(nh, nw)
        nh, nw = scale * nh, scale * nw
    nh, nw = int(nh + 0.5), int(nw + 0.5)
    return nh // size_divisor * size_divisor, nw // size_divisor * size_divisor


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc -> (bounds [out,2] int32, coeffs [out,ksize] int32)."""
    support0 = 2.0
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img: np.ndarray, out_size: int) -> np.ndarray:
    """One horizontal pass over a [rows, in_size, C] uint8 array -> [rows, out_size, C] uint8."""
    rows, in_size, C = img.shape
    bounds, kk = precompute_coeffs(in_size, out_size)
    out = np.empty((rows, out_size, C), np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        acc = (1 << (PRECISION_BITS - 1)) + (src[:, xmin:xmin + n, :] * kk[xx, :n, None].astype(np.int64)).sum(1)
        acc = acc.astype(np.int32)            # Resample.c accumulates in int (no overflow for 8-bit data)
        out[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def pil_bicubic_resize(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """img [H, W, 3] uint8 -> [out_h, out_w, 3] uint8, == PIL.Image.fromarray(img).resize((out_w, out_h), BICUBIC)."""
    h, w, _ = img.shape
    t = img
    if out_w != w:
        t = _pass(t, out_w)
    if out_h != h:
        t = _pass(t.transpose(1, 0, 2), out_h).transpose(1, 0, 2)
    return np.ascontiguousarray(t)


def vilt_image_processor(images, shorter: int = 384, size_divisor: int = 32):
    """list of [H, W, 3] uint8 arrays -> (pixel_values float32 [B,3,Hm,Wm], pixel_mask int64 [B,Hm,Wm])."""
    outs = []
    for im in images:
        nh, nw = resize_output_size(im.shape[0], im.shape[1], shorter, size_divisor)
        r = pil_bicubic_resize(im, nh, nw).transpose(2, 0, 1)
        v = (r.astype(np.float64) * (1 / 255)).astype(np.float32)        # transforms.rescale: float64 product
        v = (v - np.float32(0.5)) / np.float32(0.5)                      # transforms.normalize in float32
        outs.append(v)
    Hm, Wm = max(o.shape[1] for o in outs), max(o.shape[2] for o in outs)
    px = np.zeros((len(outs), 3, Hm, Wm), np.float32)
    pm = np.zeros((len(outs), Hm, Wm), np.int64)
    for i, o in enumerate(outs):
        px[i, :, :o.shape[1], :o.shape[2]] = o
        pm[i, :o.shape[1], :o.shape[2]] = 1
    return px, pm


def synthetic_images(shapes, seed: int):
    """Seeded uint8 test images: noise, plus a smooth gradient and hard edges (clipping / ringing paths of the filter)."""
    rng = np.random.default_rng(seed)
    out = []
    for i, (h, w) in enumerate(shapes):
        if i % 3 == 0:
            im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        elif i % 3 == 1:
            yy, xx = np.mgrid[0:h, 0:w]
            im = np.stack([(255 * xx / max(w - 1, 1)), (255 * yy / max(h - 1, 1)), ((xx + yy) % 256)], -1).astype(np.uint8)
        else:
            im = np.zeros((h, w, 3), np.uint8)
            im[(np.arange(h)[:, None] // 7 + np.arange(w)[None, :] // 5) % 2 == 0] = 255
        out.append(im)
    return out
