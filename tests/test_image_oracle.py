"""oracle/image_oracle.py against the fixture captured from Pillow + transformers' ViltImageProcessor (the processor the
reference calls in src/modeling/vilt.py:87-100): bit-exact."""
import zlib

import numpy as np

from oracle import image_oracle as IO
from tests.golden_util import load

CASES = {"mixed": [(480, 640), (333, 500), (600, 300), (100, 150)], "big": [(1200, 1600), (900, 675)],
         "tiny": [(37, 211), (384, 384), (50, 40)]}


def test_size_rule():
    assert IO.resize_output_size(480, 640) == (384, 512)
    assert IO.resize_output_size(600, 300) == (608, 320)       # longer edge capped at 640, then floored to /32
    assert IO.resize_output_size(37, 211) == (96, 608)
    assert IO.resize_output_size(384, 384) == (384, 384)


def test_image_oracle_matches_pillow_and_hf_fixture(golden_dir):
    g = load(golden_dir, "g7_images.npz")
    for name, shapes in CASES.items():
        imgs = IO.synthetic_images(shapes, seed=zlib.crc32(name.encode()))
        nh, nw = IO.resize_output_size(*imgs[0].shape[:2])
        r = IO.pil_bicubic_resize(imgs[0], nh, nw)
        assert zlib.crc32(r.tobytes()) == int(g[f"{name}.resize0_crc"][0]), name
        px, pm = IO.vilt_image_processor(imgs)
        assert tuple(px.shape) == tuple(g[f"{name}.shape"])
        assert zlib.crc32(np.ascontiguousarray(px).tobytes()) == int(g[f"{name}.px_crc"][0]), name
        assert zlib.crc32(np.ascontiguousarray(pm).tobytes()) == int(g[f"{name}.pm_crc"][0]), name
        assert np.array_equal(px.reshape(-1)[::997], g[f"{name}.px_sample"])
        assert np.array_equal(pm.sum((1, 2)), g[f"{name}.pm_sum"])
