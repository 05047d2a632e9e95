"""Writes tests/golden/g7_images.npz: what Pillow + transformers' ViltImageProcessor (the processor the reference calls
in src/modeling/vilt.py:87-100) produce for seeded uint8 images -- checksums of the full outputs plus strided samples, so
the fixture stays small.  Run in the build container: python oracle/make_image_golden.py"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.image_oracle import synthetic_images  # noqa: E402

CASES = {"mixed": [(480, 640), (333, 500), (600, 300), (100, 150)], "big": [(1200, 1600), (900, 675)],
         "tiny": [(37, 211), (384, 384), (50, 40)]}


def main():
    from PIL import Image
    from transformers import ViltImageProcessor
    ip = ViltImageProcessor()
    rec = {}
    for name, shapes in CASES.items():
        imgs = synthetic_images(shapes, seed=zlib.crc32(name.encode()))
        out = ip(images=[Image.fromarray(i) for i in imgs], return_tensors="np")
        px, pm = out["pixel_values"], out["pixel_mask"]
        rec[f"{name}.shape"] = np.array(px.shape, np.int64)
        rec[f"{name}.px_crc"] = np.array([zlib.crc32(np.ascontiguousarray(px).tobytes())], np.int64)
        rec[f"{name}.pm_crc"] = np.array([zlib.crc32(np.ascontiguousarray(pm).tobytes())], np.int64)
        rec[f"{name}.px_sample"] = px.reshape(-1)[::997].copy()
        rec[f"{name}.pm_sum"] = pm.sum((1, 2))
        # Pillow alone, first image: the uint8 resize result
        from oracle.image_oracle import resize_output_size
        nh, nw = resize_output_size(*imgs[0].shape[:2])
        r = np.asarray(Image.fromarray(imgs[0]).resize((nw, nh), resample=Image.BICUBIC))
        rec[f"{name}.resize0_crc"] = np.array([zlib.crc32(r.tobytes())], np.int64)
        print(name, px.shape, rec[f"{name}.px_crc"])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g7_images.npz"), **rec)


if __name__ == "__main__":
    main()
