"""feddat_vilt_image_preprocess (HIP) against the oracle and the Pillow / transformers fixture: bit-exact."""
import zlib

import numpy as np
import pytest
import torch

from oracle import image_oracle as IO
from tests.golden_util import load

pytestmark = pytest.mark.gpu
CASES = {"mixed": [(480, 640), (333, 500), (600, 300), (100, 150)], "big": [(1200, 1600), (900, 675)],
         "tiny": [(37, 211), (384, 384), (50, 40)]}


@pytest.fixture(scope="module")
def proc():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from feddat_amd import image_processing
    return image_processing


@pytest.mark.parametrize("name", list(CASES))
def test_image_processor_bit_exact(proc, golden_dir, name):
    g = load(golden_dir, "g7_images.npz")
    imgs = IO.synthetic_images(CASES[name], seed=zlib.crc32(name.encode()))
    enc = proc.ViltImageProcessor("cuda")(imgs)
    px, pm = enc["pixel_values"].cpu().numpy(), enc["pixel_mask"].cpu().numpy()
    assert tuple(px.shape) == tuple(g[f"{name}.shape"]) and pm.dtype == np.int64
    assert zlib.crc32(np.ascontiguousarray(px).tobytes()) == int(g[f"{name}.px_crc"][0])
    assert zlib.crc32(np.ascontiguousarray(pm).tobytes()) == int(g[f"{name}.pm_crc"][0])
    rpx, rpm = IO.vilt_image_processor(imgs)
    assert np.array_equal(px, rpx) and np.array_equal(pm, rpm)


def test_image_processor_many_images_and_fixed_frame(proc):
    """More images than one descriptor group (48), ragged sizes, fixed 384 x 640 frame as the engine wants it."""
    rng = np.random.default_rng(5)
    shapes = [(int(rng.integers(60, 700)), int(rng.integers(60, 700))) for _ in range(53)]
    imgs = IO.synthetic_images(shapes, seed=9)
    enc = proc.ViltImageProcessor("cuda", pad_to=(640, 640))(imgs)
    px, pm = enc["pixel_values"].cpu().numpy(), enc["pixel_mask"].cpu().numpy()
    rpx, rpm = IO.vilt_image_processor(imgs)
    H, W = rpx.shape[2:]
    assert np.array_equal(px[:, :, :H, :W], rpx) and np.array_equal(pm[:, :H, :W], rpm)
    assert not px[:, :, H:, :].any() and not px[:, :, :, W:].any() and not pm[:, H:, :].any() and not pm[:, :, W:].any()


def test_processor_output_feeds_the_engine(proc):
    """End to end: uint8 images -> device processor -> engine forward == oracle forward on the oracle's processor output."""
    from feddat_amd import engine
    from oracle import feddat_oracle as O
    imgs = IO.synthetic_images([(480, 640), (300, 420), (375, 500), (384, 384)], seed=3)
    enc = proc.ViltImageProcessor("cuda", pad_to=(384, 640))(imgs)
    d = O.ViltDims(layers=2)
    P = O.make_params(d, ["art"], bias_std=0.02)
    b = O.synthetic_batch(4, 384, 77)
    rpx, rpm = IO.vilt_image_processor(imgs)
    px = torch.zeros(4, 3, 384, 640)
    pm = torch.zeros(4, 384, 640, dtype=torch.long)
    px[:, :, :rpx.shape[2], :rpx.shape[3]] = torch.from_numpy(rpx)
    pm[:, :rpm.shape[1], :rpm.shape[2]] = torch.from_numpy(rpm)
    b["pixel_values"], b["pixel_mask"] = px, pm
    eng = engine.ViltDatEngine(P, ["art"], "cuda", batch=4, res=(384, 640), layers=2)
    dev_b = {k: v.to("cuda") for k, v in b.items()}
    dev_b["pixel_values"], dev_b["pixel_mask"] = enc["pixel_values"], enc["pixel_mask"]
    pooled, logits = eng.forward(dev_b, "gating", "art")
    with torch.no_grad():
        rp, rl = O.vilt_forward(P, d, b, "gating", "art")
    assert (pooled.cpu() - rp).abs().max() < 3e-2 and (logits.cpu() - rl).abs().max() < 3e-2
