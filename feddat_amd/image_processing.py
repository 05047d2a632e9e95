"""Device-side mirror of the image half of the reference's input pipeline: HF `ViltImageProcessor` as called by
`ViltEncoderWrapper.process_inputs` (src/modeling/vilt.py:87-100; ViltProcessor(images=..., return_tensors='pt')).

    proc = ViltImageProcessor(device)
    enc = proc(images)          # list of [H, W, 3] uint8 arrays / tensors (decoded RGB) -> {'pixel_values', 'pixel_mask'}

Same defaults (shorter edge 384, longer <= 640, size_divisor 32, PIL BICUBIC, 1/255, mean = std = 0.5, zero padding) and
the same output dict, produced by `feddat_vilt_image_preprocess` (bit-exact with Pillow + transformers); the reference
runs this on the host three times per batch.  The text half (BertTokenizerFast WordPiece) needs the vocabulary file the
reference loads from ./models/bert-base-uncased and stays with the caller."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Sequence

import numpy as np
import torch

from . import lib as L

MAX_SHORTER, MAX_LONGER = 800, 1333


def resize_output_size(h: int, w: int, shorter: int = 384, size_divisor: int = 32):
    """transformers' get_resize_output_image_size for ViLT (python float arithmetic, int(x + 0.5), floor to divisor)."""
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


class ViltImageProcessor:
    model_input_names = ["pixel_values", "pixel_mask"]

    def __init__(self, device, shortest_edge: int = 384, size_divisor: int = 32, pad_to=None):
        """pad_to=(H, W): pad every batch to a fixed frame (the static shape the engine was built for) instead of the
        batch maximum."""
        L.load()
        self.device = torch.device(device)
        self.shortest_edge, self.size_divisor, self.pad_to = shortest_edge, size_divisor, pad_to
        self._ws = None

    def __call__(self, images: Sequence, return_tensors: str = "pt") -> Dict[str, torch.Tensor]:
        arrs: List[np.ndarray] = []
        for im in images:
            a = im.cpu().numpy() if isinstance(im, torch.Tensor) else np.asarray(im)
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
                raise L.FeddatHipError("images must be decoded RGB uint8 arrays of shape [H, W, 3]")
            arrs.append(np.ascontiguousarray(a))
        n = len(arrs)
        hs = np.array([a.shape[0] for a in arrs], np.int32)
        ws = np.array([a.shape[1] for a in arrs], np.int32)
        out = [resize_output_size(int(h), int(w), self.shortest_edge, self.size_divisor) for h, w in zip(hs, ws)]
        oh = np.array([o[0] for o in out], np.int32)
        ow = np.array([o[1] for o in out], np.int32)
        Hm, Wm = (int(oh.max()), int(ow.max())) if self.pad_to is None else self.pad_to
        if int(oh.max()) > Hm or int(ow.max()) > Wm:
            raise L.FeddatHipError(f"resized image {int(oh.max())}x{int(ow.max())} exceeds the pad_to frame {Hm}x{Wm}")
        sizes = np.array([a.size for a in arrs], np.int64)
        offs = np.zeros(n, np.int64)
        offs[1:] = np.cumsum(sizes)[:-1]
        packed = torch.from_numpy(np.concatenate([a.reshape(-1) for a in arrs])).to(self.device, non_blocking=True)
        ip = lambda a: a.ctypes.data_as(C.c_void_p)
        need = int(L.load().feddat_vilt_image_workspace_bytes(ip(hs), ip(ws), ip(oh), ip(ow), n))
        if need < 0:
            raise L.FeddatHipError("feddat_vilt_image_workspace_bytes rejected the image sizes")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 1), dtype=torch.uint8, device=self.device)
        px = torch.empty(n, 3, Hm, Wm, device=self.device)
        pm = torch.empty(n, Hm, Wm, dtype=torch.int64, device=self.device)
        rc = L.load().feddat_vilt_image_preprocess(packed.data_ptr(), ip(offs), ip(hs), ip(ws), ip(oh), ip(ow), n, Hm, Wm,
                                                   px.data_ptr(), pm.data_ptr(), self._ws.data_ptr(), self._ws.numel(),
                                                   torch.cuda.current_stream().cuda_stream)
        if rc:
            raise L.FeddatHipError(f"feddat_vilt_image_preprocess failed with code {rc}")
        return {"pixel_values": px, "pixel_mask": pm}
