"""Input staging for the training loop: host batches are uploaded (and, optionally, pre-processed on the device) on a
side HIP stream by a worker thread while the current train_step runs on the main stream.

The reference gets its batches from a torch DataLoader (`vqa_dataset_crossvqa.py:509-515`, num_workers=2) and moves them
to the GPU synchronously inside `process_inputs` (`vilt.py:98`, `.to(self.device)`), three times per batch.  Here the
57 MB of a B=32 fp32 pixel batch (or the ~30 MB of decoded uint8 images, with `feddat_amd.image_processing` doing the
rest on the device) cross PCIe while the previous step computes.

    pre = DevicePrefetcher(host_batches, lambda b: {k: v.to(dev, non_blocking=True) for k, v in b.items()}, dev)
    for batch in pre:            # device tensors, ready on the current stream
        engine.train_step(batch, use_graph=True)
"""
from __future__ import annotations

import queue
import threading
from typing import Callable, Iterable

import torch


def _tensors(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors(v)


class DevicePrefetcher:
    """Iterates `fn(item)` for every item of `source`, with fn running on a private stream in a worker thread, `depth`
    items ahead.  fn enqueues device work only (uploads, device pre-processing) and returns device tensors."""

    _END = object()

    def __init__(self, source: Iterable, fn: Callable, device, depth: int = 2):
        dev = torch.device(device)
        if dev.index is None:                      # 'cuda' -> the current device
            dev = torch.device("cuda", torch.cuda.current_device())
        self.source, self.fn, self.device, self.depth = source, fn, dev, depth

    def __iter__(self):
        q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        stream = torch.cuda.Stream(device=self.device)
        stop = threading.Event()

        def work():
            try:
                torch.cuda.set_device(self.device)
                for item in self.source:
                    if stop.is_set():
                        break
                    with torch.cuda.stream(stream):
                        out = self.fn(item)
                        ev = torch.cuda.Event()
                        ev.record(stream)
                    q.put((out, ev))
                q.put((self._END, None))
            except BaseException as e:      # surface worker errors in the consumer
                q.put((e, None))

        t = threading.Thread(target=work, daemon=True)
        t.start()
        try:
            while True:
                out, ev = q.get()
                if out is self._END:
                    break
                if isinstance(out, BaseException):
                    raise out
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                for x in _tensors(out):
                    if x.is_cuda:
                        x.record_stream(cur)     # allocated on the side stream, consumed on this one
                yield out
        finally:
            stop.set()
            while t.is_alive():               # unblock a producer waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    t.join(timeout=0.05)


def pin_batch(batch):
    """Page-locked copies of the CPU tensors of a batch (what DataLoader(pin_memory=True) hands over)."""
    return {k: (v.pin_memory() if isinstance(v, torch.Tensor) and not v.is_cuda else v) for k, v in batch.items()}
