"""Input side of the pipelined CLIs: image files -> device batches, ahead of the encoder.

The reference's loops (infer_full.py:95-128, infer_vae.py:53-72) open, resize, normalise and run ONE image at a time on one
thread.  Here a bounded thread pool decodes (`Image.open(p).convert("RGB")`: Pillow releases the GIL inside its decoders), the
decoded uint8 pixels are copied into pinned staging buffers by the workers, cross PCIe at 1 B per sample on a SIDE stream, and
the resize (Pillow's resample, bit for bit: vt_resize_u8) + ToTensor + Normalize (vt_preprocess_u8) run there too, so batch
n + 1's input is being built while batch n is in the encoder.  Skip-and-count is preserved: a file that fails to open or decode
yields (path, exception) and the loop goes on (infer_full.py:130-132).

`host_resize=True` is the reference's own route (PIL transforms on the CPU, fp32 tensors over PCIe): same bits, slower.
"""
import collections
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


def default_workers():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(16, n))


class _PinnedPool:
    """Pinned uint8 staging buffers: taken by the decode workers, returned once the H2D copy that read them has completed.
    Allocated on demand; the feeder's window of outstanding decodes bounds how many exist."""

    def __init__(self):
        self.free = queue.LifoQueue()

    def get(self, nbytes):
        try:
            buf = self.free.get_nowait()
        except queue.Empty:
            buf = None
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(nbytes, 1 << 20) * 5 // 4, dtype=torch.uint8).pin_memory()    # (rounded up: one size class serves the run)
        return buf

    def put(self, buf):
        self.free.put(buf)


class BatchFeeder:
    """Iterates over `paths` in order and yields, per batch of up to `batch_size` successfully decoded images:

        (names, x, ready, failed)

    names: the paths of the batch's images; x: fp32 [b,3,res,res] on `pipe.device`, normalised to [-1,1]; ready: a
    torch.cuda.Event recorded on the side stream behind the last kernel that wrote x (the consumer's stream must wait on it);
    failed: [(path, exception)] for the files of this stretch that could not be opened / decoded.  A batch may be empty
    (names == [], x is None) when every file of its stretch failed.

    The consumer enqueues batch n and only then asks for batch n + 1: the side stream uploads and resizes it while the consumer's
    stream is in the encoder.  Staging buffers recycle when the copy that read them has completed.
    """

    def __init__(self, pipe, paths, batch_size, resolution, workers=None, host_resize=False, transform=None):
        self.pipe, self.paths, self.bs, self.res = pipe, list(paths), max(1, int(batch_size)), int(resolution)
        self.host_resize = bool(host_resize)
        self.transform = transform
        self.workers = workers or default_workers()
        self.pool = ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="vt-decode")
        self.side = torch.cuda.Stream(device=pipe.device)
        self.staging = _PinnedPool()
        self._busy = collections.deque()            # (event, [pinned buffers]) of uploads in flight
        self._next_submit = 0
        self._futures = collections.deque()
        self._window = self.workers + 2 * self.bs    # decoded images held at most (bounds host memory: ~3 MB each at 1024^2)

    # ---- worker side (no GPU calls) ----
    def _decode(self, p):
        from PIL import Image
        img = Image.open(p).convert("RGB")
        if self.host_resize:
            return self.transform(img)               # fp32 [3,res,res], the reference's get_image_transform
        a = np.asarray(img, dtype=np.uint8)
        h, w, _ = a.shape
        buf = self.staging.get(a.size)
        np.copyto(buf.numpy()[: a.size].reshape(h, w, 3), a)
        return buf, h, w

    def _submit_more(self):
        while self._next_submit < len(self.paths) and len(self._futures) < self._window:
            p = self.paths[self._next_submit]
            self._futures.append((p, self.pool.submit(self._decode, p)))
            self._next_submit += 1

    def _recycle(self, block=False):
        while self._busy and (block or len(self._busy) > 2 or self._busy[0][0].query()):
            ev, bufs = self._busy.popleft()
            ev.synchronize()
            for b in bufs:
                self.staging.put(b)

    def __iter__(self):
        try:
            self._submit_more()
            while self._futures:
                take = [self._futures.popleft() for _ in range(min(self.bs, len(self._futures)))]
                self._submit_more()
                names, items, failed = [], [], []
                for p, f in take:
                    try:
                        items.append(f.result())
                        names.append(p)
                    except Exception as e:  # noqa: BLE001 - skip-and-count (infer_full.py:130-132)
                        failed.append((p, e))
                self._submit_more()
                if not names:
                    yield [], None, None, failed
                    continue
                yield (names,) + self._stage(items) + (failed,)
                self._recycle()
        finally:
            self.close()

    # ---- main thread: H2D + device resize / normalise on the side stream ----
    def _stage(self, items):
        dev = self.pipe.device
        with torch.cuda.stream(self.side):
            if self.host_resize:
                x = torch.stack(items).to(dev)                                   # pageable fp32: the reference's own route
            else:
                u8 = torch.empty(len(items), self.res, self.res, 3, dtype=torch.uint8, device=dev)
                bufs = []
                for k, (buf, h, w) in enumerate(items):
                    raw = torch.empty(h, w, 3, dtype=torch.uint8, device=dev)
                    raw.copy_(buf[: h * w * 3].view(h, w, 3), non_blocking=True)
                    self.pipe.resize_u8_into(raw, u8[k], self.pipe.FILTER_BILINEAR, tag="feeder")
                    bufs.append(buf)
                x = self.pipe.normalize_u8(u8)
                ev = torch.cuda.Event()
                ev.record(self.side)
                self._busy.append((ev, bufs))
            ready = torch.cuda.Event()
            ready.record(self.side)
        return x, ready

    def close(self):
        for _, f in self._futures:
            f.cancel()
        self._futures.clear()
        self.pool.shutdown(wait=True)
        self._recycle(block=True)
