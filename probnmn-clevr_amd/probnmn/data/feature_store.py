"""Image-feature ingest: a page-locked fp32 feature store on the host and a prefetching batch loader
(SURVEY 8f-1).

The reference reads one float64 row per example from HDF5 on the host, casts it, lets the DataLoader
collate a batch and moves it to the GPU inside ``_Trainer.step`` (reference: probnmn/data/readers.py:63-108,
datasets.py:137-142,222-228, trainers/_trainer.py:272-287) -- at ~10 questions/s that does not matter; at
28 k questions/s it is 22.6 GB/s of features and the first bottleneck.  Here:

  * :class:`PinnedFeatureStore` holds all features ONCE as fp32 in page-locked host memory (288 GB of HBM
    hold 70 000 CLEVR images at 14x14; the host store is the general case), filled in chunks from any
    array-like (numpy array, ``np.memmap``, an h5py dataset -- whatever ``__getitem__`` with a slice
    returns; float64 sources are cast once, here);
  * ``gather(indices)`` is ONE kernel launch (``pnmn_gather_features``): the GPU reads the selected rows
    over PCIe out of the pinned store and writes them in the NHWC layout the stem reads -- no host gather,
    no staging copy, no per-row memcpy, no separate layout pass (the engine takes a ``channels_last``
    tensor in place);
  * :class:`PrefetchingLoader` issues the gather of batch k+1 on its own stream while the trainer runs
    batch k, and hands batches over with the stream dependency already in place.
"""
from typing import Dict, Iterable, Iterator, Optional

import os

import numpy as np
import torch

from probnmn import _hip


class PinnedFeatureStore:
    def __init__(self, features, chunk_rows: int = 256):
        """``features``: array-like of shape (N, C, H, W), float32 or float64."""
        shape = tuple(int(d) for d in features.shape)
        if len(shape) != 4:
            raise ValueError("features must be (N, C, H, W), got %s" % (shape,))
        self.shape = shape
        self.store = torch.empty(shape, dtype=torch.float32).pin_memory()
        view = self.store.numpy()
        for lo in range(0, shape[0], chunk_rows):  # bounded temporaries for out-of-core sources
            hi = min(shape[0], lo + chunk_rows)
            view[lo:hi] = np.asarray(features[lo:hi], dtype=np.float32)

    def __len__(self) -> int:
        return self.shape[0]

    @property
    def image_feature_size(self):
        return self.shape[1:]

    def gather(self, indices: torch.Tensor, device: torch.device, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Features of ``indices`` as a (n, C, H, W) ``channels_last`` device tensor (physically NHWC), on
        the current stream of ``device``.  ``indices``: int64, on the host (validated here) or already on
        the device (the caller vouches for the range; the kernel clamps)."""
        if device.type != "cuda":
            raise _hip.HipLibraryError("the feature store feeds a ROCm device, got %s" % device)
        n, (N, C, H, W) = int(indices.numel()), self.shape
        if indices.device.type == "cpu":
            if n and (int(indices.min()) < 0 or int(indices.max()) >= N):
                raise IndexError("feature index out of range [0, %d)" % N)
            indices = _hip.small_to_device(indices.to(torch.long).tolist(), torch.long, device)
        if out is None:
            out = torch.empty((n, C, H, W), dtype=torch.float32, device=device, memory_format=torch.channels_last)
        elif tuple(out.shape) != (n, C, H, W) or not out.is_contiguous(memory_format=torch.channels_last):
            raise ValueError("`out` must be a channels_last (n, C, H, W) tensor")
        _hip.check(_hip.lib().pnmn_gather_features(self.store.data_ptr(), indices.data_ptr(), out.data_ptr(), n, N, C,
                                                   H * W, _hip.stream_ptr(device)), "gather_features")
        return out


    def copy_rows(self, indices: torch.Tensor, device: torch.device, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Features of ``indices`` (HOST int64) as a contiguous (n, C, H, W) device tensor, moved by the copy engines
        (``pnmn_copy_rows_h2d``: one asynchronous copy per row, queued by the library) on the current stream of
        ``device`` -- no compute unit takes part, so a training step running beside it is not slowed down."""
        if device.type != "cuda":
            raise _hip.HipLibraryError("the feature store feeds a ROCm device, got %s" % device)
        if indices.device.type != "cpu":
            raise ValueError("copy_rows takes host indices (the copies are queued by the host)")
        idx = np.ascontiguousarray(indices.to(torch.long).numpy())
        n, (N, C, H, W) = int(idx.size), self.shape
        if out is None:
            out = torch.empty((n, C, H, W), dtype=torch.float32, device=device)
        elif tuple(out.shape) != (n, C, H, W) or not out.is_contiguous():
            raise ValueError("`out` must be a contiguous (n, C, H, W) tensor")
        # (checked here: the library queues row by row, so a bad index found there would leave `out` half overwritten, and
        # its -1 is PNMN_EINVAL for every kind of bad argument)
        if n and (int(idx.min()) < 0 or int(idx.max()) >= N):
            raise IndexError("feature index out of range [0, %d)" % N)
        _hip.check(_hip.lib().pnmn_copy_rows_h2d(self.store.data_ptr(), idx.ctypes.data, out.data_ptr(), n, N, C * H * W * 4,
                                                 _hip.stream_ptr(device)), "copy_rows_h2d")
        return out


class PrefetchingLoader:
    """Wraps an iterable of host-side batches ``{"image_index": LongTensor[B], ...other CPU tensors}`` and
    yields device batches with ``"image"`` filled from the store, one batch ahead: while the trainer works
    on batch k, the gather of batch k+1 runs on this loader's stream (PCIe reads beside compute).  Token
    tensors are small and go up with the batch; ``supervision`` and, if ``keep_on_host`` names it,
    ``program`` stay on the host (they drive host-side scheduling)."""

    def __init__(self, batches: Iterable[Dict[str, torch.Tensor]], store: PinnedFeatureStore, device: torch.device,
                 keep_on_host=("supervision",), method: str = "kernel"):
        """``method``: "kernel" (default): a dozen persistent workgroups read the rows over PCIe and write the NHWC batch
        the stem uses in place (``PinnedFeatureStore.gather``); "dma": one copy-engine transfer per row into a plain
        NCHW batch (``copy_rows``).  Measured beside the 1024-question joint step (bench.py: joint_training_ingest):
        39.5 ms per step with the kernel, 51.4 with the copy engines, 32.4 with resident features."""
        if method not in ("dma", "kernel"):
            raise ValueError("method must be 'dma' or 'kernel'")
        self.method = method
        self.batches, self.store, self.device = batches, store, device
        self.keep_on_host = set(keep_on_host)
        # (high priority: the ingest is a trickle of long-latency PCIe reads -- or copy-engine transfers -- that must
        # not queue behind the step's thousands of workgroups)
        self.stream = torch.cuda.Stream(device=device, priority=-1)
        self._buffers = [None, None]  # two image buffers: the one in use and the one being filled

    def _stage(self, host_batch, slot: int):
        idx = host_batch["image_index"]
        n = int(idx.numel())
        C, H, W = self.store.image_feature_size
        buf = self._buffers[slot]
        if buf is None or buf.size(0) < n:
            fmt = torch.channels_last if self.method == "kernel" else torch.contiguous_format
            buf = torch.empty((n, C, H, W), dtype=torch.float32, device=self.device, memory_format=fmt)
            self._buffers[slot] = buf
        with torch.cuda.stream(self.stream):
            # token tensors go up through the page-locked staging ring: a `.to(device)` from PAGEABLE memory is a
            # synchronous copy that first waits for everything this stream waits on -- i.e. for the step the compute
            # stream is still working through -- and the host loses the run-ahead the whole pipeline relies on
            # (measured: the ingest-fed 1024-question step 40.3 ms instead of 32.4 + nothing)
            out = {k: (v if k in self.keep_on_host else self._upload(v))
                   for k, v in host_batch.items() if k != "image_index"}
            if self.method == "kernel":
                out["image"] = self.store.gather(idx, self.device, out=buf[:n])
            else:
                out["image"] = self.store.copy_rows(idx, self.device, out=buf[:n])
            ready = torch.cuda.Event()
            ready.record(self.stream)
        return out, ready

    def _upload(self, v: torch.Tensor) -> torch.Tensor:
        if not isinstance(v, torch.Tensor) or v.is_cuda:
            return v
        if v.is_pinned():
            return v.to(self.device, non_blocking=True)
        host = v.contiguous()
        raw = host.view(torch.uint8).numpy().reshape(-1) if host.dtype != torch.bool else host.numpy().view(np.uint8).reshape(-1)
        if raw.size == 0:
            return torch.empty(host.shape, dtype=host.dtype, device=self.device)
        return _hip.to_device(raw, self.device).view(host.dtype).view(host.shape)

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        it = iter(self.batches)
        slot = 0
        try:
            pending = self._stage(next(it), slot)
        except StopIteration:
            return
        while pending is not None:
            batch, ready = pending
            nxt = next(it, None)
            slot ^= 1
            # the buffer about to be refilled was read by the step before last: order the refill behind
            # everything the compute stream has queued so far
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            pending = self._stage(nxt, slot) if nxt is not None else None
            torch.cuda.current_stream(self.device).wait_event(ready)
            for v in batch.values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(torch.cuda.current_stream(self.device))
            yield batch
