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


class ResidentRows:
    """A batch of rows of a :class:`DeviceFeatureStore`: no data of its own -- the store and the HOST indices.  The NMN
    takes it where it takes an ``image`` tensor (``nmn(batch["image"], programs, answers)``); ``subset(rows)`` is what
    ``image[rows]`` is for a tensor."""

    def __init__(self, store: "DeviceFeatureStore", index):
        self.store = store
        self.index = np.ascontiguousarray(torch.as_tensor(index).to(torch.long).numpy() if not isinstance(index, np.ndarray)
                                          else index, dtype=np.int64)
        if self.index.ndim != 1:
            raise ValueError("a batch of rows takes a 1-d index")
        if self.index.size and (int(self.index.min()) < 0 or int(self.index.max()) >= len(store)):
            raise IndexError("feature index out of range [0, %d)" % len(store))

    @property
    def device(self) -> torch.device:
        return self.store.device

    @property
    def shape(self):
        return (int(self.index.size),) + tuple(self.store.image_feature_size)

    def size(self, d: Optional[int] = None):
        return self.shape if d is None else self.shape[d]

    def subset(self, rows) -> "ResidentRows":
        rows = torch.as_tensor(rows).cpu().numpy() if not isinstance(rows, np.ndarray) else rows
        return ResidentRows(self.store, self.index[rows])

    def pointers(self) -> np.ndarray:
        """Device address of every row's [H*W][C] map (int64, host)."""
        return self.store.data_ptr() + self.index * np.int64(self.store.row_bytes)

    def materialize(self) -> torch.Tensor:
        """The rows as an ordinary (n, C, H, W) ``channels_last`` tensor (a gathered copy: tests, evaluation code that
        wants a tensor)."""
        idx = torch.from_numpy(self.index).to(self.store.device)
        return self.store.rows[idx].permute(0, 3, 1, 2)


class DeviceFeatureStore:
    """ALL features once in HBM, in the layout the stem reads (NHWC: ``rows[N][H][W][C]`` fp32).  70 000 CLEVR train
    images x 0.8 MB = 56.2 GB at 14x14 (15 000 val images: 12 GB) of the 288 GB of an MI355X; 28x28 maps: 225 GB -- the
    train set alone still fits.  A step then moves NO feature bytes over PCIe and runs no gather or layout pass: the
    first stem convolution and its weight gradient take per-example pointers (``pnmn_conv_item.in`` /
    ``pnmn_wgrad_item.x``), which point straight at the selected rows (:class:`ResidentRows`).  What does not fit keeps
    the pinned host store (:class:`PinnedFeatureStore`) and its gather kernel.  The reference re-reads float64 rows
    from HDF5 per item (probnmn/data/readers.py:63-108, datasets.py:137-142).

    Filled in chunks from any array-like (float32 / float64, (N, C, H, W)): a chunk goes through a page-locked staging
    buffer and the ingest kernel, which writes the NHWC rows."""

    def __init__(self, features, device: torch.device, chunk_rows: int = 512):
        shape = tuple(int(d) for d in features.shape)
        if len(shape) != 4:
            raise ValueError("features must be (N, C, H, W), got %s" % (shape,))
        device = torch.device(device)
        if device.type != "cuda":
            raise _hip.HipLibraryError("the resident feature store lives on a ROCm device, got %s" % device)
        N, C, H, W = shape
        self.shape, self.device = shape, device
        self.row_bytes = C * H * W * 4
        free, _ = torch.cuda.mem_get_info(device)
        if N * self.row_bytes > free:
            raise MemoryError("%d rows x %.1f MB = %.1f GB do not fit the %.1f GB free on %s: use PinnedFeatureStore"
                              % (N, self.row_bytes / 1e6, N * self.row_bytes / 1e9, free / 1e9, device))
        self.rows = torch.empty((N, H, W, C), dtype=torch.float32, device=device)
        self.device = device = self.rows.device  # ('cuda' resolved to 'cuda:<current>': compared with parameter devices)
        chunk_rows = max(1, min(chunk_rows, N))
        stage = torch.empty((chunk_rows, C, H, W), dtype=torch.float32).pin_memory()
        view = stage.numpy()
        st = _hip.stream_ptr(device)
        for lo in range(0, N, chunk_rows):
            hi = min(N, lo + chunk_rows)
            torch.cuda.current_stream(device).synchronize()  # (the staging buffer is reused)
            view[: hi - lo] = np.asarray(features[lo:hi], dtype=np.float32)
            idx = _hip.small_to_device(list(range(hi - lo)), torch.long, device) if hi - lo <= 4096 else \
                torch.arange(hi - lo, device=device)
            _hip.check(_hip.lib().pnmn_gather_features(stage.data_ptr(), idx.data_ptr(), self.rows[lo:hi].data_ptr(), hi - lo,
                                                       chunk_rows, C, H * W, st), "gather_features")
        torch.cuda.current_stream(device).synchronize()

    @classmethod
    def from_device(cls, features: torch.Tensor) -> "DeviceFeatureStore":
        """Adopt features that are already in HBM in the stem's layout -- an (N, C, H, W) fp32 ``channels_last`` tensor,
        which is what ``probnmn.data.feature_extractor`` writes -- without a copy."""
        if features.dim() != 4 or features.dtype != torch.float32 or features.device.type != "cuda":
            raise ValueError("expected an (N, C, H, W) fp32 tensor on a ROCm device")
        rows = features.permute(0, 2, 3, 1)
        if not rows.is_contiguous():
            raise ValueError("features must be in channels_last memory format (NHWC storage)")
        self = cls.__new__(cls)
        self.shape, self.device = tuple(int(d) for d in features.shape), features.device
        self.row_bytes = self.shape[1] * self.shape[2] * self.shape[3] * 4
        self.rows = rows
        return self

    def __len__(self) -> int:
        return self.shape[0]

    @property
    def image_feature_size(self):
        return self.shape[1:]

    def data_ptr(self) -> int:
        return self.rows.data_ptr()

    def batch(self, indices) -> ResidentRows:
        return ResidentRows(self, indices)


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
        if isinstance(store, DeviceFeatureStore):
            method = "resident"  # (no feature bytes move: a batch's "image" is a ResidentRows)
        if method not in ("dma", "kernel", "resident"):
            raise ValueError("method must be 'dma', 'kernel' or 'resident'")
        self.method = method
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if method == "resident" and store.device != device:
            raise _hip.HipLibraryError("the resident feature store lives on %s, the loader feeds %s" % (store.device, device))
        self.batches, self.store, self.device = batches, store, device
        self.keep_on_host = set(keep_on_host)
        # (high priority: the ingest is a trickle of long-latency PCIe reads -- or copy-engine transfers -- that must
        # not queue behind the step's thousands of workgroups)
        # (one loader stream per device and process: see trainers.joint_training.shared_stream)
        from probnmn.trainers.joint_training import shared_stream

        self.stream = shared_stream(device, "feature ingest", priority=-1)
        self._buffers = [None, None]  # two image buffers: the one in use and the one being filled

    def _stage(self, host_batch, slot: int):
        idx = host_batch["image_index"]
        n = int(idx.numel())
        C, H, W = self.store.image_feature_size
        buf = self._buffers[slot]
        if self.method != "resident" and (buf is None or buf.size(0) < n):
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
            if self.method == "resident":
                out["image"] = self.store.batch(idx)
            elif self.method == "kernel":
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
