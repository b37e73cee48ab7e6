"""`.npz` video datasets: same constructor / `from_npz` / item contract as the reference's
PymunkNPZDataset (kvae/dataloader/pymunk_dataset.py:51-220 there) — items are {'images': float32 [T,C,H,W]
(per-frame min-max normalised when normalize=True), 'state': float32 [T,D]} — plus a device-side batch path
(`DeviceBatches`) built for one process per GPU: the uint8 frames stay in pinned host memory, a batch crosses
PCIe as uint8 (4x fewer bytes than the reference's float32 DataLoader batches) on a dedicated copy stream one batch
ahead, and is widened / normalised on the GPU.
"""
from pathlib import Path
from typing import Any, Dict, Optional

import numpy as np
import torch
from torch.utils.data import Dataset


def _as_sequences(imgs: np.ndarray, seq_len: int, stride: int) -> np.ndarray:
    """Standardise the accepted layouts to (N, T, C, H, W):
    (N,T,C,H,W) | (N,T,H,W) -> C=1 | (F,C,H,W), (F,H,W) flat frames -> sliding windows of seq_len."""
    def windows(frames):
        F = frames.shape[0]
        if F < seq_len:
            raise ValueError(f"Not enough frames ({F}) for seq_len={seq_len}")
        starts = range(0, F - seq_len + 1, stride)
        return np.stack([frames[s:s + seq_len] for s in starts], axis=0)

    if imgs.ndim == 5:
        return imgs
    if imgs.ndim == 4:
        _, d1, d2, d3 = imgs.shape
        if d2 >= 8 and d3 >= 8:                       # (N,T,H,W): the last two axes look like an image
            return imgs[:, :, None]
        return windows(imgs[:, None])                 # (F,H,W)-like frames
    if imgs.ndim == 3:
        return windows(imgs[:, None])
    if imgs.ndim > 5:                                 # collapse extra middle axes into channels
        n, t, h, w = imgs.shape[0], imgs.shape[1], imgs.shape[-2], imgs.shape[-1]
        return imgs.reshape(n, t, -1, h, w)
    raise ValueError(f"Unsupported image array shape: {imgs.shape}")


class PymunkNPZDataset(Dataset):
    def __init__(self, npz_path, image_key: str = "images", state_key: Optional[str] = "state", seq_len: int = 10,
                 stride: int = 1, normalize: bool = True, load_in_memory: bool = True):
        self.path = Path(npz_path)
        if not self.path.exists():
            raise FileNotFoundError(self.path)
        self.image_key, self.state_key = image_key, state_key
        self.seq_len, self.stride, self.normalize = int(seq_len), int(stride), bool(normalize)
        with np.load(self.path, allow_pickle=False) as z:       # never unpickles
            if image_key not in z.files:
                raise KeyError(f"Image key '{image_key}' not in NPZ. Available: {list(z.files)}")
            self.raw = {k: np.asarray(z[k]) for k in z.files if k in (image_key, state_key)}
        self.seq_data = np.ascontiguousarray(_as_sequences(self.raw[image_key], self.seq_len, self.stride))
        self.N, self.T, self.C, self.H, self.W = self.seq_data.shape
        self.state_data = None
        if state_key is not None and state_key in self.raw:
            st = self.raw[state_key]
            if st.ndim != 3 or st.shape[:2] != (self.N, self.T):
                raise ValueError(f"State array shape {st.shape} does not match images {(self.N, self.T)}")
            self.state_data = st.astype(np.float32)
        self.index = list(range(self.N))

    @classmethod
    def from_npz(cls, npz_path, **kwargs) -> "PymunkNPZDataset":
        return cls(npz_path, **kwargs)

    def __len__(self) -> int:
        return len(self.index)

    def __getitem__(self, idx: int) -> Dict[str, Any]:
        if isinstance(idx, slice):
            raise NotImplementedError("Slicing not implemented")
        seq = self.seq_data[self.index[idx]].astype(np.float32)
        if self.normalize:
            seq = seq - seq.min(axis=(2, 3), keepdims=True)
            denom = seq.max(axis=(2, 3), keepdims=True)
            denom[denom == 0] = 1.0
            seq = seq / denom
        out = {"images": torch.from_numpy(seq)}
        if self.state_data is not None:
            out["state"] = torch.from_numpy(self.state_data[self.index[idx]])
        return out


class DeviceBatches:
    """Iterate {'images': float32 [B,T,C,H,W] on `device`} over a PymunkNPZDataset (optionally one rank's shard).

    resident=True (the default when the uint8 sequences fit in a quarter of the device memory - the reference's data sets
    are a few GB against 288 GB of HBM): the whole data set crosses PCIe ONCE, batches are gathered and normalised on
    the GPU, and an epoch costs no host work per step.  Otherwise: uint8 over PCIe from pinned memory on a copy stream,
    one batch ahead, normalisation on the GPU (host-bound at ~11 ms per 256 x 50 batch, i.e. 3x the training step)."""

    def __init__(self, dataset: PymunkNPZDataset, batch_size: int, device, shuffle=True, seed=0, rank=0, world_size=1,
                 drop_last=True, resident=None):
        self.ds, self.bs, self.device = dataset, batch_size, torch.device(device)
        self.shuffle, self.seed, self.rank, self.world = shuffle, seed, rank, world_size
        self.drop_last, self.epoch = drop_last, 0
        data = torch.from_numpy(dataset.seq_data)
        on_gpu = self.device.type == "cuda"
        if resident is None:
            resident = on_gpu and data.numel() * data.element_size() <= torch.cuda.get_device_properties(self.device).total_memory // 4
        self.resident = bool(resident) and on_gpu
        self.dev_data = data.to(self.device) if self.resident else None
        self.host = None if self.resident else (data.pin_memory() if on_gpu else data)
        self.copy_stream = torch.cuda.Stream(self.device) if (on_gpu and not self.resident) else None

    def _per_rank(self):
        """Sequences every rank iterates over: the same number on all ranks (the trailing remainder of the
        permutation is dropped), so that every rank issues the same number of all-reduces per epoch."""
        per = len(self.ds) // self.world
        return (per // self.bs) * self.bs if self.drop_last else per

    def __len__(self):
        per = self._per_rank()
        return per // self.bs if self.drop_last else -(-per // self.bs)

    def _upload(self, idx):
        if self.resident:
            return self._normalise(self.dev_data[idx.to(self.device)].float())
        if self.copy_stream is None:
            return self._normalise(self.host[idx].float())
        with torch.cuda.stream(self.copy_stream):
            staged = self.host[idx].pin_memory()                      # gather on the host, then one async H2D copy
            x = self._normalise(staged.to(self.device, non_blocking=True).float())
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return x, ev

    def _normalise(self, x):
        if not self.ds.normalize:
            return x
        lo = x.amin(dim=(-2, -1), keepdim=True)
        x = x - lo
        hi = x.amax(dim=(-2, -1), keepdim=True)
        return x / torch.where(hi == 0, torch.ones_like(hi), hi)

    def __iter__(self):
        n = len(self.ds)
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        order = torch.randperm(n, generator=g) if self.shuffle else torch.arange(n)
        per = self._per_rank()
        order = order[: (n // self.world) * self.world][self.rank::self.world][:per]   # truncate, THEN stride: equal shards
        chunks = list(order.split(self.bs)) if per > 0 else []
        self.epoch += 1
        nxt = self._upload(chunks[0]) if chunks else None
        for i in range(len(chunks)):
            cur = nxt
            nxt = self._upload(chunks[i + 1]) if i + 1 < len(chunks) else None
            if self.copy_stream is None:
                yield {"images": cur}
            else:
                x, ev = cur
                torch.cuda.current_stream(self.device).wait_event(ev)
                x.record_stream(torch.cuda.current_stream(self.device))
                yield {"images": x}
