#!/usr/bin/env python3
"""Data-path-inclusive training rate: the same C2 step as bench.py, but every batch comes through
kvae.dataloader.DeviceBatches - resident mode (data set uploaded once, batches gathered on the GPU) and streaming mode
(uint8 gather on the host -> pinned -> async H2D on a copy stream), both normalised on the GPU."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "kalman-vae_amd")]
import numpy as np, torch
from kvae.dataloader.pymunk_dataset import DeviceBatches, PymunkNPZDataset
from kvae.model.model import KVAE
from kvae.train.synthetic import bouncing_ball
from kvae.train.train import Trainer, train_one_epoch
from kvae.utils.config import KVAEConfig
import tempfile
dev = torch.device("cuda")
frames = bouncing_ball(8192, 50, seed=0).numpy()[:, :, 0] * 255        # 8192 sequences = 32 batches of 256
frames = frames.astype(np.uint8)                                       # the reference's .npz format: uint8 (N,T,H,W)
d = tempfile.mkdtemp(); np.savez(f"{d}/bb.npz", images=frames)
ds = PymunkNPZDataset.from_npz(f"{d}/bb.npz", seq_len=50, state_key=None)
torch.manual_seed(0)
model = KVAE(KVAEConfig(dynamics_model="lstm", num_modes=3)).to(dev)
model.beta = 1.0
tr = Trainer(model)
for resident in (True, False):
    loader = DeviceBatches(ds, 256, dev, shuffle=True, resident=resident)
    train_one_epoch(tr, loader, dev)                                    # warm-up epoch (capture etc.)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        train_one_epoch(tr, loader, dev)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    steps = 3 * len(loader)
    print(f"{'resident data set' if resident else 'streaming over PCIe'}: {steps} steps, {1e3 * dt / steps:.3f} ms/step, "
          f"{256 * steps / dt:.0f} sequences/s", flush=True)
