"""§8f row 4: `.npz` dataset contract (the reference's tests/test_pymunk_dataset.py recipe) and checkpoint payloads."""
import numpy as np
import torch

from kvae.dataloader.pymunk_dataset import DeviceBatches, PymunkNPZDataset
from kvae.train.checkpoint import Checkpointer, load_checkpoint


def _make_npz(path, N=5, T=20, H=32, W=32, D=4):
    rng = np.random.default_rng(0)
    np.savez_compressed(path, images=rng.integers(0, 256, size=(N, T, H, W), dtype=np.uint8),
                        state=rng.standard_normal((N, T, D)).astype(np.float32))


def test_npz_dataset_contract(tmp_path):
    p = tmp_path / "data.npz"
    _make_npz(p)
    ds = PymunkNPZDataset.from_npz(str(p), seq_len=20)
    assert len(ds) == 5
    item = ds[0]
    assert item["images"].shape == (20, 1, 32, 32) and item["images"].dtype == torch.float32
    assert float(item["images"].min()) == 0.0 and float(item["images"].max()) == 1.0   # per-frame min-max
    assert item["state"].shape == (20, 4)


def test_flat_frames_become_windows(tmp_path):
    p = tmp_path / "flat.npz"
    np.savez(p, images=np.random.default_rng(1).integers(0, 256, size=(30, 6, 6), dtype=np.uint8))
    ds = PymunkNPZDataset(p, seq_len=10, stride=5, state_key=None)
    assert len(ds) == 5 and ds[0]["images"].shape == (10, 1, 6, 6)


def test_device_batches_match_items_and_shard(tmp_path):
    p = tmp_path / "data.npz"
    _make_npz(p, N=8)
    ds = PymunkNPZDataset.from_npz(str(p), seq_len=20)
    loader = DeviceBatches(ds, batch_size=2, device="cpu", shuffle=False)
    batches = list(loader)
    assert len(batches) == 4 == len(loader)
    assert torch.allclose(batches[1]["images"][0], ds[2]["images"])
    seen = []
    for r in range(2):
        for b in DeviceBatches(ds, 2, "cpu", shuffle=True, seed=3, rank=r, world_size=2):
            seen.append(b["images"])
    assert len(seen) == 4   # 2 ranks x 2 batches: every sequence exactly once


def test_checkpoint_roundtrip(tmp_path):
    from kvae.model.model import KVAE
    from kvae.utils.config import KVAEConfig
    torch.manual_seed(0)
    m = KVAE(KVAEConfig(dynamics_model="switching"))
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    ck = Checkpointer(tmp_path / "ckpt", ckpt_every=5)
    ck.save_checkpoints(1.5, 1.25, m, opt, epoch=5)
    assert (tmp_path / "ckpt" / "kvae-best.pt").exists() and (tmp_path / "ckpt" / "kvae-ckpt-epoch=005.pt").exists()
    torch.manual_seed(1)
    m2 = KVAE(KVAEConfig(dynamics_model="switching"))
    meta = load_checkpoint(tmp_path / "ckpt" / "kvae-best.pt", m2)
    assert meta["epoch"] == 5 and set(meta) == {"epoch", "train_loss", "val_loss"}
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
