"""§8f row 4: `.npz` dataset contract (the reference's tests/test_pymunk_dataset.py recipe) and checkpoint payloads."""
import numpy as np
import pytest
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


def test_shards_are_equal_when_the_dataset_does_not_divide(tmp_path):
    """1023 sequences, 2 ranks, batch 256: every rank must see the same number of batches (one all-reduce per step
    on every rank, or the job deadlocks), and len() must agree with iteration."""
    p = tmp_path / "odd.npz"
    np.savez(p, images=np.zeros((1023, 2, 8, 8), np.uint8))
    ds = PymunkNPZDataset.from_npz(str(p), seq_len=2, state_key=None)
    for drop_last, want in ((True, [256, 256]), (False, [256, 255])):
        per_rank = []
        for r in range(2):
            loader = DeviceBatches(ds, 256, "cpu", shuffle=True, seed=0, rank=r, world_size=2, drop_last=drop_last)
            sizes = [b["images"].shape[0] for b in loader]
            assert len(sizes) == len(loader)
            per_rank.append(sizes)
        assert per_rank[0] == per_rank[1] == want[: len(per_rank[0])]
    # a shard smaller than one batch yields nothing (never an empty batch)
    small = DeviceBatches(ds, 2048, "cpu", shuffle=False, rank=0, world_size=2, drop_last=True)
    assert len(small) == 0 and list(small) == []


def test_block_mask_builders():
    from kvae.train.imputation import config_block_mask, make_training_mask, mask_impute_planning, mask_impute_random
    from kvae.utils.config import KVAEConfig
    m = mask_impute_planning(3, 10, t_init_mask=4, t_steps_mask=12)
    assert m.shape == (3, 10) and m[:, :4].eq(1).all() and m[:, 4:].eq(0).all()          # clipped at T
    m = mask_impute_planning(2, 20)
    assert m[:, :4].eq(1).all() and m[:, 4:16].eq(0).all() and m[:, 16:].eq(1).all()
    assert torch.equal(config_block_mask(KVAEConfig(t_init_mask=2, t_steps_mask=3), 2, 8),
                       mask_impute_planning(2, 8, 2, 3))
    torch.manual_seed(0)
    r = mask_impute_random(4, 50, t_init_mask=5, drop_prob=0.5)
    assert r[:, :5].eq(1).all() and 0.2 < float(r[:, 5:].mean()) < 0.8
    assert make_training_mask(2, 6).eq(1).all()
    assert torch.equal(make_training_mask(2, 20, strategy="block"), mask_impute_planning(2, 20))


@pytest.mark.parametrize("kind", ["lstm", "switching"])
def test_reference_written_checkpoint_loads(kind, tmp_path):
    """tests/golden/ref_checkpoint_*.pt were written by the REFERENCE's Checkpointer (kvae/train/utils.py:165-210;
    generated by tests/golden/make_goldens_r2.py).  They must load with weights_only=True, strictly, into this
    repo's KVAE and optimizer; the payload layout must be what our own Checkpointer writes."""
    from golden_util import GOLDEN, load
    from kvae.model.model import KVAE
    from kvae.utils.config import KVAEConfig
    g = load(f"ref_checkpoint_{kind}")
    cfg = KVAEConfig(dynamics_model=kind, num_modes=3, encoder_channels=g["encoder_channels"].tolist(),
                     decoder_channels=g["decoder_channels"].tolist(), dynamics_hidden_dim=int(g["dynamics_hidden_dim"]))
    with pytest.warns(UserWarning):          # narrow VAE / small RNN: off the specialised kernels, and the model says so
        model = KVAE(cfg)
    opt = torch.optim.Adam(model.parameters(), lr=1.0)
    path = GOLDEN / f"ref_checkpoint_{kind}.pt"
    meta = load_checkpoint(path, model, opt)
    assert meta["epoch"] == 1 and set(meta) == {"epoch", "train_loss", "val_loss"}
    assert abs(meta["train_loss"] - float(g["train_loss"])) < 1e-6 and abs(meta["val_loss"] - float(g["val_loss"])) < 1e-6
    assert abs(opt.param_groups[0]["lr"] - float(g["lr"])) < 1e-12
    assert len(opt.state) == len(list(model.parameters()))            # Adam moments of the reference's one step
    raw = torch.load(path, map_location="cpu", weights_only=True)
    assert list(raw["model_state"]) == list(model.state_dict())        # same keys, same order
    ours = Checkpointer.payload(model, opt, 1, meta["train_loss"], meta["val_loss"])
    assert set(ours) == set(raw) and set(ours["optimizer_state"]) == set(raw["optimizer_state"])
    # ... down to the types of every hyperparameter, also with an LRScheduler attached (torch stores `initial_lr` as a clone
    # of a tensor lr) and a tensor learning rate (what the Trainer's optimizer holds)
    opt2 = torch.optim.Adam(model.parameters(), lr=torch.tensor(1e-3), weight_decay=0.0)
    torch.optim.lr_scheduler.ExponentialLR(opt2, gamma=0.9)
    ours2 = Checkpointer.payload(model, opt2, 1, 0.0, 0.0)["optimizer_state"]["param_groups"][0]
    theirs = raw["optimizer_state"]["param_groups"][0]
    assert type(ours2["initial_lr"]) is float and type(ours2["lr"]) is float
    for k, v in theirs.items():
        assert type(ours2[k]) is type(v), (k, type(ours2[k]), type(v))
