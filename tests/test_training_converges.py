"""The whole training path (npz loader -> DeviceBatches -> Trainer -> HIP chain [host-simulated here]) reduces the loss."""
import numpy as np
import torch

from hostsim.build import build as build_hostsim


def test_loss_decreases_on_bouncing_ball(tmp_path):
    from kvae import _native
    from kvae.dataloader.pymunk_dataset import DeviceBatches, PymunkNPZDataset
    from kvae.model.model import KVAE
    from kvae.train.synthetic import bouncing_ball
    from kvae.train.train import Trainer, train_one_epoch
    from kvae.utils.config import KVAEConfig
    _native._set_test_backend(_native.LgssmLib(build_hostsim()))
    try:
        torch.manual_seed(0)
        frames = bouncing_ball(32, 10, seed=0).numpy()[:, :, 0] * 255
        np.savez_compressed(tmp_path / "bb.npz", images=frames.astype(np.uint8))
        ds = PymunkNPZDataset.from_npz(tmp_path / "bb.npz", seq_len=10, state_key=None)
        loader = DeviceBatches(ds, 16, "cpu", shuffle=True, seed=0)
        model = KVAE(KVAEConfig(dynamics_model="lstm"))
        tr = Trainer(model, lr=7e-3, use_graph=False)
        tr.set_beta(1.0)
        losses = [train_one_epoch(tr, loader, "cpu")["loss"] for _ in range(4)]
        assert all(np.isfinite(losses)) and losses[-1] < 0.7 * losses[0], losses
    finally:
        _native._set_test_backend(None)
