#!/usr/bin/env python3
"""Train the KVAE on synthetic bouncing-ball video with the MI355X-native path (one process per GPU).

  python examples/train_bouncing_ball.py --epochs 3                       # 1 GPU
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_bouncing_ball.py

Mirrors the reference's kvae/train/train.py main loop (:246-336): the three training phases ("vae" with kf_weight 0, then
"warmup" with the alpha-network frozen, then "all"; each switch re-captures the step's hipGraph once), beta schedule, Adam +
ExponentialLR and the tau decay (all of which reach the captured graph through device scalars), grad clip 10,
reference-compatible checkpoints; data come from an .npz (uint8 (N,T,H,W), the reference's format) written on the fly.
"""
import argparse
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "kalman-vae_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from kvae.dataloader.pymunk_dataset import DeviceBatches, PymunkNPZDataset  # noqa: E402
from kvae.model.model import KVAE  # noqa: E402
from kvae.train.checkpoint import Checkpointer  # noqa: E402
from kvae.train.synthetic import bouncing_ball  # noqa: E402
from kvae.train.train import Trainer, end_of_epoch_schedules, init_distributed, phase_for_epoch, train_one_epoch  # noqa: E402
from kvae.utils.config import KVAEConfig  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--sequences", type=int, default=2048)
    ap.add_argument("--seq-len", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="per GPU")
    ap.add_argument("--dynamics", default="lstm")
    ap.add_argument("--out", default=None)
    ap.add_argument("--config", default=None, help="reference-style YAML (its `kvae:` section becomes the KVAEConfig)")
    ap.add_argument("--decay-steps", type=int, default=1, help="epochs between LR decays (reference default: 20)")
    ap.add_argument("--pretrain-vae-epochs", type=int, default=1, help="epochs of phase 'vae' (reference default: 5)")
    ap.add_argument("--warmup-epochs", type=int, default=1, help="epochs of phase 'warmup' (reference default: 10)")
    args = ap.parse_args()
    rank, world, dev = init_distributed()
    out = Path(args.out or tempfile.mkdtemp(prefix="kvae_run_"))
    npz = out / "bouncing_ball.npz"
    if rank == 0:
        out.mkdir(parents=True, exist_ok=True)
        frames = bouncing_ball(args.sequences, args.seq_len, seed=0).numpy()[:, :, 0] * 255
        np.savez_compressed(npz, images=frames.astype(np.uint8))
    if world > 1:
        torch.distributed.barrier()
    ds = PymunkNPZDataset.from_npz(npz, seq_len=args.seq_len, state_key=None)
    loader = DeviceBatches(ds, args.batch, dev, shuffle=True, seed=1, rank=rank, world_size=world)
    torch.manual_seed(0)
    cfg = KVAEConfig.from_yaml(args.config) if args.config else KVAEConfig(dynamics_model=args.dynamics)
    model = KVAE(cfg).to(dev)
    trainer = Trainer(model, lr=7e-3, world_size=world, use_graph=dev.type == "cuda")
    sched = torch.optim.lr_scheduler.ExponentialLR(trainer.opt, gamma=0.85)
    ck = Checkpointer(out / "checkpoints", ckpt_every=0) if rank == 0 else None
    if rank == 0:
        print(cfg.describe(), f"| {world} rank(s), {len(loader)} steps/epoch")
    tau_start = max(1, args.pretrain_vae_epochs + args.warmup_epochs + 1)   # train.py:244 there
    phase = None
    for epoch in range(1, args.epochs + 1):
        want, kf_w, vae_w = phase_for_epoch(epoch, args.pretrain_vae_epochs, args.warmup_epochs)
        if want != phase:
            phase = want
            trainer.set_training_phase(phase, kf_weight=kf_w, vae_weight=vae_w)
            if rank == 0:
                print(f"=== training phase '{phase}' from epoch {epoch} (kf_weight {kf_w}) ===")
        trainer.set_beta(model.scheduler.get_beta(epoch) if cfg.scheduled_beta else 1.0)
        stats = train_one_epoch(trainer, loader, dev)
        lr, tau = end_of_epoch_schedules(trainer, sched, epoch, decay_steps=args.decay_steps, tau_decay_start_epoch=tau_start)
        if rank == 0:
            print(f"epoch {epoch} [{phase}]: loss {stats['loss']:.4f} elbo_kf {stats['elbo_kf']:.4f} elbo_vae {stats['elbo_vae_total']:.4f}"
                  f" | next lr {lr:.3e}" + (f" tau {tau:.3f}" if tau is not None else ""))
            ck.save_checkpoints(stats["loss"], stats["loss"], model, trainer.opt, epoch)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
