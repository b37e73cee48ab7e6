"""Checkpoints interchangeable with the reference's `Checkpointer` (kvae/train/utils.py:165-210 there): the payload is
{epoch, model_state, optimizer_state, train_loss, val_loss}, files are `kvae-best.pt` and `kvae-ckpt-epoch=NNN.pt`,
and `model_state` uses the same state_dict keys, so either side loads the other's files."""
from pathlib import Path

import torch


class Checkpointer:
    def __init__(self, checkpoint_dir, ckpt_every: int = 0):
        self.checkpoint_dir = Path(checkpoint_dir)
        self.ckpt_every = ckpt_every
        self.best_val = float("inf")
        self.checkpoint_dir.mkdir(parents=True, exist_ok=True)

    @staticmethod
    def payload(model, optimizer, epoch, train_loss, val_loss):
        """What the reference's Checkpointer writes (utils.py:190-198): plain-float hyperparameters and one state entry per
        parameter that has been updated at least once.  The Trainer keeps lr (and with an LRScheduler `initial_lr`) in device
        scalars and its Adam state in views of flat buffers with a zero-step entry for every parameter: tensor-valued
        hyperparameters become floats, state tensors are copied out of the flat buffers, and entries that never saw a
        gradient (frozen by the training phase in every step so far) are left out, as torch's Adam would have none."""
        opt_state = optimizer.state_dict()
        for g in opt_state["param_groups"]:
            for k, v in list(g.items()):
                if isinstance(v, torch.Tensor) and v.numel() == 1:
                    g[k] = float(v)
        state = {}
        for idx, st in opt_state["state"].items():
            if "step" in st and float(st["step"]) == 0.0:
                continue
            state[idx] = {k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in st.items()}
        opt_state["state"] = state
        return {"epoch": epoch, "model_state": model.state_dict(), "optimizer_state": opt_state,
                "train_loss": train_loss, "val_loss": val_loss}

    def save_checkpoint(self, path, model, optimizer, epoch, train_loss, val_loss):
        torch.save(self.payload(model, optimizer, epoch, train_loss, val_loss), path)

    def save_checkpoints(self, train_loss, val_loss, model, optimizer, epoch):
        if val_loss < self.best_val:
            self.best_val = val_loss
            self.save_checkpoint(self.checkpoint_dir / "kvae-best.pt", model, optimizer, epoch, train_loss, val_loss)
        if self.ckpt_every > 0 and epoch % self.ckpt_every == 0:
            self.save_checkpoint(self.checkpoint_dir / f"kvae-ckpt-epoch={epoch:03d}.pt", model, optimizer, epoch,
                                 train_loss, val_loss)


def load_checkpoint(path, model, optimizer=None, map_location="cpu"):
    """Load a checkpoint written by this Checkpointer or by the reference's (tensors only: weights_only=True).
    Returns the payload (without the state dicts) — note the reference itself has no resume code (train.py)."""
    payload = torch.load(path, map_location=map_location, weights_only=True)
    model.load_state_dict(payload["model_state"], strict=True)
    if optimizer is not None and "optimizer_state" in payload:
        lr_t = [g["lr"] for g in optimizer.param_groups]
        # how THIS optimizer runs is not the file's business: a reference-written file says fused None / capturable False,
        # which would turn a later capture of torch's Adam into a non-capturable one
        keep = [{k: g[k] for k in ("fused", "capturable", "foreach", "differentiable") if k in g} for g in optimizer.param_groups]
        optimizer.load_state_dict(payload["optimizer_state"])
        for g, t, kp in zip(optimizer.param_groups, lr_t, keep):
            g.update(kp)
            if isinstance(t, torch.Tensor):   # keep the device scalar a captured step reads; refill it
                t.fill_(float(g["lr"]))
                g["lr"] = t
        relink = getattr(optimizer, "_kvae_relink", None)   # a Trainer's optimizer: its state lives in flat buffers
        if relink is not None:
            relink()
    return {k: v for k, v in payload.items() if k not in ("model_state", "optimizer_state")}
