"""Frame-mask builders of the reference's imputation tooling (kvae/train/imputation.py:4-36 there): the input
pattern of BASELINE configs[3] ("observe, hide a block, observe again") and of masked training.  Same function
names, arguments and results; everything else in that module (imputation plots, MSE reports) is evaluation tooling
outside the hot path and is not shipped.  `config_block_mask` wires KVAEConfig.t_init_mask / t_steps_mask in.
"""
import torch


def mask_impute_planning(batch_size, T, t_init_mask=4, t_steps_mask=12, device=None):
    """[B,T] mask, 1 = observed: frames [t_init_mask, t_init_mask + t_steps_mask) hidden, clipped at T."""
    t = torch.arange(T, device=device)
    hidden = (t >= t_init_mask) & (t < min(t_init_mask + t_steps_mask, T))
    return (~hidden).to(torch.float32).expand(batch_size, T).contiguous()


def mask_impute_random(batch_size, T, t_init_mask=4, drop_prob=0.5, device=None):
    """First t_init_mask frames observed; every later frame dropped independently with probability drop_prob."""
    mask = torch.ones(batch_size, T, device=device)
    if T > t_init_mask:
        keep = torch.full((batch_size, T - t_init_mask), 1.0 - drop_prob, device=device)
        mask[:, t_init_mask:] = torch.bernoulli(keep)
    return mask


def make_training_mask(batch_size, T, t_init_mask=4, drop_prob=0.0, device=None, strategy="random", t_steps_mask=12):
    if strategy.lower() == "block":
        return mask_impute_planning(batch_size, T, t_init_mask=t_init_mask, t_steps_mask=t_steps_mask, device=device)
    if drop_prob <= 0:
        return torch.ones(batch_size, T, device=device)
    return mask_impute_random(batch_size, T, t_init_mask=t_init_mask, drop_prob=drop_prob, device=device)


def config_block_mask(config, batch_size, T, device=None):
    """The block mask the reference's epoch loop evaluates imputation with (train.py:318-322 there):
    KVAEConfig.t_init_mask observed frames, then KVAEConfig.t_steps_mask hidden ones."""
    return mask_impute_planning(batch_size, T, t_init_mask=config.t_init_mask, t_steps_mask=config.t_steps_mask,
                                device=device)
