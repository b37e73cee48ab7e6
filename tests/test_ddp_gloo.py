"""N>1 path on CPU: two gloo ranks, each a shard of the minibatch, ONE flat all-reduce per step
(kvae/train/train.py).  Checks (a) replicas stay bit-identical, (b) the 2-rank step equals the single-
process step on the concatenated global batch - with mask == 1 (mean of shard gradients) AND with a different
number of observed frames per rank (the ELBO is normalised by the local count, kalman_filter.py:392 / losses.py:82,
so the ranks' gradients are weighted by their counts inside the same all-reduce).  Kernels run in the test-only
host simulation."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
B_LOCAL, T = 3, 8


def _setup():
    for p in (str(ROOT), str(ROOT / "kalman-vae_amd"), str(ROOT / "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from kvae import _native
    from hostsim.build import build
    _native._set_test_backend(_native.LgssmLib(build()))


def _model(kind):
    from kvae.model.model import KVAE
    from kvae.utils.config import KVAEConfig
    torch.manual_seed(0)
    m = KVAE(KVAEConfig(dynamics_model=kind, num_modes=3)).train()
    with torch.no_grad():
        m.kalman_filter.dyn_params.A.add_(0.05 * torch.randn_like(m.kalman_filter.dyn_params.A))
        if hasattr(m.kalman_filter.dyn_params, "head_w"):
            m.kalman_filter.dyn_params.head_w.bias.zero_()
    m.beta = 1.0
    return m


def _data(world):
    from kvae.train.synthetic import bouncing_ball
    g = torch.Generator().manual_seed(3)
    B = B_LOCAL * world
    x = bouncing_ball(B, T, 11).float()
    return x, torch.randn(B * T, 2, generator=g), torch.randn(B, T, 4, generator=g), \
        -torch.empty(B, T, 3).exponential_(generator=g).log()


def _mask(world, unequal):
    """None (all observed) or a [B,T] mask whose ranks observe different numbers of frames (rank 0: 5 of 8 per
    sequence, rank 1: 7 of 8, plus one fully hidden sequence on rank 0)."""
    if not unequal:
        return None
    m = torch.ones(B_LOCAL * world, T)
    m[:B_LOCAL, 2:5] = 0.0
    m[B_LOCAL:, 6] = 0.0
    m[1, :] = 0.0
    return m


def _step(model, world, rank, x, eps_a, eps_z, gum, mask=None, phase=None):
    from kvae import noise
    from kvae.train.train import Trainer
    tr = Trainer(model, use_graph=False, world_size=world)
    if phase is not None:
        tr.set_training_phase(phase)
    sl = slice(rank * B_LOCAL, (rank + 1) * B_LOCAL) if world > 1 else slice(None)
    ea = eps_a.view(-1, T, 2)[sl].reshape(-1, 2)
    with noise.inject(eps_a=ea, eps_z=eps_z[sl], gumbel=gum[sl]):
        out = tr.step(x[sl], None if mask is None else mask[sl])
    return tr, out


def _worker(rank, world, port, kind, q, unequal=False, phase=None):
    torch.set_num_threads(1)
    _setup()
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr, out = _step(_model(kind), world, rank, *_data(world), mask=_mask(world, unequal), phase=phase)
    flat = torch.cat([p.detach().flatten() for p in tr.model.parameters()])
    q.put((rank, flat.numpy().copy(), tr.flat_grad.numpy().copy(), float(out["loss"])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,unequal,phase", [("lstm", False, None), ("switching", False, None), ("switching", True, None),
                                                ("lstm", True, None), ("lstm", False, "warmup"), ("switching", True, "vae")])
def test_two_rank_step_matches_global_batch(kind, unequal, phase):
    """phase: the step under one of the reference's training phases (train.py:142-207) - the frozen parameters' slots of the flat
    buffer travel through the all-reduce as zeros and must come out untouched on every rank."""
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, q, unequal, phase)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, p0, g0, l0), (_, p1, g1, l1) = [(r, torch.from_numpy(p), torch.from_numpy(g), l) for r, p, g, l in res]
    assert torch.equal(p0, p1), "replicas diverged"
    assert torch.equal(g0, g1), "all-reduced gradients differ between ranks"
    # single process, global batch
    _setup()
    tr, out = _step(_model(kind), 1, 0, *_data(world), mask=_mask(world, unequal), phase=phase)
    if phase is not None:   # what the phase froze did not move on any rank (bit-identical to the initial values)
        init = torch.cat([p.detach().flatten() for p in _model(kind).parameters()])
        frozen = torch.cat([torch.full((p.numel(),), not p.requires_grad) for p in tr.model.parameters()])
        assert bool(frozen.any()) and torch.equal(p0[frozen], init[frozen]) and not torch.equal(p0[~frozen], init[~frozen])
    gref = tr.flat_grad
    assert float((g0 - gref).abs().max() / gref.abs().max()) < 2e-4
    pref = torch.cat([p.detach().flatten() for p in tr.model.parameters()])
    assert float((p0 - pref).abs().max()) < 2e-3   # one Adam step of lr 7e-3: sign-level agreement
    if unequal:   # the global loss is the count-weighted mean of the rank losses
        m = _mask(world, True)
        c0, c1 = float(m[:B_LOCAL].sum()), float(m[B_LOCAL:].sum())
        assert c0 != c1
        assert abs((c0 * l0 + c1 * l1) / (c0 + c1) - float(out["loss"])) < 1e-4 * abs(float(out["loss"]))
    else:
        assert abs(0.5 * (l0 + l1) - float(out["loss"])) < 1e-4 * abs(float(out["loss"]))
