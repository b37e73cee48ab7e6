"""GPU tier: the N>1 code path of kvae.train.train.Trainer (graph cut around the single all-reduce) with two
processes sharing the one GPU of the test box.  RCCL cannot put two ranks on one device, so the collective runs
over gloo here (KVAE_DIST_BACKEND); everything else — hipGraph capture of forward+backward, eager all-reduce of the
flat gradient buffer, captured clip+Adam — is exactly what the 8-GPU run executes."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _worker(rank, world, port, q):
    for p in (str(ROOT), str(ROOT / "kalman-vae_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", KVAE_DIST_BACKEND="gloo")
    from kvae.model.model import KVAE
    from kvae.train.synthetic import bouncing_ball
    from kvae.train.train import Trainer, init_distributed
    from kvae.utils.config import KVAEConfig
    rank, world, dev = init_distributed()
    torch.manual_seed(0)
    model = KVAE(KVAEConfig(dynamics_model="lstm", num_modes=3)).to(dev).train()
    model.beta = 1.0
    x = bouncing_ball(16, 12, 1234 + rank).float().to(dev)
    tr = Trainer(model, use_graph=True, world_size=world)
    for _ in range(4):
        out = tr.step(x)
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().flatten() for p in model.parameters()]).cpu().numpy()
    q.put((rank, flat, float(out["loss"]), tr.graph_opt is not None))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_graph_split_on_one_gpu():
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][3] and res[1][3], "graph was not split around the all-reduce"
    import numpy as np
    assert np.array_equal(res[0][1], res[1][1]), "replicas diverged after 4 steps"
    assert all(np.isfinite(r[2]) for r in res)
