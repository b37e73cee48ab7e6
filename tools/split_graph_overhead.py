#!/usr/bin/env python3
"""What the multi-rank step structure costs on one GPU: a 1-rank RCCL communicator, Trainer(world_size=2) (forward+backward graph,
eager all-reduce of the flat gradient buffer, clip+Adam graph) against the single-graph step at BASELINE configs[1].
usage: python tools/split_graph_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kalman-vae_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
from kvae.model.model import KVAE
from kvae.train.synthetic import bouncing_ball
from kvae.train.train import Trainer
from kvae.utils.config import KVAEConfig

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
x = bouncing_ball(256, 50, 1234).float().cuda()
for world in (1, 2, 1, 2):
    torch.manual_seed(0)
    model = KVAE(KVAEConfig(dynamics_model="lstm", num_modes=3)).cuda().train()
    model.beta = 1.0
    tr = Trainer(model, use_graph=True, world_size=world, reference_logging=True)
    for _ in range(20):
        tr.step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        tr.step(x)
    torch.cuda.synchronize()
    print(f"world_size={world}: {(time.perf_counter() - t0) / 300 * 1e3:.3f} ms/step ({'split graph + RCCL all-reduce' if world > 1 else 'one graph'})")
dist.destroy_process_group()
