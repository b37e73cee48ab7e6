#!/usr/bin/env python3
"""Where the replayed two-stream step really is when the streams meet: one-wave clock-stamp kernels (tools/stamp.hip) captured
into the step's hipGraph at a few points of both streams (a kernel trace serialises parts of the graph), read back
after untraced replays.  usage: hipcc ... tools/stamp.hip -o tools/_bin/libstamp.so; python3 tools/join_wait_probe.py"""
import argparse, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kalman-vae_amd")]
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--replays", type=int, default=50)
ap.add_argument("--dynamics", default="lstm")
ap.add_argument("--modes", type=int, default=3)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--seq-len", type=int, default=50)
args = ap.parse_args()
import bench
from kvae.model import model as M
from kvae.train.synthetic import bouncing_ball
from kvae.train.train import Trainer

dev = torch.device("cuda:0")
bargs = argparse.Namespace(dynamics=args.dynamics, modes=args.modes, z_dim=4, seq_len=args.seq_len)
cfg, model = bench.build_model(bargs, dev)
x = bouncing_ball(args.batch, args.seq_len, 1234).float().to(dev)
events = {}


import ctypes
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_bin", "libstamp.so"))
lib.kvae_tool_stamp.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
stamps = torch.zeros(32, device=dev, dtype=torch.int64)


def mark(name, stream=None):
    slot = events.setdefault(name, len(events))
    s_ = stream if stream is not None else torch.cuda.current_stream()
    assert lib.kvae_tool_stamp(stamps.data_ptr(), slot, s_.cuda_stream) == 0


def after(obj, attr, name):
    f = getattr(obj, attr)

    def g(*a, **k):
        r = f(*a, **k)
        mark(name)
        return r
    setattr(obj, attr, g)


trainer = Trainer(model, use_graph=True, world_size=1, reference_logging=True)
fb = trainer._forward_backward


def fb_marked(x_, mask=None):
    mark("0 step start (main)")
    fb(x_, mask)
    mark("9 gradients gathered (main)")


trainer._forward_backward = fb_marked
after(model, "encode_sequence", "1 encoder done = fork (main)")
after(model.kalman_filter, "smooth", "2 filter + smoother enqueued behind (side)")
after(model.kalman_filter, "elbo", "3 ELBO forward (side)")
after(model, "decode_sequence", "4 decoder forward (main)")
dec = model.forward


def fwd(*a, **k):
    out = dec(*a, **k)
    out["x_logits"].register_hook(lambda g: (mark("5 decoder backward starts (main)"), None)[1])
    return out


model.forward = fwd
join_bwd = M._SideGradJoin.backward


def join_marked(ctx, g):
    side = ctx.holder["side"]
    mark("6 main stream reaches the join")
    mark("7 side chain done (side)", side)
    r = join_bwd(ctx, g)
    mark("8 join passed (main)")
    return r


M._SideGradJoin.backward = staticmethod(join_marked)
for _ in range(5):
    trainer.step(x)
torch.cuda.synchronize()
rows = {k: [] for k in events}
for _ in range(args.replays):
    trainer.step(x)
    torch.cuda.synchronize()
    t = stamps.cpu().tolist()
    for k, slot in events.items():
        rows[k].append((t[slot] - t[events["0 step start (main)"]]) / 100.0)   # 100 MHz
for k in sorted(rows):
    v = rows[k]
    print(f"{statistics.median(v):9.1f} us  (min {min(v):8.1f}, max {max(v):8.1f})  {k}")
