#!/usr/bin/env python3
"""bench.py — sequences/sec of one ELBO training step of the KVAE on synthetic 32x32xT bouncing-ball video.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c4|c4-lstm|c5|c5-lstm]

Workloads (per GPU; weak scaling):
  c2 (default)  BASELINE.json configs[1] / configs[2]: dynamics 'lstm', K=3 modes, z=4, a=2, B=256 sequences of T=50
  c4 / c4-lstm  BASELINE.json configs[3]: K=7, T=100, B=32 (the reference's batch), switching / lstm dynamics; also times the
                eval-mode imputation (`KVAE.impute` with the block mask) the config names
  c5 / c5-lstm  BASELINE.json configs[4] shard: z = u = 16, T = 200, B = 512 per GPU (4096 over 8 GPUs); c5 is the switching
                model (per-step Q: the 21 804 B/step SURVEY 8(d) quotes), c5-lstm the shared-Q variant
Any of --batch / --seq-len / --z-dim / --dynamics / --modes overrides the preset; `config.workload` always names what ran.
The default invocation (no workload flag) times c2 as the headline and then appends short runs of c4, c4-lstm and c5 as
`"also": {...}` (ms/step, sequences/s, roofline of their dominant LGSSM kernel; c4 with impute and its CPU baseline), so
that one driver-timed line witnesses every BASELINE config that fits one GPU (`--also none` skips them).

A step = zero_grad + forward (incl. sigmoid(x_logits)) + loss (incl. the active-unit statistics of the reference's
compute_loss, kept on the device) + backward + [one flat RCCL all-reduce] + clip_grad_norm_(10) + Adam (reference
kvae/train/train.py:44-58), fp32, inputs resident in HBM, random-init weights.  The CPU baseline leg does the same work.

Ranks: `--gpus N` with N > 1 and no WORLD_SIZE in the environment starts N rank processes itself
(`python -m torch.distributed.run`, before this process touches a GPU); under torch.distributed.run the flag must
equal WORLD_SIZE.  Fewer visible devices than ranks is an error (non-zero exit, no JSON line).
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "kalman-vae_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("DISABLE_ADDMM_CUDA_LT", "1")        # see kvae/train/train.py: hipBLASLt is not capture-safe
os.environ.setdefault("TORCH_BLAS_PREFER_HIPBLASLT", "0")

_T0 = time.perf_counter()
MFMA_F32_PEAK_TFS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA = vector peak
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy peak is ~6290 GB/s

PRESETS = {
    "c2": dict(batch=256, seq_len=50, z_dim=4, dynamics="lstm", modes=3),
    "c4": dict(batch=32, seq_len=100, z_dim=4, dynamics="switching", modes=7, impute=True),
    "c4-lstm": dict(batch=32, seq_len=100, z_dim=4, dynamics="lstm", modes=7, impute=True),
    "c5": dict(batch=512, seq_len=200, z_dim=16, dynamics="switching", modes=3),
    "c5-lstm": dict(batch=512, seq_len=200, z_dim=16, dynamics="lstm", modes=3),
}
ALSO_DEFAULT = ("c4", "c4-lstm", "c5")          # appended to the default run (python bench.py), each a few seconds
ALSO_STEPS = {"c4": 100, "c4-lstm": 100, "c5": 20}
BASELINE_NAMES = {
    (256, 50, 4, "lstm", 3): "BASELINE configs[1] (configs[2] per-GPU shard)",
    (32, 100, 4, "switching", 7): "BASELINE configs[3] (switching-LDS, K=7, T=100; the reference's own batch of 32)",
    (32, 100, 4, "lstm", 7): "BASELINE configs[3] with the LSTM alpha-net (K=7, T=100, batch of 32)",
    (512, 200, 16, "lstm", 3): "BASELINE configs[4] per-GPU shard with the LSTM alpha-net (shared Q)",
    (512, 200, 16, "switching", 3): "BASELINE configs[4] per-GPU shard (z=u=16, T=200, 512 of 4096 sequences), switching "
                                    "dynamics: the per-step-Q variant SURVEY 8(d) quotes 21 804 B/step for",
}


def log(msg):
    print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(PRESETS), default="c2")
    ap.add_argument("--batch", type=int, default=None, help="sequences per GPU")
    ap.add_argument("--seq-len", type=int, default=None)
    ap.add_argument("--dynamics", default=None)
    ap.add_argument("--modes", type=int, default=None)
    ap.add_argument("--z-dim", type=int, default=None)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-steady", action="store_true", help="skip the extra >= 2 s steady-state windows")
    ap.add_argument("--no-roofline", action="store_true",
                    help="skip the eager event-timed pass (profiling runs: keeps the trace to the replayed graph's steps)")
    ap.add_argument("--no-overlap", action="store_true", help="keep the LGSSM chain on the main stream")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="launcher/rendezvous rehearsal on the CPU (gloo): no model, no GPU, marks the line dry_run")
    ap.add_argument("--also", default=None,
                    help="comma-separated presets measured briefly after the headline and appended as `also` "
                         "('none' to skip; default: c4,c4-lstm,c5 on one GPU when no other workload flag is given)")
    ap.add_argument("--impute", action="store_true", help="also time eval-mode KVAE.impute (default for the c4 presets)")
    ap.add_argument("--graph-allreduce", action="store_true", help="N > 1: capture the RCCL all-reduce into the step graph")
    args = ap.parse_args(argv)
    plain = args.config == "c2" and all(getattr(args, k) is None for k in ("batch", "seq_len", "dynamics", "modes", "z_dim")) \
        and not (args.no_graph or args.no_roofline or args.no_steady or args.no_overlap) and args.gpus == 1
    if args.also is None:
        args.also = list(ALSO_DEFAULT) if plain else []
    else:
        args.also = [] if args.also.lower() in ("", "none") else [a for a in args.also.split(",") if a]
    for a in args.also:
        if a not in PRESETS:
            ap.error(f"--also: unknown preset {a}")
    args.cpu_in_also = False
    apply_preset(args)
    return args


def apply_preset(args):
    for k, v in PRESETS[args.config].items():
        if k == "impute":
            args.impute = bool(args.impute or v)
        elif getattr(args, k) is None:
            setattr(args, k, v)


def workload_name(args, cfg_dims):
    B, T = args.batch, args.seq_len
    n, m, p = cfg_dims
    base = BASELINE_NAMES.get((B, T, n, args.dynamics, args.modes), "custom size (not a BASELINE config)")
    return (f"{base}: bouncing-ball 32x32, T={T}, batch={B}/GPU, dynamics={args.dynamics} K={args.modes}, z={n}, u={m}, "
            f"a={p}, full train step (fwd incl. sigmoid(x_logits) + loss incl. active-unit stats + bwd + clip + Adam); "
            f"both the GPU leg and the cpu_baseline leg do this work")


# ------------------------------------------------------------------------------------------------------------------
# rank launcher
# ------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus():
    """GPUs this process could open, counted WITHOUT touching the HIP runtime (the launcher parent must stay free of it: on
    ROCm torch.cuda.device_count() calls hipGetDeviceCount, which brings HSA up in a process that only spawns ranks): KFD's
    topology in sysfs, narrowed by the *_VISIBLE_DEVICES variables when they are set."""
    n = 0
    for props in Path("/sys/class/kfd/kfd/topology/nodes").glob("*/properties"):
        try:
            kv = dict(line.split(None, 1) for line in props.read_text().splitlines() if " " in line)
            n += int(kv.get("simd_count", "0")) > 0
        except Exception:
            pass
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([d for d in v.split(",") if d.strip() != ""]))
    return n


def maybe_launch_ranks(args):
    """--gpus N > 1 without a torch.distributed.run environment: start N ranks as CHILD processes (this process never
    initialises HIP: devices are counted from sysfs) and exit with their status."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: launch with "
                             f"--nproc-per-node {args.gpus} (or drop the launcher and let bench.py spawn the ranks)")
        return
    if args.gpus <= 1:
        return
    if not args.dry_run_cpu:
        have = visible_gpus()
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible HIP devices, found {have}; "
                             "refusing to report a multi-GPU number from fewer devices")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(Path(__file__).resolve())] + sys.argv[1:]
    log(f"spawning {args.gpus} ranks: {' '.join(cmd)}")
    rc = subprocess.call(cmd)
    raise SystemExit(rc)


# ------------------------------------------------------------------------------------------------------------------
# model / accounting helpers
# ------------------------------------------------------------------------------------------------------------------
def algorithmic_bytes(n, m, p, q_per_step):
    """fp32 bytes per (sequence, step), every API-visible tensor touched once."""
    q = 1 if q_per_step else 0
    fwd = 4 * (n * n * (1 + q) + n * m + p * n + p + m + 1 + 3 * n + 3 * n * n)
    elbo = 4 * (2 * n + n * n + p + m + n * n + n * m + p * n + q * n * n + 1)
    bwd = fwd + 4 * ((n + n * n) + n * n * (1 + q) + n * m + p * n + p)
    return {"smooth_fwd": fwd, "elbo": elbo, "smooth_bwd": bwd}


def elbo_grad_output_bytes(n, m, p, q_per_step):
    """The training-mode ELBO launch also WRITES the gradient of the four terms (g_mus, g_Sigmas, gA, gB, gC, gy and, with a
    per-step Q, gQ) next to the SURVEY's forward reads; the smoother backward then takes g_mus / g_Sigmas as its upstream.
    Reported beside the §8(d) figure (never inside `achieved`) so that `traffic` can be judged against what the launch
    really has to move."""
    q = 1 if q_per_step else 0
    return 4 * (n + n * n + n * n * (1 + q) + n * m + p * n + p + 4)


def lstm_bytes(I, H):
    """alpha-net recurrence: x in, h/c/gates out (fwd); g_h, gates, c in, d_pre/dx out (bwd)."""
    return {"lstm_fwd": 4 * (I + 6 * H), "lstm_bwd": 4 * (10 * H + I)}


def build_model(args, dev):
    import torch
    from kvae.model.model import KVAE
    from kvae.utils.config import KVAEConfig
    torch.manual_seed(0)   # identical replicas on every rank
    cfg = KVAEConfig(dynamics_model=args.dynamics, num_modes=args.modes, z_dim=args.z_dim, a_dim=2, u_dim=args.z_dim)
    model = KVAE(cfg)
    with torch.no_grad():  # spread the K modes so the alpha-net / mixing path carries real gradients
        # ... by a perturbation whose growth over the sequence stays what it is at configs[1]: the spectral radius of
        # I + s * randn(n, n) is about 1 + s * sqrt(n), so s = 0.05 at (n, T) = (4, 50) and 0.00625 at (16, 200).  With a flat
        # 0.05 the n = 16 model is unstable (1.2^200): the switching variant's covariances overflow, the loss is NaN and every
        # step runs the _safe_cholesky ladder and the pivoted solves - a measurement of error paths, not of the workload.
        s = 0.05 * (2.0 / args.z_dim ** 0.5) * min(1.0, 50.0 / args.seq_len)
        model.kalman_filter.dyn_params.A.add_(s * torch.randn_like(model.kalman_filter.dyn_params.A))
        if hasattr(model.kalman_filter.dyn_params, "head_w"):
            model.kalman_filter.dyn_params.head_w.bias.zero_()
    model.beta = 1.0
    return cfg, model.to(dev).train()


def usable_cores():
    """Host cores this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(args, sd, frames, budget_s=25.0):
    """The oracle's restatement of the reference training step on the host cores (kind 'port'), on a bounded sample:
    the full batch at c2 size, the first 32 sequences of the shard (full T) at sizes whose full batch takes minutes."""
    import torch
    from oracle import torch_oracle as O
    threads = usable_cores()
    torch.set_num_threads(threads)
    log(f"cpu_baseline: {threads} threads")
    B, T = frames.shape[:2]
    Bs = B if B * T * args.z_dim ** 2 <= 256 * 50 * 16 * 4 else min(B, 32)
    x = frames[:Bs].float()
    tr = O.OracleTrainer(sd, args.dynamics, lr=1e-3, clip=10.0, beta=1.0, with_metrics=True)
    g = torch.Generator().manual_seed(5)

    def one():
        eps_a = torch.randn(Bs * T, 2, generator=g)
        eps_z = torch.randn(Bs, T, args.z_dim, generator=g)
        gum = None
        if args.dynamics == "switching":
            gum = -torch.empty(Bs, T, args.modes).exponential_(generator=g).log()
        t0 = time.perf_counter()
        tr.step(x, eps_a=eps_a, eps_z=eps_z, gumbel=gum)
        return time.perf_counter() - t0

    w = one()  # warm-up (thread pools, oneDNN primitives)
    log(f"cpu_baseline: warm-up step {w:.2f}s")
    times = [one()]
    while sum(times) + w < budget_s and len(times) < 4:
        times.append(one())
    log(f"cpu_baseline: steps {[round(t, 2) for t in times]}")
    med = sorted(times)[len(times) // 2]
    what = "the same workload" if Bs == B else f"the first {Bs} of the {B} sequences (full T; sequences are independent)"
    return {"value": round(Bs / med, 2), "unit": "sequences/s", "cores": threads, "kind": "port",
            "sample": f"{len(times)} timed steps (after 1 warm-up) of {what}, B={Bs} T={T}, "
                      f"oracle/torch_oracle.OracleTrainer (torch-CPU restatement of the reference step incl. sigmoid and "
                      f"active-unit stats), median"}


def timed_window(trainer, x, steps, world, dev):
    """`steps` training steps bracketed by barrier + device sync on both sides; MAX over ranks, seconds."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = trainer.step(x)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, out


def dry_run_cpu(args):
    """Rehearsal of the launcher + rendezvous path without a GPU (CPU test tier): gloo ranks, one all-reduce."""
    import torch
    import torch.distributed as dist
    from kvae.train.train import init_distributed
    os.environ["KVAE_DIST_BACKEND"] = "gloo"
    rank, world, _ = init_distributed(force_cpu=True)
    if world != args.gpus:
        raise SystemExit(f"bench.py: world size {world} != --gpus {args.gpus}")
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "backend": dist.get_backend() if world > 1 else "none",
                          "allreduce_sum": float(t.item()), "expected_sum": world * (world + 1) / 2,
                          "config": {"workload": workload_name(args, (args.z_dim, args.z_dim, 2))}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------
def roofline_leg(args, cfg, model, x, n_prof_small=10):
    """Roofline of the LGSSM kernel chain: HIP events around each C-ABI call (on the launch stream), eager launches."""
    import torch
    from kvae import _native
    from kvae.train.train import Trainer
    B, T = args.batch, args.seq_len
    eager = Trainer(model, lr=1e-3, use_graph=False, world_size=1, reference_logging=True)   # takes the model over from the captured trainer
    for _ in range(3):
        eager.step(x)
    n_prof = n_prof_small if B * T <= 20000 else 4
    _native.profile_start()
    for _ in range(n_prof):
        eager.step(x)
    times = _native.profile_stop()
    q_per_step = args.dynamics == "switching"
    per_unit = algorithmic_bytes(cfg.z_dim, cfg.u_dim, cfg.a_dim, q_per_step)
    per_unit.update(lstm_bytes(cfg.a_dim, cfg.dynamics_hidden_dim))
    K = args.modes
    per_unit.update({"regime_fwd": 4 * (K * K + 2 * K + 2), "regime_bwd": 4 * (2 * K * K + 3 * K + 2)})
    chain = {}
    for name, ms in times.items():
        # a call may be issued as several chunked launches per step (VAE kernels above 16384 frames):
        # account per STEP (sum of its launches), so that bytes / flop of the whole batch meet the whole time
        launches = max(1, round(len(ms) / n_prof))
        per_step_us = 1e3 * sum(ms) / n_prof
        ent = {"avg_us": round(per_step_us / launches, 2), "launches_per_step": launches,
               "per_step_us": round(per_step_us, 2)}
        if name in per_unit:
            nbytes = per_unit[name] * B * T
            ent.update(algorithmic_bytes=nbytes, GBps=round(nbytes / (per_step_us * 1e-6) / 1e9, 2))
            if name == "elbo":
                ent["grad_output_bytes"] = elbo_grad_output_bytes(cfg.z_dim, cfg.u_dim, cfg.a_dim, q_per_step) * B * T
        chain[name] = ent
    traffic, traffic_src = {}, None
    try:   # HBM bytes per launch from the PMC passes committed under profiles/ (matching config only)
        tj = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
        key = f"B{B}_T{T}_n{cfg.z_dim}_{args.dynamics}_K{args.modes}"
        if key in tj:
            traffic = tj[key]
            traffic_src = f"profiles/pmc_traffic.json[{key}]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes ({tj.get('_round', '?')}), not measured in this run"
    except Exception:
        traffic = {}
    roofline = None
    lg = {k: v for k, v in chain.items() if k in ("smooth_fwd", "smooth_bwd", "elbo")}
    if lg:
        dom = max(lg, key=lambda k: lg[k]["per_step_us"])
        ach = chain[dom]["GBps"]
        n4 = (cfg.z_dim, cfg.u_dim, cfg.a_dim) == (4, 4, 2)
        n16 = (cfg.z_dim, cfg.u_dim, cfg.a_dim) == (16, 16, 2)
        split = n4 and B <= 2048   # lgssm_m4.h: chain sweeps + per-step items as separate launches below KV_M4_SPLIT_MAX_B
        kname = {"smooth_fwd": ("k_smooth_fwd_m4 (filter) + k_gains_m4 + k_smooth_fwd_m4 (smoother)" if split else "k_smooth_fwd_m4")
                 if n4 else ("k_smooth_fwd_n16" if n16 else "k_smooth_fwd"),
                 "smooth_bwd": ("k_smooth_bwd_m4 (two chain sweeps) + k_rts_bwd_items_m4 + k_filter_bwd_items_m4" if split
                                else "k_smooth_bwd_m4") if n4 else ("k_smooth_bwd_n16" if n16 else "k_smooth_bwd"),
                 "elbo": "k_elbo_tpp(+probe)" if n4 else ("k_elbo4_n16 / k_elbo_n16 (+probe)" if n16 else "k_elbo(+probe)")}[dom]
        roofline = {"kernel": kname,
                    "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": traffic.get(dom), "traffic_source": traffic_src,
                    "bytes_per_unit": per_unit[dom], "units_per_launch": B * T, "avg_launch_us": chain[dom]["per_step_us"],
                    "note": "T-deep dependent recursion, " + ("sixteen sequences" if n4 else "one sequence") +
                            " per wavefront: latency-bound, not byte-bound, whenever the batch is far below the "
                            "wave slots of the chip (" + ("B/16" if n4 else "B") + " wavefronts here)" +
                            ("; avg_launch_us is the sum of the launches named in 'kernel' (one C-ABI call)" if split else "")}
        # the kernel that dominates the STEP is outside the LGSSM path: the decoder 32->128 block on the f32 matrix cores
        up = chain.get("dec_up_fwd_s8")
        if up and (cfg.img_size, tuple(cfg.decoder_channels)) == (32, (32, 32, 32)):
            wino = os.environ.get("KVAE_WINO", "1") != "0"
            direct = 2.0 * B * T * 64 * 128 * 288          # MACs x 2: 64 pixels, 128 output channels, 32 x 9 taps
            flop = direct / 2.25 if wino else direct       # Winograd F(2x2,3x3): 16 multiplies per 2x2 tile instead of 36
            tf = flop / (up["per_step_us"] * 1e-6) / 1e12  # all chunk launches of the step together
            roofline["step_dominant_kernel"] = {
                "kernel": ("k_dec_up_fwd_wino<8> (conv 32->128 3x3 as Winograd F(2x2,3x3) + PixelShuffle + ReLU, exact-f32 MFMA)"
                           if wino else "k_dec_up_fwd<8> (conv 32->128 3x3 + PixelShuffle + ReLU, exact-f32 MFMA)"),
                "bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_F32_PEAK_TFS, "unit": "TFLOP/s",
                "frac": round(tf / MFMA_F32_PEAK_TFS, 4), "flop_per_step": flop, "launches_per_step": up["launches_per_step"],
                "per_step_us": up["per_step_us"],
                "note": ("flop = the matrix-core multiplies the kernel really issues (direct convolution / 2.25); "
                         f"direct-convolution-equivalent rate {direct / (up['per_step_us'] * 1e-6) / 1e12:.0f} TFLOP/s; " if wino else "") +
                        "its data-gradient and weight-gradient twins run within 20 % of the same rate (profiles/)"}
        tot_us = sum(c["per_step_us"] for c in lg.values())
        tot_b = sum(c["algorithmic_bytes"] for c in lg.values())
        chain["chain_total"] = {"per_step_us": round(tot_us, 2), "algorithmic_bytes": tot_b,
                                "GBps": round(tot_b / (tot_us * 1e-6) / 1e9, 2)}
    return roofline, chain


def impute_leg(args, cfg, model, x, steps):
    """Eval-mode KVAE.impute (kvae/model/model.py:243-301 there: forward with the block mask of imputation.py:4-12, the two
    emission read-outs, three decoder passes) - what BASELINE configs[3] checks; sequences/s, eager and hipGraph-replayed."""
    import torch
    from kvae.train.imputation import config_block_mask
    B, T = x.shape[:2]
    mask = config_block_mask(cfg, B, T, device=x.device)
    was_training = model.training
    out = {"mask": f"block: t_init {cfg.t_init_mask}, t_steps {cfg.t_steps_mask} (imputation.py:4-12)"}

    def timed(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    s = timed(lambda: model.impute(x, mask), steps)
    out["eager"] = {"ms_per_call": round(1e3 * s, 4), "sequences_per_s": round(B / s, 1)}
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            model.impute(x, mask)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            model.impute(x, mask)
        s = timed(graph.replay, steps)
        out["hipgraph"] = {"ms_per_call": round(1e3 * s, 4), "sequences_per_s": round(B / s, 1)}
        del graph
    except Exception as e:   # report the eager figure only, and why
        out["hipgraph"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    model.train(was_training)
    return out


def run_workload(args, dev, rank, world, full):
    """One workload end to end: build, capture, time `args.steps` steps (barrier + device sync on both sides, max over ranks),
    then - rank 0 - the extra legs.  full: steady-state windows and the CPU baseline at full budget (the headline);
    otherwise a short version of each (the `also` entries of the default run)."""
    import gc
    import torch
    from kvae.train.synthetic import bouncing_ball
    from kvae.train.train import Trainer
    cfg, model = build_model(args, dev)
    sd_cpu = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    B, T = args.batch, args.seq_len
    frames = bouncing_ball(B, T, 1234 + rank)
    x = frames.float().to(dev)

    capture = "hipgraph"
    # lr: the reference's TrainingConfig default (train.py:347).  The noise of every workload starts from the same generator state,
    # whatever ran before it in this process (the `also` entries follow the headline and its CPU leg).
    torch.manual_seed(4321 + rank)
    mk_trainer = lambda graph, w=world, ov=not args.no_overlap: Trainer(
        model, lr=1e-3, use_graph=graph, world_size=w, overlap_lgssm=ov, reference_logging=True,
        graph_allreduce=args.graph_allreduce)
    trainer = mk_trainer(not args.no_graph)
    try:
        for _ in range(max(args.warmup, 1)):
            out = trainer.step(x)
        torch.cuda.synchronize()
    except Exception as e:  # capture not possible on this stack: measure the eager step instead, and say so
        if args.no_graph:
            raise
        print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); falling back to eager launches", file=sys.stderr)
        capture = "eager"
        cfg, model = build_model(args, dev)
        trainer = mk_trainer(False)
        for _ in range(max(args.warmup, 1)):
            out = trainer.step(x)
    if args.no_graph:
        capture = "eager"
    log(f"[{args.config}] warm-up done ({capture}); timing {args.steps} steps")

    # ---- the timed region the contract defines: EXACTLY --steps steps --------------------------------------------
    elapsed, out = timed_window(trainer, x, args.steps, world, dev)
    loss = float(out["loss"])
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B * args.steps / elapsed
    log(f"[{args.config}] timed region: {ms_per_step:.3f} ms/step, {value:.1f} seq/s, loss {loss:.5f}")
    if not (loss == loss and abs(loss) != float("inf")):
        log(f"[{args.config}] WARNING: the loss is not finite - the timing above measured error paths, not the workload")

    # ---- steady state: further windows (>= 5 and >= 2 s or >= 200 steps in all for the headline); median and spread ----
    steady = None
    if not args.no_steady:
        per = max(4, min(200, int(0.4 / max(ms_per_step * 1e-3, 1e-6)) + 1))   # ~0.4 s per window
        min_wins, min_s, min_steps = (5, 2.0, 200) if full else (3, 0.6, 60)
        wins, total_steps, total_s = [], 0, 0.0
        while len(wins) < min_wins or (total_s < min_s and total_steps < min_steps):
            el, _ = timed_window(trainer, x, per, world, dev)
            wins.append(1e3 * el / per)
            total_steps += per
            total_s += el
            if len(wins) >= 40:
                break
        sw = sorted(wins)
        med = sw[len(sw) // 2]
        steady = {"windows": len(wins), "steps_per_window": per, "total_steps": total_steps, "total_s": round(total_s, 3),
                  "ms_per_step_median": round(med, 4), "ms_per_step_min": round(sw[0], 4), "ms_per_step_max": round(sw[-1], 4),
                  "value_median": round(world * B / (med * 1e-3), 2)}
        log(f"[{args.config}] steady state: median {med:.3f} ms/step over {len(wins)} windows of {per} steps "
            f"(min {sw[0]:.3f}, max {sw[-1]:.3f})")

    roofline, chain, impute = None, {}, None
    if rank == 0 and not args.no_roofline:
        roofline, chain = roofline_leg(args, cfg, model, x, n_prof_small=10 if full else 5)
    if rank == 0 and args.impute:
        impute = impute_leg(args, cfg, model, x, steps=max(10, min(50, args.steps)))
        log(f"[{args.config}] impute: {impute}")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and (full or args.cpu_in_also):
        cpu = cpu_baseline(args, sd_cpu, frames, budget_s=25.0 if full else 6.0)
    res = {"value": round(value, 2), "ms_per_step": round(ms_per_step, 4), "steps": args.steps, "warmup": args.warmup,
           "config": {"workload": workload_name(args, (cfg.z_dim, cfg.u_dim, cfg.a_dim)), "preset": args.config,
                      "global_batch": world * B, "seq_len": T, "parallelism": f"dp{world}", "capture": capture,
                      "final_loss": round(loss, 5) if loss == loss and abs(loss) != float("inf") else None,
                      "loss_finite": bool(loss == loss and abs(loss) != float("inf"))},
           "steady_state": steady, "roofline": roofline, "lgssm_chain": chain, "cpu_baseline": cpu}
    if impute is not None:
        res["impute"] = impute
    if cpu:
        res["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 1)
    del trainer, model, x
    gc.collect()
    torch.cuda.empty_cache()
    return res


def main():
    args = parse_args()
    maybe_launch_ranks(args)
    if args.dry_run_cpu:
        return dry_run_cpu(args)

    import copy
    import torch
    import torch.distributed as dist
    from kvae import _native
    from kvae.train.train import init_distributed
    rank, world, dev = init_distributed()
    if dev.type != "cuda":
        raise SystemExit("bench.py needs a HIP device (the LGSSM path has no CPU fallback)")
    if world != args.gpus:
        raise SystemExit(f"bench.py: world size {world} != --gpus {args.gpus}")
    _native.hip_lib()
    backend = dist.get_backend() if world > 1 else "none"
    log(f"rank {rank}/{world} on {torch.cuda.get_device_name(dev)} (device {dev.index}), backend {backend} "
        f"[nccl == RCCL on ROCm]")
    res = run_workload(args, dev, rank, world, full=True)

    # ---- the other BASELINE configs under the same clock: short runs appended as `also` (default invocation only) ----------
    also = {}
    if args.also:
        for name in args.also:
            a = copy.copy(args)
            for k in ("batch", "seq_len", "z_dim", "dynamics", "modes"):
                setattr(a, k, None)
            a.config = name
            apply_preset(a)
            a.steps, a.warmup = ALSO_STEPS.get(name, 30), 5
            a.impute = PRESETS[name].get("impute", False)
            a.cpu_in_also = name.startswith("c4")   # seconds at B = 32; the stress shard's CPU leg is minutes: `--config c5` has it
            t0 = time.perf_counter()
            r = run_workload(a, dev, rank, world, full=False)
            r["wall_s"] = round(time.perf_counter() - t0, 1)
            also[name] = r

    if rank == 0:
        line = {
            "metric": "sequences/sec (ELBO training step, 32x32xT bouncing-ball)", "value": res["value"],
            "unit": "sequences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": dict(res["config"], dist_backend=backend),
            "steady_state": res["steady_state"],
            "roofline": res["roofline"], "lgssm_chain": res["lgssm_chain"], "cpu_baseline": res["cpu_baseline"],
        }
        for k in ("impute", "speedup_vs_cpu_baseline"):
            if k in res:
                line[k] = res[k]
        if also:
            line["also"] = also
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()   # rank 0 ran the extra legs alone: nobody tears the process group down under it
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
