"""Time the conv VAE halves (fwd+bwd, B*T=12800 frames) under MIOpen/PyTorch settings."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "kalman-vae_amd")]
import torch
from kvae.model.model import KVAE
from kvae.utils.config import KVAEConfig
dev = "cuda"
variant = sys.argv[1] if len(sys.argv) > 1 else "default"
if "bench" in variant:
    torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
model = KVAE(KVAEConfig(dynamics_model="lstm")).to(dev).train()
if "cl" in variant:
    model = model.to(memory_format=torch.channels_last)
N = 12800
x = (torch.rand(N, 1, 32, 32, device=dev) < 0.1).float()
a = torch.randn(N, 2, device=dev)
if "cl" in variant:
    x = x.contiguous(memory_format=torch.channels_last)
def enc():
    mu, var = model.encoder(x); (mu.sum() + var.sum()).backward()
def dec():
    model.decoder(a).sum().backward()
for name, fn in (("encoder", enc), ("decoder", dec)):
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize(); warm = time.perf_counter() - t0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    print(f"[{variant}] {name}: {s.elapsed_time(e) / 10:.3f} ms/iter (warm-up {warm:.1f}s)", flush=True)
