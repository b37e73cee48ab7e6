"""Which pieces of the training step survive hipGraph capture? Each piece runs in its own subprocess
(hipBLASLt exits the process on a capture violation)."""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
PIECES = ["linear", "conv_enc", "conv_dec", "lstm", "gru", "lgssm", "adam", "full"]

def child(piece):
    sys.path[:0] = [str(ROOT), str(ROOT / "kalman-vae_amd")]
    import torch
    import torch.nn as nn
    try:
        torch.backends.cuda.preferred_blas_library("cublas")
    except Exception as e:
        print("pref blas:", e)
    dev = "cuda"
    from kvae.model.model import KVAE
    from kvae.utils.config import KVAEConfig
    torch.manual_seed(0)
    model = KVAE(KVAEConfig(dynamics_model="switching" if piece == "gru" else "lstm")).to(dev).train()
    B, T = int(os.environ.get("PB", "64")), int(os.environ.get("PT", "20"))
    x = (torch.rand(B, T, 1, 32, 32, device=dev) < 0.1).float()
    a = torch.randn(B, T, 2, device=dev)
    if piece == "linear":
        lin = nn.Linear(512, 2).to(dev); inp = torch.randn(B * T, 512, device=dev)
        fn = lambda: lin(inp).sum().backward()
    elif piece == "conv_enc":
        fn = lambda: sum(o.sum() for o in model.encoder(x.flatten(0, 1))).backward()
    elif piece == "conv_dec":
        fn = lambda: model.decoder(a.flatten(0, 1)).sum().backward()
    elif piece == "lstm":
        fn = lambda: model.kalman_filter.dyn_params.alpha_sequence(a).sum().backward()
    elif piece == "gru":
        fn = lambda: sum(o.sum() for o in model.kalman_filter.dyn_params.markov_regime_posterior(a)).backward()
    elif piece == "lgssm":
        def fn():
            model.kalman_filter.dyn_params.reset_state()
            u = torch.zeros(B, T, 4, device=dev)
            outs = model.kalman_filter.smooth(a, u)
            model.kalman_filter.elbo(outs[0], outs[1], a, u, outs[6], outs[7], outs[8]).backward()
    elif piece == "adam":
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True, fused=True)
        for p in model.parameters():
            p.grad = torch.randn_like(p)
        fn = lambda: opt.step()
    elif piece == "full":
        from kvae.train.train import Trainer
        tr = Trainer(model, use_graph=True)
        for _ in range(3):
            out = tr.step(x)
        torch.cuda.synchronize()
        print("RESULT full OK loss", float(out["loss"]), flush=True)
        return
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    print(f"RESULT {piece} OK", flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for env_name, extra in (("default", {}), ("miopen_rocblas", {"MIOPEN_GEMM_ENFORCE_BACKEND": "1"})):
            for piece in PIECES:
                env = dict(os.environ, DISABLE_ADDMM_CUDA_LT="1", TORCH_BLAS_PREFER_HIPBLASLT="0", **extra)
                r = subprocess.run([sys.executable, __file__, piece], env=env, capture_output=True, text=True, timeout=300)
                tail = [l for l in (r.stdout + r.stderr).splitlines() if l.strip() and "amdgpu.ids" not in l and not l.startswith("Error code")]
                ok = any(l.startswith("RESULT") for l in tail)
                print(f"[{env_name}] {piece}: {'OK' if ok else 'FAIL rc=%d' % r.returncode} :: {tail[-1][:300] if tail else ''}", flush=True)
