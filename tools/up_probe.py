#!/usr/bin/env python3
"""Checks / times the register-stationary MFMA decoder blocks against torch (conv2d + pixel_shuffle + relu)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kalman-vae_amd"))
import torch, torch.nn.functional as F
from kvae import _native as N
dev = torch.device("cuda")
lib = N.hip_lib()
vp = C.c_void_p
lib.dll.kvae_dec_up_fwd.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, vp]
for side, n in ((8, 5), (4, 11), (8, 12800), (4, 12800)):
    g = torch.Generator().manual_seed(side + n)
    x = torch.relu(torch.randn(n, 32, side, side, generator=g)).to(dev)
    W = (0.08 * torch.randn(128, 32, 3, 3, generator=g)).to(dev)
    b = (0.1 * torch.randn(128, generator=g)).to(dev)
    out = torch.empty(n, 32, 2 * side, 2 * side, device=dev)
    st = vp(torch.cuda.current_stream().cuda_stream)
    rc = lib.dll.kvae_dec_up_fwd(vp(x.data_ptr()), vp(W.data_ptr()), vp(b.data_ptr()), vp(out.data_ptr()), n, 32, side, st)
    torch.cuda.synchronize()
    ref = torch.relu(F.pixel_shuffle(F.conv2d(x, W, b, padding=1), 2))
    err = float((out - ref).abs().max() / ref.abs().max())
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        lib.dll.kvae_dec_up_fwd(vp(x.data_ptr()), vp(W.data_ptr()), vp(b.data_ptr()), vp(out.data_ptr()), n, 32, side, st)
    e.record(); torch.cuda.synchronize()
    print(f"side {side} frames {n}: rc {rc} rel err {err:.2e}  {a.elapsed_time(e) / 10 * 1e3:.1f} us", flush=True)
