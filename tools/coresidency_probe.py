#!/usr/bin/env python3
"""Can a small kernel on a second stream get onto the chip while a persistent Winograd decoder kernel (512 threads x ~232
registers = 480 of a SIMD's 512) runs on the first?  Stream A: the 8x8 block's forward; stream B, started 40 us of spinning later:
a clock-stamp kernel (tools/stamp.hip, a handful of registers), a torch add of two small tensors, one of this library's column-sum
launches - each followed by a stamp.  Printed: when each B kernel finished relative to the start and the end of the A kernel.
usage: hipcc ... tools/stamp.hip -o tools/_bin/libstamp.so; python3 tools/coresidency_probe.py"""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "kalman-vae_amd")]
import torch
from kvae import _native
from kvae.vae.fused import DecoderUp

dev = torch.device("cuda")
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_bin", "libstamp.so"))
lib.kvae_tool_stamp.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
stamps = torch.zeros(16, device=dev, dtype=torch.int64)
A, B = torch.cuda.Stream(), torch.cuda.Stream()
stamp = lambda slot, s: lib.kvae_tool_stamp(stamps.data_ptr(), slot, s.cuda_stream)

N = 12800
g = torch.Generator().manual_seed(1)
x = torch.relu(torch.randn(N, 32, 8, 8, generator=g)).to(dev)
W = (0.08 * torch.randn(128, 32, 3, 3, generator=g)).to(dev)
b = (0.1 * torch.randn(128, generator=g)).to(dev)
p, q = torch.randn(12800, 4, device=dev), torch.randn(12800, 4, device=dev)
part = torch.randn(256, 1152, device=dev)
rows = {k: [] for k in ("A done", "B stamp", "B torch add (51 KB)", "B k_colsum_v4 [256 x 1152]")}
for busy in (True, False):
    for k in rows:
        rows[k].clear()
    for it in range(25):
        torch.cuda.synchronize()
        with torch.cuda.stream(A):
            stamp(0, A)
            if busy:
                DecoderUp.apply(x, W, b)
            stamp(1, A)
        with torch.cuda.stream(B):
            torch.cuda._sleep(80000)   # ~40 us: stream A's kernel is resident before stream B's first launch
            stamp(2, B)
            p + q
            stamp(3, B)
            _native.colsum(part)
            stamp(4, B)
        torch.cuda.synchronize()
        t = stamps.cpu().tolist()
        if it < 5:
            continue
        us = lambda i: (t[i] - t[0]) / 100.0
        rows["A done"].append(us(1)); rows["B stamp"].append(us(2)); rows["B torch add (51 KB)"].append(us(3))
        rows["B k_colsum_v4 [256 x 1152]"].append(us(4))
    print("stream A runs k_dec_up_fwd_wino<8>" if busy else "stream A idle")
    for k, v in rows.items():
        print(f"   {statistics.median(v):8.1f} us after A's first stamp (min {min(v):7.1f}, max {max(v):7.1f})  {k}")
