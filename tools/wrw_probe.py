"""Time conv weight-gradient (wrw) and data-gradient passes for NCHW vs channels_last operands (MIOpen), decoder/encoder shapes."""
import sys, time, torch
dev = "cuda"
shapes = [("dec3", 12800, 32, 128, 8, 8, 1), ("dec0", 12800, 32, 128, 4, 4, 1), ("dec6", 12800, 32, 4, 16, 16, 1),
          ("enc2", 12800, 32, 32, 16, 16, 2), ("enc0", 12800, 1, 32, 32, 32, 2)]
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for name, N, ci, co, H, W, st in shapes:
    x = torch.randn(N, ci, H, W, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev)
    y = torch.nn.functional.conv2d(x, w, None, st, 1)
    gy = torch.randn_like(y)
    xc, gyc = x.contiguous(memory_format=torch.channels_last), gy.contiguous(memory_format=torch.channels_last)
    wc = w.contiguous(memory_format=torch.channels_last)
    def bw(xx, gg, ww, mask):
        return torch.ops.aten.convolution_backward(gg, xx, ww, None, [st, st], [1, 1], [1, 1], False, [0, 0], 1, mask)
    r = {}
    r["wrw_nchw"] = t(lambda: bw(x, gy, w, [False, True, False]))
    r["wrw_nhwc"] = t(lambda: bw(xc, gyc, wc, [False, True, False]))
    r["bwd_nchw"] = t(lambda: bw(x, gy, w, [True, False, False]))
    r["bwd_nhwc"] = t(lambda: bw(xc, gyc, wc, [True, False, False]))
    r["fwd_nchw"] = t(lambda: torch.nn.functional.conv2d(x, w, None, st, 1))
    r["fwd_nhwc"] = t(lambda: torch.nn.functional.conv2d(xc, wc, None, st, 1))
    r["to_nhwc(x+gy)"] = t(lambda: (x.contiguous(memory_format=torch.channels_last), gy.contiguous(memory_format=torch.channels_last)))
    print(name, {k: round(v, 1) for k, v in r.items()}, flush=True)
