"""micro-benchmark of the segmentation CE kernels at the bench shape (16x19x512x1024, NHWC storage).
usage: python scripts/segce_bench.py"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiyolov5_amd import _lib as L
from multiyolov5_amd import synth

dev = torch.device('cuda:0')
N, Cc, H, W = 16, 19, 512, 1024
lib = L.lib()


def timeit(f, iters=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for dt in (torch.float16, torch.float32):
    x = (torch.randn(N, H, W, Cc, device=dev) * 2).to(dt)
    xv = x.permute(0, 3, 1, 2)
    g = torch.empty_like(x)
    gv = g.permute(0, 3, 1, 2)
    m = synth.seg_targets(N, H, W, 19, seed=1).to(dev)
    acc = torch.empty(2, dtype=torch.float64, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    go = torch.ones(1, dtype=torch.float32, device=dev)
    st = L.stream_ptr()
    es = x.element_size()
    fwd = lambda: L.check(lib.myolo_seg_ce_fwd(L.ptr(x), L.DT[dt], N, Cc, H, W, *xv.stride(), L.ptr(m), -1, L.ptr(acc), None, L.ptr(loss), st), 'f')
    fg = lambda: L.check(lib.myolo_seg_ce_fwd_grad(L.ptr(x), L.ptr(g), L.DT[dt], N, Cc, H, W, L.ptr(m), -1, L.ptr(acc), L.ptr(loss), st), 'fg')
    bwd = lambda: L.check(lib.myolo_seg_ce_bwd(L.ptr(x), L.ptr(g), L.DT[dt], N, Cc, H, W, *xv.stride(), *gv.stride(), L.ptr(m), -1, L.ptr(acc),
                                               L.ptr(go), None, None, C.c_float(0.0), st), 'b')
    tf, tfg, tb = timeit(fwd), timeit(fg), timeit(bwd)
    rb = x.numel() * es + m.numel() * 8
    print(f'{dt}: ce_fwd {tf:.1f} us ({rb / tf / 1e3:.0f} GB/s)  ce_fwd_grad {tfg:.1f} us ({(rb + x.numel() * es) / tfg / 1e3:.0f} GB/s)  '
          f'ce_bwd {tb:.1f} us ({(rb + x.numel() * es) / tb / 1e3:.0f} GB/s)   loss {float(loss):.5f}')
