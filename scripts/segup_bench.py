"""micro-benchmark of the seg-head boundary kernels (upsample fwd / bwd) at the bench shape, with a parity check of the
fast channels-last path against the generic strided kernel.  usage: python scripts/segup_bench.py"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiyolov5_amd import _lib as L

dev = torch.device('cuda:0')
N, Cc, h, w, H, W = 16, 19, 64, 128, 512, 1024
lib, st = L.lib(), None


def tdesc(t_nhwc):
    n, hh, ww, c = t_nhwc.shape
    return L.Tensor(L.ptr(t_nhwc), n, hh, ww, c, hh * ww * c, ww * c, c, L.DT[t_nhwc.dtype], 0)


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
    g = (torch.randn(N, H, W, Cc, device=dev) * 0.1).to(dt)          # NHWC storage
    gl = torch.zeros(N, h, w, Cc, device=dev, dtype=dt)
    gl2 = torch.zeros_like(gl)
    d, d2 = tdesc(gl), tdesc(gl2)
    gv = g.permute(0, 3, 1, 2)
    sn, sc, sh, sw = gv.stride()

    def run(desc):
        L.check(lib.myolo_seg_upsample_bwd(L.ptr(g), L.DT[dt], H, W, sn, sc, sh, sw, C.byref(desc), 0, None, L.stream_ptr()), "bwd")
    us = timeit(lambda: run(d))
    os.environ['MYOLO_NO_FAST_UPB'] = '1'
    us2 = timeit(lambda: run(d2), iters=3)
    del os.environ['MYOLO_NO_FAST_UPB']
    err = (gl.float() - gl2.float()).abs().max().item() / gl2.float().abs().max().item()
    print(f'{dt}: seg_upsample_bwd fast {us:.1f} us ({g.numel() * g.element_size() / us / 1e3:.0f} GB/s)  generic {us2:.1f} us  rel err {err:.2e}')
