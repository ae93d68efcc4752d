"""micro-benchmark of myolo_bilinear_bwd at the PyramidPooling shapes (k x k -> 64x128, 32 channels out of a 256-wide buffer)."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiyolov5_amd import _lib as L

dev = torch.device('cuda:0')
lib = L.lib()


def timeit(f, iters=30):
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


n, c, H, W, SW = 16, 32, 64, 128, 256
buf = (torch.randn(n, H, W, SW, device=dev) * 0.1).half()
gd = L.Tensor(L.ptr(buf), n, H, W, c, H * W * SW, W * SW, SW, L.DT[torch.float16], 0)
for k in (1, 2, 3, 6):
    gx = torch.zeros(n, k, k, c, device=dev, dtype=torch.float16)
    xd = L.Tensor(L.ptr(gx), n, k, k, c, k * k * c, k * c, c, L.DT[torch.float16], 0)
    scratch = torch.zeros(n * k * k * c, device=dev)

    def split():
        scratch.zero_()
        L.check(lib.myolo_bilinear_bwd(C.byref(gd), C.byref(xd), 0, L.ptr(scratch), L.stream_ptr()), 's')
    big = lambda: L.check(lib.myolo_bilinear_bwd(C.byref(gd), C.byref(xd), 0, None, L.stream_ptr()), 'b')
    z = timeit(lambda: scratch.zero_())
    print(f'k={k}: split {timeit(split) - z:.1f} us   one-workgroup-per-pixel {timeit(big):.1f} us')
