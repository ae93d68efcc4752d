"""standalone timing of the three BN(+SiLU) passes at the training step's layer shapes (no side-stream contention).
usage: python scripts/bn_bench.py"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiyolov5_amd import _lib as L

dev = torch.device('cuda:0')
lib = L.lib()
SHAPES = [(16, 256, 512, 32, 32), (16, 128, 256, 64, 64), (16, 128, 256, 32, 64), (16, 64, 128, 64, 64), (16, 64, 128, 64, 256),
          (16, 64, 128, 128, 128), (16, 32, 64, 128, 128), (16, 32, 64, 128, 256), (16, 32, 64, 256, 256), (16, 16, 32, 256, 512),
          (16, 16, 32, 512, 512)]


def view(n, h, w, c, sw, dt=torch.float16):
    buf = (torch.randn(n, h, w, sw, device=dev) * 0.5).to(dt)
    return buf, L.Tensor(L.ptr(buf), n, h, w, c, h * w * sw, w * sw, sw, L.DT[dt], 0)


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


flush = torch.empty(1 << 28, dtype=torch.uint8, device=dev)        # 256 MB: evict L2 / MALL between calls
tot = [0.0, 0.0, 0.0]
for (n, h, w, c, sw) in SHAPES:
    yb, y = view(n, h, w, c, sw)
    ob, o = view(n, h, w, c, sw)
    gb, g = view(n, h, w, c, sw)
    db, d = view(n, h, w, c, sw)
    stats = torch.rand(L.STAT_COPIES * 2 * c, device=dev) + 1.0
    gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
    saved = torch.cat([torch.randn(c, device=dev) * 0.1, torch.rand(c, device=dev) + 0.5])
    dsum = torch.zeros(L.STAT_COPIES * 2 * c, device=dev)
    dgam, dbet = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    st = L.stream_ptr()
    null = L.Tensor()

    def fwd():
        L.check(lib.myolo_bn_act_fwd(C.byref(y), L.ptr(stats), L.ptr(gamma), L.ptr(beta), None, None, None, L.ptr(saved),
                                     C.c_float(1e-3), C.c_float(0.03), 1, C.byref(null), C.byref(o), st), 'fwd')

    def red():
        L.check(lib.myolo_bn_act_bwd_reduce(C.byref(g), C.byref(y), L.ptr(saved), L.ptr(gamma), L.ptr(beta), 1, L.ptr(dsum), st), 'red')

    def app():
        L.check(lib.myolo_bn_act_bwd_apply(C.byref(g), C.byref(y), L.ptr(saved), L.ptr(gamma), L.ptr(beta), 1, L.ptr(dsum),
                                           L.ptr(dgam), L.ptr(dbet), C.byref(d), C.byref(null), 0, st), 'app')
    e = n * h * w * c * 2
    res = []
    for k, (f, units) in enumerate(((fwd, 2), (red, 2), (app, 3))):
        hot = timeit(f)
        cold = timeit(lambda: (flush.zero_(), f()), iters=5) - timeit(lambda: flush.zero_(), iters=5)
        res.append(f'{hot:6.1f} us hot {e * units / hot / 1e3:5.0f} GB/s | {cold:6.1f} us cold {e * units / cold / 1e3:5.0f} GB/s')
        tot[k] += cold
    print((n, h, w, c, sw), ' fwd', res[0], ' || red', res[1], ' || app', res[2])
print('cold sums (one of each shape):', [round(t) for t in tot])
