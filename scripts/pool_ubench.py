"""the pooling / resize backward launches of the bs-16 step at their real shapes, kernel forms side by side (HIP events, 200 launches each).
usage: python scripts/pool_ubench.py"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiyolov5_amd import _lib as L      # noqa: E402

DEV = 'cuda:0'
lib = L.lib()
dt = torch.float16


def td(t):
    n, h, w, c = t.shape
    return L.Tensor(L.ptr(t), n, h, w, c, h * w * c, w * c, c, L.DT[t.dtype], 0)


def timeit(f, n=200):
    for _ in range(10):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


# SPP (layer 8 of yolov5s at 512x1024, batch 16): 16 x 16 x 32 x 256.  Activations are smooth: neighbouring outputs share their arg-max position
# (white noise would hide the same-address LDS atomics of the plane kernel)
SMOOTH = 'noise' not in sys.argv
if SMOOTH:
    lo = torch.randn(16, 256, 4, 8, device=DEV)
    x = (torch.nn.functional.interpolate(lo, size=(16, 32), mode='bilinear', align_corners=True) + 0.01 * torch.randn(16, 256, 16, 32, device=DEV))
    x = x.permute(0, 2, 3, 1).contiguous().to(dt)
else:
    x = torch.randn(16, 16, 32, 256, device=DEV).to(dt)
print('input:', 'smooth field + 1 % noise' if SMOOTH else 'white noise')
outs = [torch.empty_like(x) for _ in range(3)]
idx = torch.empty(3 * x.numel(), dtype=torch.uint8, device=DEV)
L.check(lib.myolo_spp_pool_fwd(C.byref(td(x)), C.byref(td(outs[0])), C.byref(td(outs[1])), C.byref(td(outs[2])), L.ptr(idx), L.stream_ptr()), 'fwd')
gs = [torch.randn_like(x) for _ in range(3)]
gx = torch.zeros_like(x)
for name, opt in (('channel lanes (round 6)', {}), ('plane (round 5)', {b'spp_bwd_form': 1}), ('per output vector', {b'spp_naive': 1})):
    for k, v in opt.items():
        lib.myolo_set_option(k, v)
    t = timeit(lambda: L.check(lib.myolo_spp_pool_bwd(C.byref(td(gs[0])), C.byref(td(gs[1])), C.byref(td(gs[2])), L.ptr(idx), C.byref(td(gx)), 0, L.stream_ptr()), 'bwd'))
    for k in opt:
        lib.myolo_set_option(k, 0)
    print(f'myolo_spp_pool_bwd 16x16x32x256 f16, {name}: {t:.1f} us')
t = timeit(lambda: L.check(lib.myolo_spp_pool_fwd(C.byref(td(x)), C.byref(td(outs[0])), C.byref(td(outs[1])), C.byref(td(outs[2])), L.ptr(idx), L.stream_ptr()), 'fwd'))
print(f'myolo_spp_pool_fwd 16x16x32x256 f16 (with index planes): {t:.1f} us')
