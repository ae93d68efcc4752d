"""one-launch BatchNorm backward (myolo_bn_act_bwd_fused, round 6) against the two launches it replaces, per layer shape of the bs-16 step:
us per call, HIP events around 30 back-to-back calls (the launches of a shape are dependent through dsum / dy like in the step)"""
import ctypes as C
import sys

import torch

sys.path.insert(0, '.')
from multiyolov5_amd import _lib as L  # noqa: E402

DEV = 'cuda:0'
lib = L.lib()
SHAPES = [(16, 16, 32, 128), (16, 16, 32, 256), (16, 16, 32, 512), (16, 32, 64, 128), (16, 32, 64, 256), (16, 64, 128, 64), (8, 64, 128, 128),
          (16, 64, 128, 128), (16, 128, 256, 32)]


def td(t):
    n, h, w, c = t.shape
    sn, sh, sw, _ = t.stride()
    return L.Tensor(t.data_ptr(), n, h, w, c, sn, sh, sw, L.F16, 0)


bar = torch.zeros(19 * 32, dtype=torch.int32, device=DEV)
for (n, h, w, c) in SHAPES:
    y = torch.randn(n, h, w, c, device=DEV).half()
    g = (torch.randn(n, h, w, c, device=DEV) * 0.1).half()
    dy = torch.empty_like(y)
    saved = torch.cat([torch.zeros(c), torch.ones(c)]).to(DEV)
    ga, be = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
    dga, dbe = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    dsum = torch.zeros(L.STAT_COPIES * 2 * c, device=DEV)
    tg, ty, tdy = td(g), td(y), td(dy)
    none = L.Tensor(0, 0, 0, 0, 0, 0, 0, 0, L.F16, 0)
    st = L.stream_ptr()

    def two():
        lib.myolo_bn_act_bwd_reduce(C.byref(tg), C.byref(ty), L.ptr(saved), L.ptr(ga), L.ptr(be), 1, L.ptr(dsum), st)
        lib.myolo_bn_act_bwd_apply(C.byref(tg), C.byref(ty), L.ptr(saved), L.ptr(ga), L.ptr(be), 1, L.ptr(dsum), L.ptr(dga), L.ptr(dbe),
                                   C.byref(tdy), C.byref(none), 0, st)

    def one():
        lib.myolo_bn_act_bwd_fused(C.byref(tg), C.byref(ty), L.ptr(saved), L.ptr(ga), L.ptr(be), 1, L.ptr(dsum), L.ptr(dga), L.ptr(dbe),
                                   C.byref(tdy), C.byref(none), 0, None, L.ptr(bar), st)

    res = {}
    for name, fn in (('two', two), ('fused', one), ('two', two), ('fused', one)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(name, []).append(e0.elapsed_time(e1) / 30 * 1e3)
    ok = lib.myolo_bn_act_bwd_fused_ok(L.F16, n * h * w, c)
    mb = n * h * w * c * 2 / 1e6
    print(f'{n}x{h}x{w}x{c}  {mb:6.1f} MB/tensor  two launches {min(res["two"]):6.1f} us   fused {min(res["fused"]):6.1f} us   fused_ok {ok}   timeout {int(bar[18 * 32])}', flush=True)
