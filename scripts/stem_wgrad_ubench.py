"""myolo_bn_wgrad_stem (round 6) against the three launches it replaces at the stem's shape (16 x 256 x 512, 12 -> 32), per workgroup count"""
import ctypes as C
import sys

import torch

sys.path.insert(0, '.')
from multiyolov5_amd import _lib as L, engine as E  # noqa: E402

DEV = 'cuda:0'
lib = L.lib()
B, H, W, cin, cout = 16, 256, 512, 12, 32


def td(t):
    n, h, w, c = t.shape
    sn, sh, sw, _ = t.stride()
    return L.Tensor(t.data_ptr(), n, h, w, c, sn, sh, sw, L.F16, 0)


x = torch.rand(B, H, W, 16, device=DEV).half()
y = torch.randn(B, H, W, cout, device=DEV).half()
g = (torch.randn(B, H, W, cout, device=DEV) * 0.1).half()
dy = torch.empty_like(g)
saved = torch.cat([torch.zeros(cout), torch.ones(cout)]).to(DEV)
ga, be = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
dga, dbe = torch.zeros(cout, device=DEV), torch.zeros(cout, device=DEV)
dw = torch.zeros(cout, cin, 3, 3, device=DEV)
dsum = torch.zeros(L.STAT_COPIES * 2 * cout, device=DEV)
nbytes = int(lib.myolo_bn_wgrad_stem_ws_bytes())
ws = torch.empty(nbytes // 4, device=DEV)
ws2 = torch.empty(48 << 18, device=DEV)
wd = L.WgradDesc()
wd.x, wd.dw = td(x), dw.data_ptr()
wd.ntaps, wd.stride, wd.up_shift, wd.ksplit, wd.cout, wd.cin = 9, 1, 0, 0, cout, cin
E.fill_taps(wd, *E.taps_fwd(3, 1, 1)[:2])
wd2 = L.WgradDesc()
wd2.x, wd2.dy, wd2.dw = td(x), td(dy), dw.data_ptr()
wd2.ntaps, wd2.stride, wd2.up_shift, wd2.ksplit, wd2.cout, wd2.cin = 9, 1, 0, 0, cout, cin
E.fill_taps(wd2, *E.taps_fwd(3, 1, 1)[:2])
wd2.ws, wd2.ws_bytes = ws2.data_ptr(), ws2.numel() * 4
tg, ty, tdy, none = td(g), td(y), td(dy), E.null_tensor()
st = L.stream_ptr()


def three():
    lib.myolo_bn_act_bwd_reduce(C.byref(tg), C.byref(ty), L.ptr(saved), L.ptr(ga), L.ptr(be), 1, L.ptr(dsum), st)
    lib.myolo_bn_act_bwd_apply(C.byref(tg), C.byref(ty), L.ptr(saved), L.ptr(ga), L.ptr(be), 1, L.ptr(dsum), L.ptr(dga), L.ptr(dbe), C.byref(tdy), C.byref(none), 0, st)
    lib.myolo_conv_wgrad(C.byref(wd2), st)


def one():
    lib.myolo_bn_wgrad_stem(C.byref(wd), C.byref(tg), C.byref(ty), L.ptr(saved), L.ptr(ga), L.ptr(be), 1, L.ptr(dga), L.ptr(dbe), L.ptr(ws), nbytes, st)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


print(f'reduce + apply + weight gradient: {timeit(three):.1f} us')
for ks in (256, 384, 512, 768, 1024):
    lib.myolo_set_option(b'stem_ks', ks)
    print(f'myolo_bn_wgrad_stem, {ks:4d} workgroups: {timeit(one):.1f} us', flush=True)
