"""one-launch Conv + BatchNorm + SiLU (myolo_conv_bn_act, round 6) against the two launches it replaces (myolo_conv + myolo_bn_act_fwd), per
layer shape of the bs-16 step: us per call, HIP events around 30 back-to-back calls"""
import ctypes as C
import sys

import torch

sys.path.insert(0, '.')
from multiyolov5_amd import _lib as L, engine as E  # noqa: E402

DEV = 'cuda:0'
lib = L.lib()
SHAPES = [(128, 128, 1, 1, 16, 32, 64), (128, 128, 3, 1, 16, 32, 64), (256, 256, 1, 1, 16, 32, 64), (256, 128, 1, 1, 16, 32, 64), (128, 256, 3, 2, 16, 64, 128),
          (512, 512, 1, 1, 16, 16, 32), (256, 256, 1, 1, 16, 16, 32), (256, 256, 3, 1, 16, 16, 32), (512, 256, 1, 1, 16, 16, 32), (1024, 512, 1, 1, 16, 16, 32)]


def td(t):
    n, h, w, c = t.shape
    sn, sh, sw, _ = t.stride()
    return L.Tensor(t.data_ptr(), n, h, w, c, sn, sh, sw, L.F16, 0)


bar = torch.zeros(19 * 32, dtype=torch.int32, device=DEV)
for (cin, cout, k, s, B, H, W) in SHAPES:
    Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
    x = (torch.randn(B, H, W, cin, device=DEV) * 0.5).half()
    w = (torch.randn(cout, cin, k, k, device=DEV) * (1.0 / (cin * k * k) ** 0.5))
    wp = torch.zeros(cout, k * k, cin, device=DEV, dtype=torch.float16)
    L.check(lib.myolo_pack_weight(L.ptr(w), L.F32, cout, cin, k, k, L.ptr(wp), L.F16, cout, cin, 0, None, L.stream_ptr()))
    y = torch.empty(B, Ho, Wo, cout, device=DEV, dtype=torch.float16)
    o = torch.empty_like(y)
    st = torch.zeros(L.STAT_COPIES * 2 * cout, device=DEV)
    saved = torch.zeros(2 * cout, device=DEV)
    ga, be, rm, rv = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV), torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)
    nbt = torch.zeros(1, dtype=torch.int64, device=DEV)
    dd = L.ConvDesc()
    dd.x, dd.y, dd.w = td(x), td(y), wp.data_ptr()
    dd.cin_pad, dd.cout_pad, dd.wtaps, dd.ntaps, dd.stride, dd.up_shift = cin, cout, k * k, k * k, s, 0
    E.fill_taps(dd, *E.taps_fwd(k, 1, k // 2))
    dd.res, dd.act, dd.stats, dd.accumulate = E.null_tensor(), L.ACT_NONE, st.data_ptr(), 0
    ff = L.BnFwdFuse()
    ff.gamma, ff.beta, ff.running_mean, ff.running_var, ff.nbt, ff.saved = ga.data_ptr(), be.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), saved.data_ptr()
    ff.eps, ff.momentum, ff.act, ff.res, ff.out, ff.barrier = 1e-3, 0.03, 1, E.null_tensor(), td(o), bar.data_ptr()
    none = E.null_tensor()
    ot = td(o)
    stp = L.stream_ptr()

    def two():
        st.zero_()
        lib.myolo_conv(C.byref(dd), stp)
        lib.myolo_bn_act_fwd(C.byref(dd.y), L.ptr(st), L.ptr(ga), L.ptr(be), L.ptr(rm), L.ptr(rv), L.ptr(nbt), L.ptr(saved), C.c_float(1e-3), C.c_float(0.03), 1,
                             C.byref(none), C.byref(ot), stp)

    def one():
        st.zero_()
        lib.myolo_conv_bn_act(C.byref(dd), C.byref(ff), stp)

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
    ok = lib.myolo_conv_bn_act_ok(C.byref(dd))
    print(f'{cin:4d}->{cout:4d} k{k}s{s} @{B}x{Ho}x{Wo}  conv + bn_act_fwd {min(res["two"]):6.1f} us   fused {min(res["fused"]):6.1f} us   (each incl. a ~2 us statistics memset)   '
          f'fused_ok {ok}  timeout {int(bar[18 * 32])}', flush=True)
