"""-m gpu: `myolo_bn_wgrad_stem` (csrc/stem_wgrad.hip, round 6): the first layer's BatchNorm backward AND weight gradient in one pass over
(gout, y, x) -- dy is never formed -- through the raw C ABI against
  * fp64 first principles on the CPU: autograd of Conv2d(3x3) -> BatchNorm2d(train) -> SiLU w.r.t. the weight, gamma and beta for a given output
    gradient (reference models/common.py:42-43; Focus.conv, common.py:540-551), and
  * the three launches it replaces (`myolo_bn_act_bwd_reduce` + `myolo_bn_act_bwd_apply` + `myolo_conv_wgrad`) on the same inputs,
at the stem's real shape (16 x 256 x 512, 12(16) -> 32), on small / odd tile counts, with a channel-padded x view, gradients accumulated into
non-zero buffers, large |mean| / std ratios (the k0 / k1 cancellation) and loss-scaled output gradients."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import check

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _td(L, t, c=None):
    n, h, w, cc = t.shape
    sn, sh, sw, _ = t.stride()
    return L.Tensor(t.data_ptr(), n, h, w, cc if c is None else c, sn, sh, sw, L.F16, 0)


def _run(B, H, W, cin=12, cout=32, ks=0, gscale=1.0, mean_shift=0.0, seed=0):
    from multiyolov5_amd import _lib as L, engine as E
    lib = L.lib()
    g = torch.Generator().manual_seed(seed)
    xp = torch.zeros(B, H, W, 16)
    xp[..., :cin] = torch.rand(B, H, W, cin, generator=g)
    xp = xp.half()
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.3).half()
    y = (F.conv2d(xp[..., :cin].float().permute(0, 3, 1, 2), w.float(), None, 1, 1).permute(0, 2, 3, 1) + mean_shift).half()   # the saved raw output, as stored
    gout = (torch.randn(B, H, W, cout, generator=g) * 0.1 * gscale).half()
    gamma, beta = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
    yf = y.double().reshape(-1, cout)
    M = yf.shape[0]
    mean, var = yf.mean(0), yf.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-3)
    xhat = (y.double() - mean) * invstd
    z = xhat * gamma.double() + beta.double()
    sg = torch.sigmoid(z)
    gg = gout.double() * (sg * (1 + z * (1 - sg)))
    d0, d1 = gg.sum((0, 1, 2)), (gg * xhat).sum((0, 1, 2))
    dy = (gamma.double() * invstd) * (gg - d0 / M - xhat * d1 / M)
    dw_ref = torch.nn.grad.conv2d_weight(xp[..., :cin].double().permute(0, 3, 1, 2), (cout, cin, 3, 3), dy.permute(0, 3, 1, 2), 1, 1).float()

    xd, yd, gd = xp.to(DEV), y.to(DEV), gout.to(DEV)
    saved = torch.cat([mean, invstd]).float().to(DEV)
    ga, be = gamma.to(DEV), beta.to(DEV)
    wd = L.WgradDesc()
    wd.x = _td(L, xd, 16)
    wd.ntaps, wd.stride, wd.up_shift, wd.ksplit, wd.cout, wd.cin = 9, 1, 0, ks, cout, cin
    E.fill_taps(wd, *E.taps_fwd(3, 1, 1)[:2])
    nbytes = int(lib.myolo_bn_wgrad_stem_ws_bytes())
    ws = torch.empty(nbytes // 4, device=DEV)
    tg, ty = _td(L, gd), _td(L, yd)
    assert lib.myolo_bn_wgrad_stem_ok(C.byref(wd), C.byref(tg)) == 0            # (dw not set yet)
    dw = torch.full((cout, cin, 3, 3), 0.5, device=DEV)
    dga, dbe = torch.full((cout,), 2.0, device=DEV), torch.full((cout,), -1.0, device=DEV)
    wd.dw = dw.data_ptr()
    assert lib.myolo_bn_wgrad_stem_ok(C.byref(wd), C.byref(tg)) == 1
    L.check(lib.myolo_bn_wgrad_stem(C.byref(wd), C.byref(tg), C.byref(ty), L.ptr(saved), L.ptr(ga), L.ptr(be), L.ACT_SILU, L.ptr(dga), L.ptr(dbe),
                                    L.ptr(ws), nbytes, L.stream_ptr()), 'stem')
    torch.cuda.synchronize()
    bad = []
    tag = f'stem_wgrad/{B}x{H}x{W}x{cin}->{cout}/ks{ks}/g{gscale:g}/m{mean_shift:g}'
    check(tag + '/dw', dw.cpu() - 0.5, dw_ref, 2e-3, collect=bad)
    check(tag + '/dgamma', dga.cpu() - 2.0, d1.float(), 1e-3, collect=bad)
    check(tag + '/dbeta', dbe.cpu() + 1.0, d0.float(), 1e-3, collect=bad)
    # the three launches it replaces
    dsum = torch.zeros(L.STAT_COPIES * 2 * cout, device=DEV)
    dyd = torch.empty_like(gd)
    dw2 = torch.zeros(cout, cin, 3, 3, device=DEV)
    dga2, dbe2 = torch.zeros(cout, device=DEV), torch.zeros(cout, device=DEV)
    none = E.null_tensor()
    tdy = _td(L, dyd)
    st = L.stream_ptr()
    L.check(lib.myolo_bn_act_bwd_reduce(C.byref(tg), C.byref(ty), L.ptr(saved), L.ptr(ga), L.ptr(be), L.ACT_SILU, L.ptr(dsum), st))
    L.check(lib.myolo_bn_act_bwd_apply(C.byref(tg), C.byref(ty), L.ptr(saved), L.ptr(ga), L.ptr(be), L.ACT_SILU, L.ptr(dsum), L.ptr(dga2), L.ptr(dbe2), C.byref(tdy),
                                       C.byref(none), 0, st))
    wd2 = L.WgradDesc()
    wd2.x, wd2.dy, wd2.dw = wd.x, tdy, dw2.data_ptr()
    wd2.ntaps, wd2.stride, wd2.up_shift, wd2.ksplit, wd2.cout, wd2.cin = 9, 1, 0, 0, cout, cin
    E.fill_taps(wd2, *E.taps_fwd(3, 1, 1)[:2])
    ws2 = torch.empty(48 << 18, device=DEV)
    wd2.ws, wd2.ws_bytes = ws2.data_ptr(), ws2.numel() * 4
    L.check(lib.myolo_conv_wgrad(C.byref(wd2), st))
    torch.cuda.synchronize()
    check(tag + '/dw_vs_three_launches', dw.cpu() - 0.5, dw2.cpu(), 3e-3, collect=bad)
    check(tag + '/three_launches_vs_ref', dw2.cpu(), dw_ref, 3e-3, collect=bad)
    assert not bad, '\n'.join(bad)


def test_stem_at_its_real_shape():
    _run(16, 256, 512)                              # BASELINE configs[1]: 16 384 tiles over 512 workgroups


@pytest.mark.parametrize('case', [(1, 8, 16), (2, 24, 48), (3, 40, 80), (1, 64, 128)], ids=lambda c: 'x'.join(map(str, c)))
def test_small_and_odd_tile_counts(case):
    _run(*case, seed=1)                             # 1, 18, 75, 64 tiles: fewer tiles than workgroups, odd tiles per split
    _run(*case, ks=7, seed=2)


def test_variants():
    _run(2, 32, 64, cin=16, cout=32, seed=3)        # all 16 input channels real
    _run(2, 32, 64, cin=8, cout=16, seed=4)         # one x segment, two of the four channel segments
    _run(4, 64, 128, gscale=65536.0 / 16, seed=5)   # loss-scaled output gradients (train.py:265 GradScaler)
    _run(4, 64, 128, mean_shift=3.0, seed=6)        # |mean| >> std: the k0 / k1 terms cancel most of A
    _run(2, 64, 64, ks=1024, seed=7)                # more workgroups than tiles allow


def test_layers_that_do_not_qualify_are_refused():
    from multiyolov5_amd import _lib as L, engine as E
    lib = L.lib()
    x = torch.zeros(1, 12, 16, 16, device=DEV, dtype=torch.float16)          # height not a multiple of 8
    g = torch.zeros(1, 12, 16, 32, device=DEV, dtype=torch.float16)
    dw = torch.zeros(32, 12, 3, 3, device=DEV)
    wd = L.WgradDesc()
    wd.x, wd.dw = _td(L, x), dw.data_ptr()
    wd.ntaps, wd.stride, wd.up_shift, wd.cout, wd.cin = 9, 1, 0, 32, 12
    E.fill_taps(wd, *E.taps_fwd(3, 1, 1)[:2])
    tg = _td(L, g)
    assert lib.myolo_bn_wgrad_stem_ok(C.byref(wd), C.byref(tg)) == 0
    assert lib.myolo_bn_wgrad_stem(C.byref(wd), C.byref(tg), C.byref(tg), None, None, None, L.ACT_SILU, None, None, None, 0, L.stream_ptr()) == L.EINVAL
