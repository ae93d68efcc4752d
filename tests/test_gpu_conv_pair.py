"""-m gpu: csrc/conv_pair.hip through the raw C ABI (`myolo_conv_pair`, myolo.h): the fused model's Bottleneck -- 1x1 Conv + folded BatchNorm
+ SiLU feeding a 3x3 Conv + folded BatchNorm + SiLU (+ shortcut), reference models/common.py:95-105 with Conv.fuseforward -- in ONE launch,
against (a) torch fp32 on the CPU over the same fp16-rounded input / weights with the intermediate rounded to fp16 where the two-launch form
stores it (2e-3 relative L2 on the fp16-rounded output) and (b) the two-launch form itself (`myolo_conv(a); myolo_conv(b)`, which is the
definition of the result).  Every template variant (64 / 128 / 256 channels, tile heights 4 and 8, one and two mid passes), ragged maps
(tile rows / columns past the image, halo outside on every side), batch > 1, strided channel-slice views for input, output and shortcut,
pairs that do NOT qualify (the library must fall back), and that the fused launch leaves the intermediate tensor untouched."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import check

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _view(L, t, c0, c):
    n, h, w, cc = t.shape
    sn, sh, sw, _ = t.stride()
    return L.Tensor(t.data_ptr() + c0 * 2, n, h, w, c, sn, sh, sw, L.F16, 0)


def _gate(C_, B, H, W, th, cout):
    """csrc/conv_pair.hip pair_try: forced tile heights always fuse; auto only from 96 four-row tiles on and never the two-pass form"""
    if th:
        return True
    bn = 64 if (C_ == 64 or (cout or C_) % 128) else 128
    return C_ // bn == 1 and B * ((W + 15) // 16) * ((H + 3) // 4) >= 96


def _run(C_, B, H, W, th, res=True, cout=None, sliced=False, k2=3, d2=1, seed=0, expect_fused=None, inplace=False):
    from multiyolov5_amd import _lib as L, engine as E
    lib = L.lib()
    cout = cout or C_
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(B, H, W, C_, generator=g) * 0.7).half()
    w1 = (torch.randn(C_, C_, 1, 1, generator=g) * (1.0 / C_ ** 0.5)).half()
    w2 = (torch.randn(cout, C_, k2, k2, generator=g) * (1.0 / (C_ * k2 * k2) ** 0.5)).half()
    sc1, sh1 = torch.rand(C_, generator=g) + 0.5, torch.randn(C_, generator=g) * 0.2
    sc2, sh2 = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
    # torch fp32 over the fp16-rounded operands; the intermediate goes through fp16 like the tensor the two-launch form stores
    xn = x.float().permute(0, 3, 1, 2)
    t = F.silu(F.conv2d(xn, w1.float()) * sc1.view(1, -1, 1, 1) + sh1.view(1, -1, 1, 1)).half().float()
    ref = F.silu(F.conv2d(t, w2.float(), None, 1, d2 * (k2 // 2), d2) * sc2.view(1, -1, 1, 1) + sh2.view(1, -1, 1, 1))
    if res:
        ref = ref + xn[:, :cout]
    ref = ref.permute(0, 2, 3, 1).contiguous()
    # device tensors; `sliced`: x, y and the shortcut are channel slices of wider buffers (concat members)
    pad_c = 64 if sliced else 0
    xb = torch.zeros(B, H, W, C_ + pad_c, dtype=torch.float16, device=DEV)
    xb[..., pad_c // 2:pad_c // 2 + C_] = x.to(DEV)
    yb = torch.full((B, H, W, cout + pad_c), 7.0, dtype=torch.float16, device=DEV)
    tb = torch.full((B, H, W, C_), 3.0, dtype=torch.float16, device=DEV)
    yb2, tb2 = yb.clone(), tb.clone()
    xb2 = xb
    if inplace:        # an in-place Bottleneck: the output IS the input buffer (and the shortcut).  Harmless as two launches, a race in one
        assert cout == C_
        xb, xb2 = xb.clone(), xb.clone()
        yb, yb2 = xb, xb2

    def pack(w, co, k):
        wp = torch.zeros(E.rup(co, 32), k * k, E.rup(C_, 32), device=DEV, dtype=torch.float16)
        L.check(lib.myolo_pack_weight(L.ptr(w.float().to(DEV)), L.F32, co, C_, k, k, L.ptr(wp), L.F16, wp.shape[0], wp.shape[2], 0, None, L.stream_ptr()))
        return wp
    wp1, wp2 = pack(w1, C_, 1), pack(w2, cout, k2)
    k1d = [v.to(DEV).contiguous() for v in (sc1, sh1, sc2, sh2)]

    def descs(yt, tt, xb=None):
        xb = xb_fused if xb is None else xb
        a, b = L.ConvDesc(), L.ConvDesc()
        a.x, a.y, a.w = _view(L, xb, pad_c // 2, C_), _view(L, tt, 0, C_), wp1.data_ptr()
        a.cin_pad, a.cout_pad, a.wtaps, a.ntaps, a.stride, a.up_shift = wp1.shape[2], wp1.shape[0], 1, 1, 1, 0
        E.fill_taps(a, *E.taps_fwd(1, 1, 0))
        a.scale, a.shift, a.act, a.res = k1d[0].data_ptr(), k1d[1].data_ptr(), L.ACT_SILU, E.null_tensor()
        b.x, b.y, b.w = _view(L, tt, 0, C_), _view(L, yt, pad_c // 2, cout), wp2.data_ptr()
        b.cin_pad, b.cout_pad, b.wtaps, b.ntaps, b.stride, b.up_shift = wp2.shape[2], wp2.shape[0], k2 * k2, k2 * k2, 1, 0
        E.fill_taps(b, *E.taps_fwd(k2, d2, d2 * (k2 // 2)))
        b.scale, b.shift, b.act = k1d[2].data_ptr(), k1d[3].data_ptr(), L.ACT_SILU
        b.res = _view(L, xb, pad_c // 2, cout) if res else E.null_tensor()
        return a, b
    xb_fused = xb
    a, b = descs(yb, tb)
    lib.myolo_set_option(b'pair_th', th)
    lib.myolo_trace_start(1)
    try:
        L.check(lib.myolo_conv_pair(C.byref(a), C.byref(b), L.stream_ptr()))
        torch.cuda.synchronize()
    finally:
        lib.myolo_set_option(b'pair_th', 0)
    sites = L.launch_trace()
    lib.myolo_trace_start(0)
    fused = any('cpair' in s for s in sites)
    if expect_fused is None:
        expect_fused = _gate(C_, B, H, W, th, cout)
    assert fused == expect_fused, (sorted(sites), expect_fused)
    a2, b2 = descs(yb2, tb2, xb2)
    L.check(lib.myolo_conv(C.byref(a2), L.stream_ptr()))
    L.check(lib.myolo_conv(C.byref(b2), L.stream_ptr()))
    torch.cuda.synchronize()
    tag = f'pair/{C_}->{cout} k{k2}d{d2} {B}x{H}x{W} th{th}' + ('+res' if res else '') + ('+sliced' if sliced else '')
    bad = []
    lo = pad_c // 2
    check(tag + '/vs_torch', yb[..., lo:lo + cout], ref, 2e-3, collect=bad)
    check(tag + '/vs_two_launches', yb[..., lo:lo + cout], yb2[..., lo:lo + cout], 1e-3, collect=bad)
    check(tag + '/two_launches_vs_torch', yb2[..., lo:lo + cout], ref, 2e-3, collect=bad)
    if sliced and not inplace:                 # the neighbours of the output slice are untouched
        assert bool((yb[..., :lo] == 7.0).all()) and bool((yb[..., lo + cout:] == 7.0).all())
    if fused:                                  # (myolo.h: a->y is NOT written by the fused kernel)
        assert bool((tb == 3.0).all())
    assert not bad, '\n'.join(bad)


CASES = [
    # C, B, H, W, tile height (0 = the library's choice)
    (64, 1, 128, 256, 0),        # 4.m.* of a 2048x1024 frame
    (64, 1, 64, 128, 0),         # ... of a 1024x512 frame (tile height 4: few tiles)
    (64, 2, 37, 53, 8),          # ragged: rows and columns past the image, halo outside on every side
    (64, 2, 37, 53, 4),
    (128, 1, 64, 128, 0),        # 6.m.*
    (128, 1, 64, 128, 8),
    (128, 3, 19, 70, 8),
    (128, 3, 19, 70, 4),
    (256, 1, 32, 64, 4),         # 9.m.0's shape: two mid passes, two N tiles (forced: the auto gate leaves 256 channels to the two-launch form)
    (256, 2, 13, 21, 4),
    (256, 1, 32, 64, 0),         # ... and the auto gate's answer for it (two launches)
    (64, 16, 64, 128, 8),        # batch 16: persistent workgroups walk several tiles
]


@pytest.mark.parametrize('case', CASES, ids=[f'{c[0]}_{c[1]}x{c[2]}x{c[3]}_th{c[4]}' for c in CASES])
@pytest.mark.parametrize('res', [True, False], ids=['shortcut', 'plain'])
def test_conv_pair_matches_torch_and_the_two_launch_form(case, res):
    _run(*case, res=res)


def test_conv_pair_on_channel_slices_and_narrower_outputs():
    _run(128, 2, 24, 40, 8, sliced=True)
    _run(64, 2, 24, 40, 4, sliced=True)
    _run(128, 1, 32, 48, 0, res=False, cout=64)        # 64 output channels from 128 mid channels: a 64-wide N tile
    _run(256, 1, 16, 32, 4, res=False, cout=128)


def test_pairs_that_do_not_qualify_run_as_two_launches():
    _run(32, 1, 64, 96, 0, expect_fused=False)                      # 32 channels (2.m.0)
    _run(64, 1, 32, 48, 0, d2=2, res=False, expect_fused=False)     # dilated 3x3
    _run(64, 1, 32, 48, 0, k2=1, res=False, expect_fused=False)     # 1x1 -> 1x1
    # the output aliases the input halo / the shortcut (in-place Bottleneck): refused even with a forced tile height, and the two
    # launches it falls back to give the reference's numbers
    _run(64, 1, 64, 128, 8, inplace=True, expect_fused=False)
    _run(128, 2, 24, 40, 4, inplace=True, sliced=True, expect_fused=False)
    from multiyolov5_amd import _lib as L
    L.lib().myolo_set_option(b'pair_mode', 0)
    try:
        _run(64, 1, 32, 48, 0, expect_fused=False)
    finally:
        L.lib().myolo_set_option(b'pair_mode', 1)
