"""-m gpu: conv_mid.hip (LDS-DMA staged 128-byte K steps, 8 waves per tile) through the raw C ABI (`myolo_conv`, myolo.h) against
torch's fp32 convolution on the CPU over the SAME fp16-rounded inputs and weights: forward conv with BatchNorm statistics, dgrad-shaped
calls (accumulate, residual, strided parity output view), every tile variant, ragged pixel counts, 25 taps, stride 2, dilation.
Tolerance 2e-3 relative L2 on the fp16-rounded output (fp32 accumulation: the only error is the output rounding, ~3e-4)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import check

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _tdesc(L, t, c=None):
    n, h, w, cc = t.shape
    sn, sh, sw, _ = t.stride()
    return L.Tensor(t.data_ptr(), n, h, w, cc if c is None else c, sn, sh, sw, L.F16, 0)


def _run(cin, cout, k, s, d, B, H, W, var, stats=True, accumulate=False, res=False, mode=2, seed=0, midx=0):
    from multiyolov5_amd import _lib as L, engine as E
    lib = L.lib()
    g = torch.Generator().manual_seed(seed)
    Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
    x = (torch.randn(B, H, W, cin, generator=g) * 0.5).half()
    w = (torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)).half()
    y0 = (torch.randn(B, Ho, Wo, cout, generator=g) * 0.3).half()
    r0 = (torch.randn(B, Ho, Wo, cout, generator=g) * 0.3).half()
    pad = d * (k // 2)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), None, s, pad, d).permute(0, 2, 3, 1).contiguous()
    raw = ref.clone()
    if res:
        ref = ref + r0.float()
    if accumulate:
        ref = ref + y0.float()
    xd, yd, rd = x.to(DEV), y0.to(DEV).clone(), r0.to(DEV)
    cin_pad, cout_pad = E.conv_pad(cin, 32, torch.float16), E.conv_pad(cout, 32, torch.float16)
    wp = torch.zeros(cout_pad, k * k, cin_pad, device=DEV, dtype=torch.float16)
    L.check(lib.myolo_pack_weight(L.ptr(w.float().to(DEV)), L.F32, cout, cin, k, k, L.ptr(wp), L.F16, cout_pad, cin_pad, 0, None, L.stream_ptr()))
    st = torch.zeros(L.STAT_COPIES * 2 * cout, device=DEV)
    dd = L.ConvDesc()
    dd.x, dd.y, dd.w = _tdesc(L, xd), _tdesc(L, yd), wp.data_ptr()
    dd.cin_pad, dd.cout_pad, dd.wtaps, dd.ntaps, dd.stride, dd.up_shift = cin_pad, cout_pad, k * k, k * k, s, 0
    E.fill_taps(dd, *E.taps_fwd(k, d, pad))
    dd.res = _tdesc(L, rd) if res else E.null_tensor()
    dd.act, dd.stats, dd.accumulate = L.ACT_NONE, (st.data_ptr() if stats else None), int(accumulate)
    lib.myolo_set_option(b'mid_mode', mode)
    lib.myolo_set_option(b'mid_var', var)
    lib.myolo_set_option(b'midx_mode', midx)               # (conv_midx is consulted first: off unless the test is about it)
    try:
        L.check(lib.myolo_conv(C.byref(dd), L.stream_ptr()))
        torch.cuda.synchronize()
    finally:
        lib.myolo_set_option(b'mid_mode', 2)
        lib.myolo_set_option(b'mid_var', 0)
        lib.myolo_set_option(b'midx_mode', 1)
    bad = []
    tag = f'mid{"x" if midx else ""}/{cin}->{cout} k{k}s{s}d{d} {B}x{H}x{W} var{var}'
    check(tag + '/y', yd, ref, 2e-3, collect=bad)
    if stats:
        ss = st.view(L.STAT_COPIES, 2, cout).sum(0).cpu()
        flat = raw.reshape(-1, cout)
        check(tag + '/sum', ss[0], flat.sum(0), 1e-3, collect=bad)
        check(tag + '/sumsq', ss[1], (flat * flat).sum(0), 1e-4, collect=bad)
    assert not bad, '\n'.join(bad)


SHAPES = [
    # cin, cout, k, s, d, B, H, W
    (128, 128, 3, 1, 1, 2, 32, 64),          # 6.m.0.cv2
    (128, 128, 1, 1, 1, 2, 32, 64),          # two K steps: the ring's prologue + drain only
    (256, 256, 1, 1, 1, 2, 32, 64),          # two N tiles
    (256, 128, 3, 1, 1, 1, 64, 128),         # PSP head 3x3 (K = 2304)
    (256, 256, 3, 1, 1, 2, 16, 32),          # 9.m.0.cv2: M = 1024
    (128, 256, 3, 2, 1, 2, 64, 128),         # 5.conv (stride 2)
    (64, 64, 3, 1, 2, 2, 32, 64),            # dilation 2, 64-wide tiles
    (64, 64, 5, 1, 1, 1, 32, 48),            # RFB1's 5x5: 25 taps, ragged pixel count (1536 = 12 tiles)
    (128, 192, 1, 1, 1, 1, 31, 37),          # ragged: M = 1147, cout 192 (64-wide tiles)
    (512, 512, 1, 1, 1, 2, 16, 32),          # 8 K steps
    (64, 64, 1, 1, 1, 2, 64, 128),           # ONE K step: prologue + drain of a ring that is longer than the loop
    (64, 128, 1, 1, 1, 1, 40, 56),
]


RAGGED_SHAPES = [
    # round 6: yolov5m's 48 / 96-channel layers (models/yolov5m_city_seg.yaml) -- the last 128-byte K step holds 32 or 48 channels, the rest of
    # it comes from the zero page; 96 output channels are an N tile of 128 with 32 masked columns
    (96, 96, 1, 1, 1, 2, 32, 64),            # 2.cv3-like 1x1: ONE ragged K step per tap
    (96, 96, 3, 1, 1, 2, 32, 64),            # 4.m.*.cv2: 3x3 96 -> 96, K = 9 x (64 + 32)
    (48, 96, 3, 2, 1, 2, 64, 128),           # 1.conv: 3x3 stride 2, 48 -> 96
    (96, 192, 3, 2, 1, 1, 64, 64),           # 3.conv
    (48, 48, 1, 1, 1, 2, 64, 128),           # 2.m.*.cv1: one K step of 48 channels
    (304, 256, 1, 1, 1, 2, 32, 64),          # Lab head FFM: 4 x 64 + 48
    (96, 48, 1, 1, 1, 1, 31, 37),            # ragged pixels AND channels
]


@pytest.mark.parametrize('var', [0, 1, 2, 3, 5])
@pytest.mark.parametrize('shape', RAGGED_SHAPES, ids=[f'{s[0]}-{s[1]}k{s[2]}s{s[3]}d{s[4]}_{s[5]}x{s[6]}x{s[7]}' for s in RAGGED_SHAPES])
def test_conv_mid_ragged_last_k_chunk(shape, var):
    from multiyolov5_amd import _lib as L
    L.lib().myolo_trace_start(1)
    _run(*shape, var=var)
    sites = L.launch_trace()
    L.lib().myolo_trace_start(0)
    assert any('mid::launch' in s for s in sites), sites          # really conv_mid (not a fallback to conv_igemm / conv_stream)
    _run(*shape, var=var, stats=False, accumulate=True, res=True, seed=3)


@pytest.mark.parametrize('var', [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize('shape', SHAPES, ids=[f'{s[0]}-{s[1]}k{s[2]}s{s[3]}d{s[4]}_{s[5]}x{s[6]}x{s[7]}' for s in SHAPES])
def test_conv_mid_forward_with_statistics(shape, var):
    _run(*shape, var=var)


M_SHAPES = [
    # yolov5m's widths (models/yolov5m_city_seg.yaml), forced onto the 192-wide N tile of round 5 (var 6; auto takes it from 256 tiles on)
    (96 * 2, 192, 1, 1, 1, 2, 32, 64),       # 4.cv3-like: 192 -> 192
    (192, 192, 3, 1, 1, 2, 32, 64),          # 3x3 192 -> 192: K = 1728
    (192, 384, 1, 1, 1, 2, 32, 64),          # two 192-wide tiles
    (192, 384, 3, 2, 1, 1, 64, 64),          # 5.conv of yolov5m (stride 2)
    (128, 192, 1, 1, 1, 1, 31, 37),          # ragged pixel count
    (384, 192, 1, 1, 1, 8, 64, 128),         # lab reduce at batch 8: 512 tiles (the shape `auto` picks the tile for)
]


@pytest.mark.parametrize('shape', M_SHAPES, ids=[f'{s[0]}-{s[1]}k{s[2]}s{s[3]}_{s[5]}x{s[6]}x{s[7]}' for s in M_SHAPES])
def test_conv_mid_192_wide_tiles(shape):
    from multiyolov5_amd import _lib as L
    _run(*shape, var=6)
    _run(*shape, var=6, stats=False, accumulate=True, res=True)
    if shape[5] == 8:                           # auto selection really is the 192-wide tile here
        L.lib().myolo_trace_start(1)
        _run(*shape, var=0)
        sites = L.launch_trace()
        L.lib().myolo_trace_start(0)
        assert any('BN = 192' in s_ for s_ in sites), sorted(sites)


@pytest.mark.parametrize('var', [1, 2, 3, 4, 5])
def test_conv_mid_dgrad_epilogues(var):
    """the dgrad call shapes: no statistics, accumulate into y, residual add, both"""
    _run(128, 128, 3, 1, 1, 2, 32, 64, var, stats=False, accumulate=True)
    _run(256, 128, 1, 1, 1, 2, 32, 64, var, stats=False, res=True)
    _run(128, 256, 1, 1, 1, 1, 33, 64, var, stats=False, accumulate=True, res=True)


def test_conv_mid_takes_the_layers_and_is_deterministic():
    """mode 1 (default): a layer conv_igemm would run goes to conv_mid -- same call twice gives identical bits (no atomics on y), and
    mode 0 (conv_igemm) agrees within fp16 rounding"""
    from multiyolov5_amd import _lib as L
    lib = L.lib()
    _run(128, 128, 3, 1, 1, 2, 32, 64, 0, mode=1)
    _run(128, 128, 3, 1, 1, 2, 32, 64, 0, mode=0)


@pytest.mark.parametrize('var', [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize('case', ['1x1', '3x3_acc', 'two_segments', 'narrow_segments', 'slice_end', 'wide192', 'two_192'])
def test_conv_mid_bn_backward_sums_in_the_epilogue(case, var):
    """myolo_conv_desc.bnb through conv_mid: the sums the conv launch leaves in `dsum` equal what myolo_bn_act_bwd_reduce computes from the
    gradient that launch stored.  'two_segments': 128 + 128 channels (one N tile each); 'narrow_segments': 64 + 192 (not tile aligned for
    the 128-wide tiles: the library's own reduce launch); 'slice_end': a segment that ends at y.c = 192 inside the last tile"""
    from multiyolov5_amd import _lib as L, engine as E
    lib = L.lib()
    torch.manual_seed(1)
    n, H, W, cin = 2, 24, 40, 128
    k = 3 if case == '3x3_acc' else 1
    cout = {'1x1': 128, '3x3_acc': 128, 'two_segments': 256, 'narrow_segments': 256, 'slice_end': 192, 'wide192': 192, 'two_192': 384}[case]
    segs = {'1x1': [(0, 128)], '3x3_acc': [(0, 128)], 'two_segments': [(0, 128), (128, 256)], 'narrow_segments': [(0, 64), (64, 256)],
            'slice_end': [(0, 64), (64, 192)], 'wide192': [(0, 192)], 'two_192': [(0, 192), (192, 384)]}[case]      # (var 6: the 192-wide N tile)
    acc = case == '3x3_acc'

    def td(t, c0=0, c=None):
        nn, h, w, cc = t.shape
        return L.Tensor(t.data_ptr() + c0 * 2, nn, h, w, cc if c is None else c, h * w * cc, w * cc, cc, L.F16, 0)
    x = (torch.randn(n, H, W, cin, device=DEV) * 0.3).half()
    wt = torch.randn(cout, cin, k, k, device=DEV) * (1.0 / (cin * k * k) ** 0.5)
    cin_pad, cout_pad = E.rup(cin, 32), E.rup(cout, 32)
    wp = torch.zeros(cout_pad, k * k, cin_pad, device=DEV, dtype=torch.float16)
    L.check(lib.myolo_pack_weight(L.ptr(wt), L.F32, cout, cin, k, k, L.ptr(wp), L.F16, cout_pad, cin_pad, 0, None, L.stream_ptr()))
    gx = (torch.randn(n, H, W, cout, device=DEV) * 0.1).half() if acc else torch.zeros(n, H, W, cout, device=DEV, dtype=torch.float16)
    yraw = [torch.randn(n, H, W, c1 - c0, device=DEV).half() for c0, c1 in segs]
    saved = [torch.cat([torch.randn(c1 - c0, device=DEV) * 0.2, torch.rand(c1 - c0, device=DEV) + 0.5]) for c0, c1 in segs]
    gam = [torch.rand(c1 - c0, device=DEV) + 0.5 for c0, c1 in segs]
    bet = [torch.randn(c1 - c0, device=DEV) * 0.1 for c0, c1 in segs]
    dsum = [torch.zeros(L.STAT_COPIES * 2 * (c1 - c0), device=DEV) for c0, c1 in segs]
    bnb = (L.BnBwdSeg * len(segs))()
    for i, (c0, c1) in enumerate(segs):
        bnb[i].c0, bnb[i].c1, bnb[i].y = c0, c1, td(yraw[i])
        bnb[i].saved, bnb[i].gamma, bnb[i].beta, bnb[i].dsum, bnb[i].act = saved[i].data_ptr(), gam[i].data_ptr(), bet[i].data_ptr(), \
            dsum[i].data_ptr(), L.ACT_SILU
    d = L.ConvDesc()
    d.x, d.y, d.w = td(x), td(gx), wp.data_ptr()
    d.cin_pad, d.cout_pad, d.wtaps, d.ntaps, d.stride, d.up_shift = cin_pad, cout_pad, k * k, k * k, 1, 0
    E.fill_taps(d, *E.taps_fwd(k, 1, k // 2))
    d.res, d.act, d.accumulate = E.null_tensor(), L.ACT_NONE, int(acc)
    d.nbnb, d.bnb = len(segs), C.cast(bnb, C.POINTER(L.BnBwdSeg))
    lib.myolo_set_option(b'mid_mode', 2)
    lib.myolo_set_option(b'mid_var', var)
    lib.myolo_set_option(b'midx_mode', 0)
    try:
        L.check(lib.myolo_conv(C.byref(d), L.stream_ptr()), 'myolo_conv')
        torch.cuda.synchronize()
    finally:
        lib.myolo_set_option(b'mid_var', 0)
        lib.myolo_set_option(b'midx_mode', 1)
    for i, (c0, c1) in enumerate(segs):
        ref = torch.zeros_like(dsum[i])
        gd, yd = td(gx, c0, c1 - c0), td(yraw[i])
        L.check(lib.myolo_bn_act_bwd_reduce(C.byref(gd), C.byref(yd), L.ptr(saved[i]), L.ptr(gam[i]), L.ptr(bet[i]), L.ACT_SILU, L.ptr(ref),
                                            L.stream_ptr()))
        got = dsum[i].view(L.STAT_COPIES, 2, c1 - c0).sum(0)
        want = ref.view(L.STAT_COPIES, 2, c1 - c0).sum(0)
        assert float(want.abs().max()) > 1e-3
        check(f'mid_bnb/{case}/var{var}/seg{i}', got, want, 2e-4)


@pytest.mark.parametrize('var', [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize('shape', [(128, 128, 3, 1, 1, 1, 64, 128), (64, 64, 3, 1, 1, 1, 128, 256), (256, 192, 1, 1, 1, 1, 33, 64), (128, 256, 3, 2, 1, 1, 64, 128)],
                         ids=['128-128k3', '64-64k3', '256-192k1_ragged', '128-256k3s2'])
def test_conv_mid_eval_epilogue(shape, var):
    """the eval epilogue (folded BatchNorm scale / shift, SiLU, Bottleneck residual; reference common.py:45-46 fuseforward + :105) of the
    layers conv_mid takes in a detect.py frame, against torch fp32 on the same fp16-rounded operands"""
    from multiyolov5_amd import _lib as L, engine as E
    lib = L.lib()
    cin, cout, k, s, d, B, H, W = shape
    g = torch.Generator().manual_seed(7)
    Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
    x = (torch.randn(B, H, W, cin, generator=g) * 0.5).half()
    w = (torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)).half()
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.3
    r0 = (torch.randn(B, Ho, Wo, cout, generator=g) * 0.3).half()
    pad = d * (k // 2)
    conv = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), None, s, pad, d).permute(0, 2, 3, 1)
    ref = F.silu(conv * scale + shift) + r0.float()
    xd, rd = x.to(DEV), r0.to(DEV)
    yd = torch.zeros(B, Ho, Wo, cout, device=DEV, dtype=torch.float16)
    cin_pad, cout_pad = E.rup(cin, 32), E.rup(cout, 32)
    wp = torch.zeros(cout_pad, k * k, cin_pad, device=DEV, dtype=torch.float16)
    L.check(lib.myolo_pack_weight(L.ptr(w.float().to(DEV)), L.F32, cout, cin, k, k, L.ptr(wp), L.F16, cout_pad, cin_pad, 0, None, L.stream_ptr()))
    sc, sh = scale.to(DEV), shift.to(DEV)
    dd = L.ConvDesc()
    dd.x, dd.y, dd.w = _tdesc(L, xd), _tdesc(L, yd), wp.data_ptr()
    dd.cin_pad, dd.cout_pad, dd.wtaps, dd.ntaps, dd.stride, dd.up_shift = cin_pad, cout_pad, k * k, k * k, s, 0
    E.fill_taps(dd, *E.taps_fwd(k, d, pad))
    dd.res = _tdesc(L, rd)
    dd.act, dd.stats, dd.accumulate = L.ACT_SILU, None, 0
    dd.scale, dd.shift = sc.data_ptr(), sh.data_ptr()
    lib.myolo_set_option(b'mid_mode', 2)
    lib.myolo_set_option(b'mid_var', var)
    lib.myolo_set_option(b'small_off', 1)              # (the split-K small-map kernel would take the batch-1 shapes first)
    try:
        L.check(lib.myolo_conv(C.byref(dd), L.stream_ptr()))
        torch.cuda.synchronize()
    finally:
        lib.myolo_set_option(b'mid_var', 0)
        lib.myolo_set_option(b'small_off', 0)
    check(f'mid_eval/{shape}/var{var}', yd, ref, 2e-3)


@pytest.mark.parametrize('var', [0, 2, 3])
@pytest.mark.parametrize('case', ['k128_n128', 'k256_n256', 'k512_n256_small', 'k128_n64', 'k256_bnb', 'ragged_acc'])
def test_conv_dgrad_with_batchnorm_apply_in_the_operand_path(case, var):
    """myolo_conv_dgrad_bn (round 4: the BatchNorm-backward apply pass of a 1x1 Conv+BN+SiLU layer folded into its dgrad) against the two
    launches it replaces -- myolo_bn_act_bwd_apply then myolo_conv over dy -- on the same inputs: dy identical up to fp16 rounding of
    one fused multiply-add chain, the input gradient 2e-3, dgamma / dbeta 1e-5; with `accumulate`, with a BatchNorm-backward statistics
    segment riding in the epilogue (bnb) and with a ragged pixel count"""
    from multiyolov5_amd import _lib as L, engine as E
    lib = L.lib()
    torch.manual_seed(3)
    K, N, n, H, W = {'k128_n128': (128, 128, 2, 32, 64), 'k256_n256': (256, 256, 2, 32, 64), 'k512_n256_small': (512, 256, 2, 16, 32),
                     'k128_n64': (128, 64, 1, 64, 128), 'k256_bnb': (256, 128, 2, 24, 40), 'ragged_acc': (128, 128, 1, 33, 37)}[case]
    acc = case == 'ragged_acc'
    M = n * H * W

    def td(t):
        nn, h, w, cc = t.shape
        return L.Tensor(t.data_ptr(), nn, h, w, cc, h * w * cc, w * cc, cc, L.F16, 0)
    gout = (torch.randn(n, H, W, K, device=DEV) * 0.3).half()
    y = torch.randn(n, H, W, K, device=DEV).half()
    saved = torch.cat([torch.randn(K, device=DEV) * 0.2, torch.rand(K, device=DEV) + 0.5])
    gam, bet = torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.1
    dsum = torch.zeros(L.STAT_COPIES * 2 * K, device=DEV)
    L.check(lib.myolo_bn_act_bwd_reduce(C.byref(td(gout)), C.byref(td(y)), L.ptr(saved), L.ptr(gam), L.ptr(bet), L.ACT_SILU, L.ptr(dsum), L.stream_ptr()))
    wt = torch.randn(N, K, 1, 1, device=DEV) * (1.0 / K ** 0.5)                  # the dgrad's "weights": [gx channels][dy channels]
    wp = torch.zeros(E.rup(N, 32), 1, K, device=DEV, dtype=torch.float16)
    L.check(lib.myolo_pack_weight(L.ptr(wt), L.F32, N, K, 1, 1, L.ptr(wp), L.F16, E.rup(N, 32), K, 0, None, L.stream_ptr()))
    gx0 = (torch.randn(n, H, W, N, device=DEV) * 0.1).half() if acc else torch.zeros(n, H, W, N, device=DEV, dtype=torch.float16)
    # bnb segment (the layer below): its own y / saved / sums
    yb = torch.randn(n, H, W, N, device=DEV).half()
    savedb = torch.cat([torch.randn(N, device=DEV) * 0.2, torch.rand(N, device=DEV) + 0.5])
    gamb, betb = torch.rand(N, device=DEV) + 0.5, torch.randn(N, device=DEV) * 0.1

    def run(fused):
        gx = gx0.clone()
        dy = torch.zeros_like(gout)
        dg, db = torch.zeros(K, device=DEV), torch.zeros(K, device=DEV)
        dsb = torch.zeros(L.STAT_COPIES * 2 * N, device=DEV)
        d = L.ConvDesc()
        d.x, d.y, d.w = td(gout if fused else dy), td(gx), wp.data_ptr()
        d.cin_pad, d.cout_pad, d.wtaps, d.ntaps, d.stride, d.up_shift = K, E.rup(N, 32), 1, 1, 1, 0
        E.fill_taps(d, [0], [0], [0])
        d.res, d.act, d.accumulate = E.null_tensor(), L.ACT_NONE, int(acc)
        keep = None
        if case == 'k256_bnb':
            bnb = (L.BnBwdSeg * 1)()
            bnb[0].c0, bnb[0].c1, bnb[0].y = 0, N, td(yb)
            bnb[0].saved, bnb[0].gamma, bnb[0].beta, bnb[0].dsum, bnb[0].act = savedb.data_ptr(), gamb.data_ptr(), betb.data_ptr(), dsb.data_ptr(), L.ACT_SILU
            d.nbnb, d.bnb = 1, C.cast(bnb, C.POINTER(L.BnBwdSeg))
            keep = bnb
        if fused:
            f = L.BnApplyFold()
            f.y, f.dy = td(y), td(dy)
            f.saved, f.gamma, f.beta, f.dsum, f.dgamma, f.dbeta, f.act = saved.data_ptr(), gam.data_ptr(), bet.data_ptr(), dsum.data_ptr(), \
                dg.data_ptr(), db.data_ptr(), L.ACT_SILU
            lib.myolo_set_option(b'mid_var', var)
            lib.myolo_set_option(b'mid_bna_strict', int(K <= 256))   # (the one-launch form must really run: no silent two-launch fallback;
                                                                    #  beyond 256 channels the library itself prefers the two launches)
            try:
                L.check(lib.myolo_conv_dgrad_bn(C.byref(d), C.byref(f), L.stream_ptr()), 'myolo_conv_dgrad_bn')
            finally:
                lib.myolo_set_option(b'mid_var', 0)
                lib.myolo_set_option(b'mid_bna_strict', 0)
        else:
            L.check(lib.myolo_bn_act_bwd_apply(C.byref(td(gout)), C.byref(td(y)), L.ptr(saved), L.ptr(gam), L.ptr(bet), L.ACT_SILU, L.ptr(dsum),
                                               L.ptr(dg), L.ptr(db), C.byref(td(dy)), C.byref(E.null_tensor()), 0, L.stream_ptr()))
            L.check(lib.myolo_conv(C.byref(d), L.stream_ptr()), 'myolo_conv')
        torch.cuda.synchronize()
        return gx, dy, dg, db, dsb.view(L.STAT_COPIES, 2, N).sum(0)
    gx_f, dy_f, dg_f, db_f, sb_f = run(True)
    gx_r, dy_r, dg_r, db_r, sb_r = run(False)
    bad = []
    check(f'dgrad_bn/{case}/var{var}/dy', dy_f, dy_r, 1e-3, collect=bad)
    check(f'dgrad_bn/{case}/var{var}/gx', gx_f, gx_r, 2e-3, collect=bad)
    check(f'dgrad_bn/{case}/var{var}/dgamma', dg_f, dg_r, 1e-5, collect=bad)
    check(f'dgrad_bn/{case}/var{var}/dbeta', db_f, db_r, 1e-5, collect=bad)
    if case == 'k256_bnb':
        check(f'dgrad_bn/{case}/var{var}/bnb_sums', sb_f, sb_r, 1e-3, collect=bad)
    # and against first principles (fp32 on the CPU): dy = gamma*istd*(dz - mean(dz) - xhat*mean(dz*xhat))
    g32, y32 = gout.float().cpu(), y.float().cpu()
    mean, istd = saved[:K].cpu(), saved[K:].cpu()
    xh = (y32 - mean) * istd
    z = xh * gam.cpu() + bet.cpu()
    sg = torch.sigmoid(z)
    dz = g32 * (sg * (1 + z * (1 - sg)))
    ref_dy = gam.cpu() * istd * (dz - dz.reshape(-1, K).mean(0) - xh * (dz * xh).reshape(-1, K).mean(0))
    check(f'dgrad_bn/{case}/var{var}/dy_vs_fp32', dy_f, ref_dy, 2e-3, collect=bad)
    ref_gx = ref_dy.half().float().reshape(M, K) @ wp[:N, 0].float().cpu().t()
    if acc:
        ref_gx = ref_gx + gx0.float().cpu().reshape(M, N)
    check(f'dgrad_bn/{case}/var{var}/gx_vs_fp32', gx_f.reshape(M, N), ref_gx, 3e-3, collect=bad)
    assert not bad, '\n'.join(bad)


XSHAPES = [
    # cin, cout, k, s, d, B, H, W
    (64, 64, 3, 1, 1, 2, 64, 128),           # 4.m.0.cv2 (128-byte pixels)
    (128, 128, 3, 1, 1, 2, 32, 64),          # 6.m.0.cv2 (256-byte pixels)
    (256, 256, 3, 1, 1, 8, 16, 32),          # 9.m.0.cv2 (512-byte pixels)
    (256, 128, 3, 1, 1, 1, 64, 128),         # PSP head 3x3
    (64, 64, 3, 1, 2, 2, 32, 64),            # dilation 2
    (64, 64, 3, 1, 3, 2, 32, 64),            # dilation 3
    (64, 64, 5, 1, 1, 1, 40, 72),            # 5x5: 25 taps, ragged tiles (40 = 5 x 8 rows, 72 = 4.5 x 16 columns)
    (128, 192, 3, 1, 1, 1, 37, 53),          # ragged in both directions, 64-wide tiles
]


@pytest.mark.parametrize('var', [0, 1, 2, 3])
@pytest.mark.parametrize('shape', XSHAPES, ids=[f'{s[0]}-{s[1]}k{s[2]}d{s[4]}_{s[5]}x{s[6]}x{s[7]}' for s in XSHAPES])
def test_conv_midx_halo_resident_input(shape, var):
    """conv_midx.hip (the input halo of a k x k stride-1 layer resident in LDS, weights streamed) through myolo_conv against torch fp32:
    forward with BatchNorm statistics and the dgrad epilogues (accumulate + residual)"""
    from multiyolov5_amd import _lib as L
    lib = L.lib()
    lib.myolo_set_option(b'midx_var', var if var else 0)
    if var == 0:
        lib.myolo_set_option(b'midx_var', 0)
    try:
        _run(*shape, var=0, midx=1)
        _run(*shape, var=0, stats=False, accumulate=True, res=True, seed=3, midx=1)
    finally:
        lib.myolo_set_option(b'midx_var', 0)
