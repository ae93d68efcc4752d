"""-m gpu: every block of the hot path through the module API (-> plan -> libmyolo C ABI), forward and backward,
against the CPU oracle restatement on the same seeded inputs.  fp32 parity mode and fp16."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import model_ref
from tests.gpu_util import TOL, check

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _randomize(mod, seed=0):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if p.dim() == 4:
                fan = p.shape[1] * p.shape[2] * p.shape[3]
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * (3.0 / fan) ** 0.5 * 1.4)
            elif n.endswith('bn.weight') or n.endswith('.1.weight'):
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        for n, b in mod.named_buffers():
            if n.endswith('running_mean'):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
            elif n.endswith('running_var'):
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)


def _oracle(fn, mod, xs, training):
    """run an oracle block function on CPU with the module's parameters; returns (out, grads dict, input grads)."""
    sd = {'m.' + k: (v.detach().cpu().float().clone() if v.dtype.is_floating_point else v.detach().cpu().clone())
          for k, v in mod.state_dict().items()}
    params = {k: v.requires_grad_() for k, v in sd.items() if v.dtype.is_floating_point and 'running' not in k}
    ctx = model_ref.Ctx(sd, training, dropout_p=0.0)
    xin = [x.detach().cpu().float().clone().requires_grad_() for x in xs]
    out = fn(ctx, 'm', xin[0] if len(xin) == 1 else xin)
    return out, params, xin, sd


CASES = {
    # name: (ctor, oracle fn, input shapes)
    'conv1x1': (lambda C: C.Conv(64, 128, 1, 1), lambda c, p, x: model_ref.conv_block(c, p, x, 1), [(2, 64, 24, 40)]),
    'conv3x3_small_tiles': (lambda C: C.Conv(16, 32, 3, 1), lambda c, p, x: model_ref.conv_block(c, p, x, 3), [(3, 16, 24, 48)]),   # Focus-sized on 8 x 16 pixel tiles: wgrad_small_halo_kernel
    'conv3x3_small': (lambda C: C.Conv(16, 32, 3, 1), lambda c, p, x: model_ref.conv_block(c, p, x, 3), [(3, 16, 37, 53)]),   # Focus-sized: compact wgrad kernel, ragged pixel count
    'conv3x3': (lambda C: C.Conv(32, 64, 3, 1), lambda c, p, x: model_ref.conv_block(c, p, x, 3), [(2, 32, 20, 36)]),
    'conv3x3s2': (lambda C: C.Conv(64, 128, 3, 2), lambda c, p, x: model_ref.conv_block(c, p, x, 3, 2), [(2, 64, 24, 40)]),
    'conv3x3s2_odd': (lambda C: C.Conv(16, 32, 3, 2), lambda c, p, x: model_ref.conv_block(c, p, x, 3, 2), [(1, 16, 13, 19)]),
    'conv_wide': (lambda C: C.Conv(512, 256, 1, 1), lambda c, p, x: model_ref.conv_block(c, p, x, 1), [(2, 512, 8, 16)]),
    'conv_c48': (lambda C: C.Conv(48, 48, 3, 1), lambda c, p, x: model_ref.conv_block(c, p, x, 3), [(2, 48, 16, 16)]),
    'bottleneck': (lambda C: C.Bottleneck(64, 64, True), lambda c, p, x: model_ref.bottleneck(c, p, x, True), [(2, 64, 16, 32)]),
    'c3': (lambda C: C.C3(64, 64, 2, True), lambda c, p, x: model_ref.c3(c, p, x, 2, True), [(2, 64, 16, 32)]),
    'c3_noshort': (lambda C: C.C3(128, 64, 1, False), lambda c, p, x: model_ref.c3(c, p, x, 1, False), [(2, 128, 8, 16)]),
    'spp': (lambda C: C.SPP(64, 64), lambda c, p, x: model_ref.spp(c, p, x), [(2, 64, 10, 18)]),
    'c3spp': (lambda C: C.C3SPP(64, 96), lambda c, p, x: model_ref.c3spp(c, p, x), [(2, 64, 8, 16)]),
    'rfb2': (lambda C: C.RFB2(96, 64, map_reduce=6), lambda c, p, x: model_ref.rfb2(c, p, x, (2, 3), False), [(2, 96, 12, 20)]),
    'rfb2_global': (lambda C: C.RFB2(128, 64, map_reduce=8, has_globel=True), lambda c, p, x: model_ref.rfb2(c, p, x, (2, 3), True), [(2, 128, 6, 10)]),
    'aspp': (lambda C: C.ASPP(64, 64, d=[3, 6, 9], has_globel=False, map_reduce=4), lambda c, p, x: model_ref.aspp(c, p, x, (3, 6, 9), False), [(2, 64, 12, 20)]),
    'aspps': (lambda C: C.ASPPs(64, 64, d=[3, 6, 9], has_globel=True, map_reduce=4), lambda c, p, x: model_ref.aspps(c, p, x, (3, 6, 9), True), [(2, 64, 12, 20)]),
    'pyramid': (lambda C: C.PyramidPooling(64), lambda c, p, x: model_ref.pyramid_pooling(c, p, x), [(2, 64, 8, 16)]),
    'pyramid_big': (lambda C: C.PyramidPooling(64), lambda c, p, x: model_ref.pyramid_pooling(c, p, x), [(2, 64, 32, 48)]),
    'ffm_k3': (lambda C: C.FFM(64, 32, k=3, is_cat=False), lambda c, p, x: model_ref.ffm(c, p, x, 3), [(2, 64, 8, 16)]),
    'rfb1': (lambda C: C.RFB1(64, 64, map_reduce=4, d=[3, 5, 7]), lambda c, p, x: model_ref.rfb1(c, p, x, (3, 5, 7), False), [(2, 64, 14, 22)]),   # 25-tap 5x5 conv: fwd, dgrad, wgrad
    'rfb1_global': (lambda C: C.RFB1(96, 64, map_reduce=6, has_globel=True), lambda c, p, x: model_ref.rfb1(c, p, x, (3, 5, 7), True), [(2, 96, 8, 12)]),
    'arm': (lambda C: C.ARM(64, 32), lambda c, p, x: model_ref.arm(c, p, x), [(2, 64, 8, 16)]),
    'attention': (lambda C: C.Attention(64), lambda c, p, x: model_ref.attention(c, p, x, 1), [(3, 64, 6, 10)]),
    'attention_r4': (lambda C: C.Attention(64, reduction=4), lambda c, p, x: model_ref.attention(c, p, x, 4), [(2, 64, 6, 10)]),
    'ffm_cat': (lambda C: C.FFM(64, 64, k=1, is_cat=True), lambda c, p, x: model_ref.ffm(c, p, torch.cat(x, 1), 1), [(2, 16, 8, 16), (2, 48, 8, 16)]),
}


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16], ids=['f32', 'f16'])
@pytest.mark.parametrize('training', [True, False], ids=['train', 'eval'])
@pytest.mark.parametrize('name', list(CASES))
def test_block(name, training, dtype):
    from multiyolov5_amd.models import common as C
    from multiyolov5_amd.utils.torch_utils import initialize_weights
    ctor, fn, shapes = CASES[name]
    torch.manual_seed(1)
    mod = ctor(C)
    initialize_weights(mod)
    _randomize(mod)
    mod = mod.to(DEV)
    mod.train(training)
    g = torch.Generator().manual_seed(3)
    xs_cpu = [torch.randn(s, generator=g) for s in shapes]
    xs = [x.to(DEV, dtype).requires_grad_(training) for x in xs_cpu]
    ref_out, ref_params, ref_xin, ref_sd = _oracle(fn, mod, [x.to(dtype) for x in xs_cpu], training)
    out = mod(xs[0] if len(xs) == 1 else xs)
    tol = TOL[dtype]
    bad = []
    check(f'{name}/out', out, ref_out, tol, collect=bad)
    if training:
        r = torch.randn(ref_out.shape, generator=g)
        (ref_out * r).sum().backward()
        (out.float() * r.to(DEV)).sum().backward()
        for i, x in enumerate(xs):
            check(f'{name}/dx{i}', x.grad, ref_xin[i].grad, tol * 2, collect=bad)
        for k, p in mod.named_parameters():
            check(f'{name}/d{k}', p.grad, ref_params['m.' + k].grad, tol * 3, collect=bad)
        for k, b in mod.named_buffers():
            if 'running' in k:
                check(f'{name}/{k}', b, ref_sd['m.' + k], tol, collect=bad)
    assert not bad, '\n'.join(bad)


def test_focus_f32_and_f16():
    from multiyolov5_amd.models import common as C
    from multiyolov5_amd.utils.torch_utils import initialize_weights
    for dtype in (torch.float32, torch.float16):
        mod = C.Focus(3, 32, 3)
        initialize_weights(mod)
        _randomize(mod)
        mod = mod.to(DEV).eval()
        x = torch.rand(2, 3, 32, 48)
        ref_out, *_ = _oracle(lambda c, p, xx: model_ref.focus(c, p, xx), mod, [x.to(dtype)], False)
        out = mod(x.to(DEV, dtype))
        check(f'focus/{dtype}', out, ref_out, TOL[dtype])


@pytest.fixture
def force_stream_kernel():
    """route every qualifying fp16 conv through the streaming kernel (conv_stream.hip), whatever the map size"""
    from multiyolov5_amd import _lib
    _lib.check(_lib.lib().myolo_set_option(b'stream_min_tiles', 1))
    _lib.check(_lib.lib().myolo_set_option(b'halo_off', 1))          # (the k x k layers would otherwise go to the LDS-halo kernel)
    yield
    _lib.check(_lib.lib().myolo_set_option(b'stream_min_tiles', 2048))
    _lib.check(_lib.lib().myolo_set_option(b'halo_off', 0))


@pytest.fixture
def force_halo_kernel():
    """route every qualifying fp16 k x k stride-1 conv through the LDS-halo kernel (conv_halo.hip), whatever the map size"""
    from multiyolov5_amd import _lib
    _lib.check(_lib.lib().myolo_set_option(b'halo_min_tiles', 1))
    yield
    _lib.check(_lib.lib().myolo_set_option(b'halo_min_tiles', 512))


@pytest.mark.parametrize('training', [True, False], ids=['train', 'eval'])
@pytest.mark.parametrize('name', ['conv3x3_small', 'conv3x3_small_tiles', 'conv3x3', 'conv3x3s2', 'conv3x3s2_odd', 'conv_c48', 'bottleneck', 'c3', 'rfb2', 'rfb2_global', 'rfb1', 'aspp', 'aspps',
                                  'ffm_k3', 'arm'])
def test_block_halo_kernel(name, training, force_halo_kernel):
    """the block parity cases with the LDS-halo conv kernel forced on (ragged tiles, dilation 2/3/5/7/9, 5x5, residual, accumulate in
    the dgrad fan-in, BatchNorm statistics, 48-channel padding) -- at the default threshold it serves maps of >= 16 tiles"""
    test_block(name, training, torch.float16)


@pytest.mark.parametrize('training', [True, False], ids=['train', 'eval'])
@pytest.mark.parametrize('name', ['conv1x1', 'conv3x3', 'conv3x3s2', 'conv3x3s2_odd', 'conv_c48', 'bottleneck', 'c3', 'rfb2', 'rfb1', 'aspp', 'ffm_k3'])
def test_block_streaming_kernel(name, training, force_stream_kernel):
    """the same block parity cases with the streaming conv kernel forced on (ragged M, stride 2, dilation, residual,
    accumulate, BatchNorm statistics) -- at the default threshold it only serves maps of >= 65536 pixels"""
    test_block(name, training, torch.float16)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16], ids=['f32', 'f16'])
@pytest.mark.parametrize('name', ['spp', 'c3spp'])
def test_block_spp_large_map_kernels(name, dtype, monkeypatch):
    """maps whose plane does not fit in LDS take the per-output-vector SPP kernels: same parity cases with that path forced"""
    from multiyolov5_amd import _lib as L
    L.lib().myolo_set_option(b'spp_naive', 1)
    try:
        test_block(name, True, dtype)
    finally:
        L.lib().myolo_set_option(b'spp_naive', 0)


def test_spp_pool_bwd_kernel_forms_agree_with_max_pool2d_autograd():
    """myolo_spp_pool_bwd (SPP / C3SPP: nn.MaxPool2d(5 / 9 / 13, 1, pad), common.py:150-163): round 6's channel-lane kernel, round 5's plane
    kernel (`spp_bwd_form` = 1) and the per-output-vector kernel (`spp_naive`) against autograd of F.max_pool2d on the same values -- with
    TIES (a quarter of the inputs repeated: the index planes decide), accumulate, ragged maps, 32 .. 256 channels, both dtypes"""
    import ctypes as C
    import torch.nn.functional as F
    from multiyolov5_amd import _lib as L
    lib = L.lib()

    def td(t, dt):
        n, h, w, c = t.shape
        return L.Tensor(L.ptr(t), n, h, w, c, h * w * c, w * c, c, L.DT[dt], 0)
    try:
        for dt in (torch.float16, torch.float32):
            for (n, h, w, c) in ((2, 16, 32, 256), (1, 10, 18, 32), (3, 7, 9, 64), (1, 20, 40, 96)):
                gen = torch.Generator().manual_seed(h * 131 + c)
                x = (torch.randint(0, 40, (n, h, w, c), generator=gen).float() / 8).to(dt).to(DEV)      # coarse values: many ties
                outs = [torch.empty_like(x) for _ in range(3)]
                idx = torch.empty(3 * x.numel(), dtype=torch.uint8, device=DEV)
                L.check(lib.myolo_spp_pool_fwd(C.byref(td(x, dt)), C.byref(td(outs[0], dt)), C.byref(td(outs[1], dt)), C.byref(td(outs[2], dt)),
                                               L.ptr(idx), L.stream_ptr()), 'spp_fwd')
                gs = [(torch.randn(n, h, w, c, generator=gen) * 0.5).to(dt).to(DEV) for _ in range(3)]
                base = (torch.randn(n, h, w, c, generator=gen) * 0.5).to(dt).to(DEV)
                xr = x.permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
                tot = 0
                for k, g, o in zip((5, 9, 13), gs, outs):
                    y = F.max_pool2d(xr, k, 1, k // 2)
                    assert torch.equal(y.float(), o.permute(0, 3, 1, 2).float().cpu())
                    tot = tot + (y * g.permute(0, 3, 1, 2).double().cpu()).sum()
                tot.backward()
                ref = xr.grad.permute(0, 2, 3, 1).float()
                for form, opts in (('chan', {}), ('plane', {b'spp_bwd_form': 1}), ('naive', {b'spp_naive': 1})):
                    for k_, v_ in opts.items():
                        L.check(lib.myolo_set_option(k_, v_))
                    for acc in (0, 1):
                        gx = base.clone()
                        L.check(lib.myolo_spp_pool_bwd(C.byref(td(gs[0], dt)), C.byref(td(gs[1], dt)), C.byref(td(gs[2], dt)), L.ptr(idx),
                                                       C.byref(td(gx, dt)), acc, L.stream_ptr()), 'spp_bwd')
                        want = ref + (base.float().cpu() if acc else 0)
                        check(f'spp_bwd/{form}/{n}x{h}x{w}x{c}/acc{acc}/{dt}', gx, want, 2e-3 if dt == torch.float16 else 1e-5)
                    for k_ in opts:
                        L.check(lib.myolo_set_option(k_, 0))
    finally:
        lib.myolo_set_option(b'spp_bwd_form', 0)
        lib.myolo_set_option(b'spp_naive', 0)


def test_pyramid_bilinear_bwd_paths_agree():
    """myolo_bilinear_bwd: the split-reduction path (scratch given) and the one-workgroup-per-input-pixel path give the same
    gradient for the PyramidPooling footprints 1x1, 2x2, 3x3, 6x6 -> 64x128 (common.py:534-537), with and without accumulate"""
    import ctypes as C
    from multiyolov5_amd import _lib as L
    lib = L.lib()
    for dt in (torch.float16, torch.float32):
        for k in (1, 2, 3, 6):
            n, c, H, W = 3, 32, 64, 128
            g = (torch.randn(n, H, W, c, device=DEV) * 0.1).to(dt)
            base = (torch.randn(n, k, k, c, device=DEV) * 0.1).to(dt)
            gd = L.Tensor(L.ptr(g), n, H, W, c, H * W * c, W * c, c, L.DT[dt], 0)
            for acc in (0, 1):
                outs = []
                for use_scratch in (True, False):
                    gx = base.clone()
                    xd = L.Tensor(L.ptr(gx), n, k, k, c, k * k * c, k * c, c, L.DT[dt], 0)
                    scratch = torch.zeros(n * k * k * c, device=DEV) if use_scratch else None
                    L.check(lib.myolo_bilinear_bwd(C.byref(gd), C.byref(xd), acc, L.ptr(scratch), L.stream_ptr()), 'bilinear_bwd')
                    outs.append(gx.float())
                # fp64 reference: transpose of F.interpolate(align_corners=True)
                lo = torch.zeros(n, c, k, k, dtype=torch.float64, requires_grad=True)
                up = torch.nn.functional.interpolate(lo, size=(H, W), mode='bilinear', align_corners=True)
                up.backward(g.permute(0, 3, 1, 2).double().cpu())
                ref = lo.grad.permute(0, 2, 3, 1).float() + (base.float().cpu() if acc else 0)
                tol = 2e-3 if dt == torch.float16 else 2e-5
                check(f'bilinear_bwd/split/k{k}/acc{acc}/{dt}', outs[0], ref, tol)
                check(f'bilinear_bwd/big/k{k}/acc{acc}/{dt}', outs[1], ref, tol)


def test_full_resolution_eval_forward_vs_oracle():
    """1x3x512x1024 (the benchmark resolution): fused eval forward in fp16 -- streaming kernel at its real tile counts -- and fp32,
    against the CPU oracle"""
    import os
    from multiyolov5_amd.models.yolo import Model
    from oracle import synth
    from tests.util import CFG, TAGS, load_cfg, synth_sd
    tag = 's_psp'
    sd = synth_sd(tag)
    x = synth.synth_images(1, 512, 1024, seed=3)
    fsd = model_ref.fuse_state_dict({k: v.clone() for k, v in sd.items()})
    with torch.no_grad():
        (rpred, _), rseg = model_ref.forward(load_cfg(tag), fsd, x, training=False)
    for dtype, tol in ((torch.float32, 2e-4), (torch.float16, 3e-2)):
        m = Model(os.path.join(CFG, TAGS[tag]))
        m.load_state_dict(sd, strict=True)
        m = m.to(DEV)
        if dtype == torch.float16:
            m = m.half()
        m.fuse().eval()
        with torch.no_grad():
            (pred, raw), seg = m(x.to(DEV, dtype))
        check(f'fullres/{dtype}/pred', pred, rpred, tol)
        check(f'fullres/{dtype}/seg', seg, rseg, tol)
        if dtype == torch.float32:         # class-index map: identical except at the oracle's own rounding-noise ties
            from tests.test_gpu_model import assert_argmax_exact_or_near_tie
            assert_argmax_exact_or_near_tie('fullres/seg_argmax', seg.argmax(1).cpu(), rseg.argmax(1), rseg, eps=1e-4)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16], ids=['f32', 'f16'])
@pytest.mark.parametrize('name', ['rfb1', 'rfb1_global', 'arm', 'attention', 'attention_r4'])
def test_block_vs_reference_class_golden(name, dtype):
    """RFB1 / ARM / Attention against outputs and gradients written by the reference's OWN classes (models/common.py:177-207,
    416-466; tests/golden/blocks.npz, oracle/make_golden.py blocks_case): train forward, input + parameter gradients, running
    statistics, eval forward"""
    from tests.test_oracle_golden import block_state_dict
    from tests.util import golden
    g = golden('blocks')
    sd, mod = block_state_dict(name)
    mod.load_state_dict(sd, strict=True)
    from multiyolov5_amd.utils.torch_utils import initialize_weights
    initialize_weights(mod)
    mod = mod.to(DEV).train()
    tol = TOL[dtype]
    x = torch.from_numpy(g[f'{name}/x']).to(DEV, dtype).requires_grad_()
    y = mod(x)
    bad = []
    check(f'gold/{name}/train_out', y, g[f'{name}/train_out'], tol, collect=bad)
    (y.float() * torch.from_numpy(g[f'{name}/r']).to(DEV)).sum().backward()
    check(f'gold/{name}/dx', x.grad, g[f'{name}/dx'], tol * 2, collect=bad)
    for k, p in mod.named_parameters():
        check(f'gold/{name}/d{k}', p.grad, g[f'{name}/grad/{k}'], tol * 3, collect=bad)
    for k, b in mod.named_buffers():
        if 'running' in k:
            check(f'gold/{name}/{k}', b, g[f'{name}/after/{k}'], tol, collect=bad)
    mod.load_state_dict({k: v.to(DEV) for k, v in sd.items()}, strict=True)
    mod.eval()
    with torch.no_grad():
        ye = mod(torch.from_numpy(g[f'{name}/x']).to(DEV, dtype))
    check(f'gold/{name}/eval_out', ye, g[f'{name}/eval_out'], tol, collect=bad)
    assert not bad, '\n'.join(bad)


@pytest.mark.parametrize('dt', [torch.float16, torch.float32], ids=['f16', 'f32'])
@pytest.mark.parametrize('case', ['1x1', '3x3', '3x3_acc', '1x1_two_segments', '1x1_smallmap', 's2'])
def test_bn_backward_sums_in_dgrad_epilogue_match_the_reduce_pass(case, dt):
    """myolo_conv_desc.bnb (the BatchNorm-backward reduce pass folded into the dgrad that completes gout) through the raw C ABI: the sums
    the conv launch leaves in `dsum` equal what myolo_bn_act_bwd_reduce computes from the gradient that launch stored -- for the
    streaming, LDS-halo and LDS-tiled kernels, with `accumulate`, with two 32-channel segments (narrower than a 64-wide N tile: the
    library falls back to its internal reduce launch) and for the fused stride-2 dgrad"""
    import ctypes as C
    from multiyolov5_amd import _lib as L
    from multiyolov5_amd import engine as E
    lib = L.lib()
    torch.manual_seed(0)
    n, cin, cout = 2, 64, 64
    k = 1 if case.startswith('1x1') else 3
    H, W = (8, 16) if case == '1x1_smallmap' else (48, 64)
    s2 = case == 's2'
    acc = case == '3x3_acc'
    es = 2 if dt == torch.float16 else 4

    def td(t, c0=0, c=None):
        nn, h, w, cc = t.shape
        return L.Tensor(t.data_ptr() + c0 * es, nn, h, w, cc if c is None else c, h * w * cc, w * cc, cc, L.DT[dt], 0)
    x = (torch.randn(n, H, W, cin, device=DEV) * 0.3).to(dt)                 # "dy" of the consumer layer
    wt = torch.randn(cout, cin, k, k, device=DEV) * (1.0 / (cin * k * k) ** 0.5)
    cin_pad, cout_pad = E.rup(cin, E.KC[dt]), E.rup(cout, 32)
    wp = torch.zeros(cout_pad, k * k, cin_pad, device=DEV, dtype=dt)
    L.check(lib.myolo_pack_weight(L.ptr(wt), L.F32, cout, cin, k, k, L.ptr(wp), L.DT[dt], cout_pad, cin_pad, 0, None, L.stream_ptr()))
    Ho, Wo = (2 * H, 2 * W) if s2 else (H, W)
    gx = (torch.randn(n, Ho, Wo, cout, device=DEV) * 0.1).to(dt) if acc else torch.zeros(n, Ho, Wo, cout, device=DEV, dtype=dt)
    segs = [(0, 32), (32, 64)] if case == '1x1_two_segments' else [(0, 64)]
    yraw = [(torch.randn(n, Ho, Wo, c1 - c0, device=DEV)).to(dt) for c0, c1 in segs]
    saved = [torch.cat([torch.randn(c1 - c0, device=DEV) * 0.2, torch.rand(c1 - c0, device=DEV) + 0.5]) for c0, c1 in segs]
    gam = [torch.rand(c1 - c0, device=DEV) + 0.5 for c0, c1 in segs]
    bet = [torch.randn(c1 - c0, device=DEV) * 0.1 for c0, c1 in segs]
    dsum = [torch.zeros(L.STAT_COPIES * 2 * (c1 - c0), device=DEV) for c0, c1 in segs]
    bnb = (L.BnBwdSeg * len(segs))()
    for i, (c0, c1) in enumerate(segs):
        bnb[i].c0, bnb[i].c1, bnb[i].y = c0, c1, td(yraw[i])
        bnb[i].saved, bnb[i].gamma, bnb[i].beta, bnb[i].dsum, bnb[i].act = saved[i].data_ptr(), gam[i].data_ptr(), bet[i].data_ptr(), \
            dsum[i].data_ptr(), L.ACT_SILU

    def desc(ydesc, taps):
        d = L.ConvDesc()
        d.x, d.y, d.w = td(x), ydesc, wp.data_ptr()
        d.cin_pad, d.cout_pad, d.wtaps, d.ntaps, d.stride, d.up_shift = cin_pad, cout_pad, k * k, len(taps[0]), 1, 0
        E.fill_taps(d, *taps)
        d.res, d.act, d.accumulate = E.null_tensor(), L.ACT_NONE, int(acc)
        return d
    keep = []
    if s2:
        full = td(gx)
        par = []
        for py in range(2):
            for px in range(2):
                taps = E.taps_dgrad(3, 1, 1, 2, py, px)
                sub = L.Tensor(full.ptr + (py * full.sh + px * full.sw) * es, n, (Ho - py + 1) // 2, (Wo - px + 1) // 2, cout,
                               full.sn, full.sh * 2, full.sw * 2, full.dtype, 0)
                par.append(desc(sub, taps))
        par[0].nbnb, par[0].bnb = len(segs), C.cast(bnb, C.POINTER(L.BnBwdSeg))
        arr = (C.POINTER(L.ConvDesc) * 4)(*[C.pointer(g) for g in par])
        keep += [par, arr]
        L.check(lib.myolo_conv_dgrad_s2(arr, 4, L.stream_ptr()), 'myolo_conv_dgrad_s2')
    else:
        d = desc(td(gx), E.taps_fwd(k, 1, k // 2))
        d.nbnb, d.bnb = len(segs), C.cast(bnb, C.POINTER(L.BnBwdSeg))
        L.check(lib.myolo_conv(C.byref(d), L.stream_ptr()), 'myolo_conv')
    torch.cuda.synchronize()
    for i, (c0, c1) in enumerate(segs):
        ref = torch.zeros_like(dsum[i])
        gd, yd = td(gx, c0, c1 - c0), td(yraw[i])
        L.check(lib.myolo_bn_act_bwd_reduce(C.byref(gd), C.byref(yd), L.ptr(saved[i]), L.ptr(gam[i]), L.ptr(bet[i]), L.ACT_SILU, L.ptr(ref),
                                            L.stream_ptr()))
        got = dsum[i].view(L.STAT_COPIES, 2, c1 - c0).sum(0)
        want = ref.view(L.STAT_COPIES, 2, c1 - c0).sum(0)
        assert float(want.abs().max()) > 1e-3
        check(f'bnb/{case}/{dt}/seg{i}', got, want, 2e-4)


def _nhwc_desc(t, c0=0, c=None):
    """myolo_tensor of channels [c0, c0+c) of a dense NHWC tensor"""
    from multiyolov5_amd import _lib as L
    n, h, w, ctot = t.shape
    c = ctot - c0 if c is None else c
    return L.Tensor(t.data_ptr() + c0 * t.element_size(), n, h, w, c, h * w * ctot, w * ctot, ctot, L.DT[t.dtype], 0)


@pytest.mark.parametrize('dt', [torch.float16, torch.float32], ids=['f16', 'f32'])
@pytest.mark.parametrize('shape', [(2, 64, 128, 32), (3, 21, 37, 16), (1, 6, 6, 8), (2, 33, 16, 24)], ids=lambda s: 'x'.join(map(str, s)))
def test_pyramid_grouped_kernels_match_the_single_op_entry_points(shape, dt):
    """myolo_pyramid_upsample_fwd / _bwd and myolo_adaptive_avgpool_bwd_multi (PyramidPooling's four branches in one launch each,
    common.py:521-537) against four calls of myolo_bilinear_fwd / myolo_bilinear_bwd / myolo_adaptive_avgpool_bwd on the same buffers"""
    from multiyolov5_amd import _lib as L
    lib = L.lib()
    n, H, W, cb = shape
    ks = (1, 2, 3, 6)
    g = torch.Generator().manual_seed(H * W)
    st = L.stream_ptr()
    # forward
    xs = [(torch.randn(n, k, k, cb, generator=g)).to(DEV, dt) for k in ks]
    out_a = torch.zeros(n, H, W, 4 * cb + 8, dtype=dt, device=DEV)          # slices of a wider buffer
    out_b = torch.zeros_like(out_a)
    xd = (L.Tensor * 4)(*[_nhwc_desc(x) for x in xs])
    L.check(lib.myolo_pyramid_upsample_fwd(xd, 4, C.byref(_nhwc_desc(out_a, 8, 4 * cb)), st), 'pyr fwd')
    for t in range(4):
        L.check(lib.myolo_bilinear_fwd(C.byref(_nhwc_desc(xs[t])), C.byref(_nhwc_desc(out_b, 8 + t * cb, cb)), st), 'bil fwd')
    assert torch.equal(out_a, out_b)
    # backward of the upsamples
    gout = torch.zeros(n, H, W, 4 * cb + 8, dtype=dt, device=DEV)
    gout[..., 8:] = (torch.randn(n, H, W, 4 * cb, generator=g) * 0.1).to(DEV, dt)
    ga = [torch.full((n, k, k, cb), 0.25, dtype=dt, device=DEV) for k in ks]
    gb = [x.clone() for x in ga]
    acc = (C.c_int32 * 4)(0, 1, 0, 1)
    scratch = torch.zeros(sum(x.numel() for x in ga), dtype=torch.float32, device=DEV)
    gad = (L.Tensor * 4)(*[_nhwc_desc(x) for x in ga])
    rc = lib.myolo_pyramid_upsample_bwd(C.byref(_nhwc_desc(gout, 8, 4 * cb)), gad, 4, acc, L.ptr(scratch), st)
    if 4 * cb // (8 if dt == torch.float16 else 4) > 16:            # more than 16 channel groups: rejected, the engine keeps four launches
        assert rc == L.EINVAL
    else:
        L.check(rc, 'pyr bwd')
        for t in range(4):
            L.check(lib.myolo_bilinear_bwd(C.byref(_nhwc_desc(gout, 8 + t * cb, cb)), C.byref(_nhwc_desc(gb[t])), acc[t], None, st), 'bil bwd')
        tol = 2e-3 if dt == torch.float16 else 2e-5
        for a, b in zip(ga, gb):
            assert float((a.float() - b.float()).abs().max()) <= tol * (float(b.float().abs().max()) + 1e-6)
    # backward of the pools
    pg = [(torch.randn(n, k, k, cb, generator=g)).to(DEV, dt) for k in ks]
    for accum in (0, 1):
        gx_a = torch.full((n, H, W, cb), 0.5, dtype=dt, device=DEV)
        gx_b = gx_a.clone()
        pgd = (L.Tensor * 4)(*[_nhwc_desc(x) for x in pg])
        L.check(lib.myolo_adaptive_avgpool_bwd_multi(pgd, 4, C.byref(_nhwc_desc(gx_a)), accum, st), 'aap multi')
        for t in range(4):
            L.check(lib.myolo_adaptive_avgpool_bwd(C.byref(_nhwc_desc(pg[t])), C.byref(_nhwc_desc(gx_b)), 1 if (t or accum) else 0, st), 'aap')
        assert float((gx_a.float() - gx_b.float()).abs().max()) <= (4e-3 if dt == torch.float16 else 1e-5) * float(gx_b.float().abs().max())


def test_weight_pack_tiled_mode_equals_element_mode():
    """myolo_pack_weights_mt: the LDS-tiled mode (chunk_elems <= 0) writes exactly what the element-per-thread mode writes into the
    valid region, for forward / transposed operands, 1x1 / 3x3 / 5x5 taps, ragged channel counts and a stacked second source"""
    from multiyolov5_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(0)
    jobs = []
    keep = []
    for (co, ci, k, co2) in [(64, 64, 1, 0), (48, 24, 3, 0), (33, 17, 3, 0), (16, 8, 5, 0), (32, 40, 1, 24), (19, 128, 1, 0), (64, 12, 3, 32)]:
        w = torch.randn(co, ci, k, k, generator=g).to(DEV)
        w2 = torch.randn(co2, ci, k, k, generator=g).to(DEV) if co2 else None
        for tr in (0, 1):
            rows, cols = (((co + co2 + 31) // 32) * 32, ((ci + 31) // 32) * 32) if not tr else (((ci + 31) // 32) * 32, ((co + co2 + 31) // 32) * 32)
            for mode in (0, 1, 2):
                dst = torch.zeros(rows, k * k, cols, dtype=torch.float16, device=DEV)
                jobs.append((mode, w, w2, dst, co, ci, k * k, rows, cols, tr))
            keep += [w, w2]
    for mode in (0, 1, 2):                       # 0: element per thread; 1: tiled, LDS tile for MYOLO_MAX_TAPS; 2: tiled, LDS tile for the 25 taps present
        rows_, chunks = [], []
        sel = [j for j in jobs if j[0] == mode]
        for ji, (_, w, w2, dst, co, ci, nt, rp, cp, tr) in enumerate(sel):
            rows_.append((w.data_ptr(), dst.data_ptr(), co, ci, nt, rp, cp, tr, L.F32, L.F16, w2.data_ptr() if w2 is not None else 0,
                          w2.shape[0] if w2 is not None else 0))
            if mode == 0:
                chunks += [(ji, s0) for s0 in range(0, rp * nt * cp, 8192)]
            else:
                ca = co + (w2.shape[0] if w2 is not None else 0)
                tco, tci = (64, 64) if nt == 1 else ((32, 16) if tr else (16, 32))
                chunks += [(ji, t) for t in range(((ca + tco - 1) // tco) * ((ci + tci - 1) // tci))]
        tab = torch.tensor(rows_, dtype=torch.int64).to(DEV)
        ch = torch.tensor(chunks, dtype=torch.int32).to(DEV)
        L.check(lib.myolo_pack_weights_mt(L.ptr(tab), L.ptr(ch), len(chunks), (8192, 0, -25)[mode], L.stream_ptr()), 'pack')
    torch.cuda.synchronize()
    a = [j[3] for j in jobs if j[0] == 0]
    b = [j[3] for j in jobs if j[0] == 1]
    c = [j[3] for j in jobs if j[0] == 2]
    for x, y, z in zip(a, b, c):
        assert torch.equal(x, y) and torch.equal(x, z)
        assert float(x.abs().max()) > 0


@pytest.mark.parametrize('dt', [torch.float16, torch.float32], ids=['f16', 'f32'])
@pytest.mark.parametrize('shape', [(2, 64, 128, 128), (1, 50, 130, 64), (3, 37, 97, 128), (1, 128, 256, 128)], ids=lambda s: 'x'.join(map(str, s)))
def test_adaptive_avgpool_multi_matches_torch(shape, dt):
    """myolo_adaptive_avgpool_fwd_multi (PyramidPooling's AdaptiveAvgPool2d(1), (2), (3), (6) of one map, common.py:521-524, in one pass)
    against F.adaptive_avg_pool2d in fp32 on the same (storage-rounded) input, incl. maps whose sizes are not multiples of k (overlapping bins)"""
    import ctypes as C
    import torch.nn.functional as F
    from multiyolov5_amd import _lib as L
    n, h, w, c = shape
    g = torch.Generator().manual_seed(7)
    x = torch.randn(n, h, w, c, generator=g).to(dt)
    xd = x.to(DEV)
    ks = (1, 2, 3, 6)
    outs = [torch.zeros(n, k, k, c, dtype=dt, device=DEV) for k in ks]
    d = lambda t: L.Tensor(t.data_ptr(), t.shape[0], t.shape[1], t.shape[2], t.shape[3], t.stride(0), t.stride(1), t.stride(2), L.DT[dt], 0)
    arr = (L.Tensor * 4)(*[d(o) for o in outs])
    scratch = torch.zeros(8 * n * sum(k * k for k in ks) * c, dtype=torch.float32, device=DEV)
    xdesc = d(xd)
    L.check(L.lib().myolo_adaptive_avgpool_fwd_multi(C.byref(xdesc), arr, 4, L.ptr(scratch), L.stream_ptr()), 'aap_multi')
    xr = x.float().permute(0, 3, 1, 2)
    for k, o in zip(ks, outs):
        ref = F.adaptive_avg_pool2d(xr, k).permute(0, 2, 3, 1)
        check(f'aap_multi/{shape}/k{k}/{dt}', o, ref, 2e-3 if dt == torch.float16 else 1e-5)
