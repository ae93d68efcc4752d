"""CPU-side checks: the C-ABI library builds/loads and exports every symbol include/myolo.h declares; host logic
(plan construction, state_dict surface) works without a GPU; the product refuses CPU tensors loudly."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests.util import ROOT, TAGS, golden, load_cfg


def test_library_exports_every_declared_symbol():
    from multiyolov5_amd import build, _lib
    build.build(verbose=False)
    hdr = open(os.path.join(ROOT, 'include', 'myolo.h')).read()
    names = set(re.findall(r'\b(myolo_[a-z0-9_]+)\s*\(', hdr))
    assert len(names) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in sorted(names):
        assert hasattr(lib, n), f'{n} declared in include/myolo.h but not exported by libmyolo.so'
    bound = set(_lib._PROTOS)
    assert names == bound, (names - bound, bound - names)
    l = _lib.lib()
    assert l.myolo_version() == 1 and l.myolo_arch() == b'gfx950'


def test_struct_layouts_match_header():
    from multiyolov5_amd import _lib
    assert ctypes.sizeof(_lib.Tensor) == 56
    # myolo_conv_desc: x, y (56 each), w (8), 6 ints, 3 tap tables of 25 ints (-> 500, 8-aligned: +4 pad), scale, shift (8 each),
    # act + accumulate (8), res (56), stats (8), det_no + nbnb (8), bnb (8)
    assert ctypes.sizeof(_lib.ConvDesc) == 56 * 2 + 8 + 6 * 4 + 3 * 25 * 4 + 4 + 8 + 8 + 8 + 56 + 8 + 8 + 8
    # the C compiler's view of include/myolo.h (gcc is part of the image): sizes and a few offsets of the descriptor structs
    import shutil, subprocess, tempfile
    if shutil.which('gcc'):
        src = ('#include <stdio.h>\n#include <stddef.h>\n#include "myolo.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(myolo_tensor), '
               'sizeof(myolo_conv_desc), sizeof(myolo_wgrad_desc), offsetof(myolo_conv_desc, scale), offsetof(myolo_conv_desc, res), '
               'offsetof(myolo_wgrad_desc, ws));printf("%zu %zu %zu\\n", sizeof(myolo_bn_bwd_seg), offsetof(myolo_conv_desc, bnb), '
               'offsetof(myolo_bn_bwd_seg, dsum));printf("%zu %zu %zu %zu\\n", sizeof(myolo_bn_split), offsetof(myolo_bn_split, dbeta2), sizeof(myolo_seg_sync_desc), offsetof(myolo_seg_sync_desc, lab_lut));printf("%zu %zu %zu\\n", sizeof(myolo_mosaic_desc), offsetof(myolo_mosaic_desc, M), offsetof(myolo_mosaic_desc, out_hwc));printf("%zu %zu %zu\\n", sizeof(myolo_prog_op), offsetof(myolo_prog_op, cond), offsetof(myolo_prog_op, a));printf("%zu %zu %zu %zu\\n", sizeof(myolo_tiny_conv_desc), offsetof(myolo_tiny_conv_desc, eps), offsetof(myolo_tiny_conv_desc, gout), offsetof(myolo_tiny_conv_desc, dbeta));return 0;}\n')
        with tempfile.TemporaryDirectory() as td:
            open(os.path.join(td, 'a.c'), 'w').write(src)
            inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include')
            subprocess.check_call(['gcc', '-I', inc, os.path.join(td, 'a.c'), '-o', os.path.join(td, 'a')])
            sizes = [int(v) for v in subprocess.check_output([os.path.join(td, 'a')]).split()]
        assert sizes == [ctypes.sizeof(_lib.Tensor), ctypes.sizeof(_lib.ConvDesc), ctypes.sizeof(_lib.WgradDesc),
                         _lib.ConvDesc.scale.offset, _lib.ConvDesc.res.offset, _lib.WgradDesc.ws.offset,
                         ctypes.sizeof(_lib.BnBwdSeg), _lib.ConvDesc.bnb.offset, _lib.BnBwdSeg.dsum.offset,
                         ctypes.sizeof(_lib.BnSplit), _lib.BnSplit.dbeta2.offset, ctypes.sizeof(_lib.SegSyncDesc),
                         _lib.SegSyncDesc.lab_lut.offset, ctypes.sizeof(_lib.MosaicDesc), _lib.MosaicDesc.M.offset,
                         _lib.MosaicDesc.out_hwc.offset, ctypes.sizeof(_lib.ProgOp), _lib.ProgOp.cond.offset, _lib.ProgOp.a.offset,
                         ctypes.sizeof(_lib.TinyConvDesc), _lib.TinyConvDesc.eps.offset, _lib.TinyConvDesc.gout.offset,
                         _lib.TinyConvDesc.dbeta.offset], sizes
    # the native executor rejects a malformed program on the host: unknown function id, wrong argument count, unknown kind
    bad = _lib.ProgOp()
    bad.kind, bad.fn, bad.nargs = _lib.OP_CALL, 10 ** 6, 1
    assert not _lib.lib().myolo_prog_create(ctypes.pointer(bad), 1)
    bad.fn, bad.nargs = _lib.lib().myolo_prog_fn_id(b'myolo_conv'), 7
    assert not _lib.lib().myolo_prog_create(ctypes.pointer(bad), 1)
    bad.kind = 9
    assert not _lib.lib().myolo_prog_create(ctypes.pointer(bad), 1)
    assert _lib.lib().myolo_prog_fn_id(b'myolo_not_there') == -1 and _lib.lib().myolo_nms_ws_bytes(0, 10) == 0
    # invalid descriptors are rejected on the host before any launch (no GPU needed)
    d = _lib.ConvDesc()
    assert _lib.lib().myolo_conv(ctypes.byref(d), None) == -22
    w = _lib.WgradDesc()
    assert _lib.lib().myolo_conv_wgrad(ctypes.byref(w), None) == -22
    t = (_lib.TinyConvDesc * 1)()
    assert _lib.lib().myolo_tiny_conv_fwd(t, 1, None) == -22 and _lib.lib().myolo_tiny_conv_bwd(t, 0, None) == -22


@pytest.mark.parametrize('tag', list(TAGS))
def test_model_surface_matches_reference(tag):
    from multiyolov5_amd.models.yolo import Model, Detect
    from oracle import shapes
    from tests.util import CFG
    m = Model(os.path.join(CFG, TAGS[tag]))
    sd = m.state_dict()
    ref = shapes.state_shapes(load_cfg(tag))
    assert set(sd) == set(ref)
    assert all(tuple(sd[k].shape) == tuple(ref[k]) for k in ref)
    g = golden('model_' + tag)
    assert sum(p.numel() for p in m.parameters()) == int(g['n_params'])
    k0 = [k for k in sd if k.endswith('running_var')][0]
    np.testing.assert_allclose(sd[k0][:4].numpy(), g['init_running_var0'], rtol=1e-6)        # stride-probe side effect
    assert int(sd[k0.replace('running_var', 'num_batches_tracked')]) == int(g['init_nbt0'])
    np.testing.assert_allclose(sd['model.25.anchors'].numpy(), g['init_anchors'], rtol=1e-6)
    assert 24 in m.save and (tag != 's_psp' or m.save == [4, 6, 10, 14, 16, 17, 19, 20, 22, 23, 24])
    assert m.stride.tolist() == [8., 16., 32.]
    det = m.model[-1]
    assert isinstance(det, Detect) and (det.nl, det.na, det.nc, det.no) == (3, 3, 10, 15)
    bn = [x for x in m.modules() if isinstance(x, torch.nn.BatchNorm2d)]
    assert all(b.eps == 1e-3 and b.momentum == 0.03 for b in bn)


def test_no_cpu_fallback():
    from multiyolov5_amd._lib import MyoloError
    from multiyolov5_amd.models.common import Conv
    with pytest.raises(MyoloError):
        Conv(8, 8, 3)(torch.zeros(1, 8, 4, 4))


def test_dgrad_tap_math():
    """stride-2 dgrad by output parity covers every (tap, pixel) pair exactly once."""
    from multiyolov5_amd.engine import taps_dgrad, taps_fwd
    k, d, pad, s, H = 3, 1, 1, 2, 8
    Ho = (H + 2 * pad - d * (k - 1) - 1) // s + 1
    fwd = set()
    dy, _, _ = taps_fwd(k, d, pad)
    for oy in range(Ho):
        for kh in range(k):
            iy = oy * s + kh * d - pad
            if 0 <= iy < H:
                fwd.add((iy, oy, kh))
    bwd = set()
    for py in range(s):
        tdy, _, tw = taps_dgrad(k, d, pad, s, py, 0)
        taps = sorted(set((a, t // k) for a, t in zip(tdy, tw)))
        for q in range((H - py + s - 1) // s):
            for off, kh in taps:
                oy = q + off
                if 0 <= oy < Ho:
                    bwd.add((q * s + py, oy, kh))
    assert fwd == bwd


@pytest.mark.parametrize('training', [True, False], ids=['train', 'eval'])
@pytest.mark.parametrize('tag', list(TAGS))
def test_plan_dry_build_on_cpu(tag, training):
    """host logic: the whole forward(+backward) launch list of every head builds without a GPU (no launch is made)."""
    from multiyolov5_amd import runtime as R
    from multiyolov5_amd.models.yolo import Model
    from tests.util import CFG
    m = Model(os.path.join(CFG, TAGS[tag]))
    m.train(training)
    for dt in (torch.float16, torch.float32):
        h = R.PlanHolder(m, [torch.zeros(2, 3, 64, 128)], ('t', 0), dt, training)
        nf = sum(len(o.fwd_calls) for o in h.plan.ops)
        nb = sum(len(o.bwd_calls) for o in h.plan.ops)
        # (eval plans: each Bottleneck is ONE launch since round 5, myolo_conv_pair: 12 launches fewer for yolov5s)
        assert nf > (70 if training else 50) and (nb > nf if training else nb == 0)
        if not training:
            assert sum(c.name == 'myolo_conv_pair' for o in h.plan.ops for c in o.fwd_calls) >= 10


def test_letterbox_geometry_matches_restatement():
    """host logic of utils/datasets.letterbox_params (datasets.py:818-846) vs the oracle's pixel-moving letterbox"""
    import numpy as np
    from multiyolov5_amd.utils.datasets import letterbox_params
    from oracle import frame_ref
    for shape, ns, auto in (((1024, 2048), 2048, True), ((1000, 2048), 2048, True), ((37, 64), 64, True), ((64, 50), 64, False),
                            ((480, 640), 640, True), ((640, 640), (640, 640), False)):
        new_unpad, ratio, pad, (t, b, l, r) = letterbox_params(shape, ns, auto=auto, stride=32)
        assert tuple(new_unpad) == (shape[1], shape[0])
        img, rratio, rpad = frame_ref.letterbox(np.zeros(shape + (3,), np.uint8), ns, auto=auto, stride=32)
        assert ratio == rratio and tuple(pad) == tuple(rpad) and img.shape[:2] == (shape[0] + t + b, shape[1] + l + r)
