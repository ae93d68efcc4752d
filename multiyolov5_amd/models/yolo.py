"""Model graph, Detect and the four segmentation heads on MI355X -- host-side mirror of the reference's
`models/yolo.py` (Model 233-370, parse_model 373-429, Detect 189-230, SegMask{BiSe,Lab,Base,PSP} 30-186).

Same public surface (class names, ctor signatures, attributes `.model .save .stride .names .yaml .nc .hyp .gr`,
`forward(x, augment=False, profile=False)`, `.fuse()`, `state_dict` keys); the forward/backward is one libmyolo
launch plan per input signature instead of ~350 ATen calls.
"""
import logging
import math
from copy import deepcopy
from pathlib import Path

import torch
import torch.nn as nn

from .. import _lib as L
from .. import engine as E
from ..runtime import PlannedModule
from ..utils.general import make_divisible
from ..utils.torch_utils import fuse_conv_and_bn, initialize_weights, model_info
from .common import *  # noqa: F401,F403  (parse_model resolves block names by eval, as yolo.py:381 does)
from .common import (ASPP, C3, C3SPP, FFM, RFB2, Concat, Conv, Focus, PyramidPooling, SPP, Upsample, emit_conv)

logger = logging.getLogger(__name__)


def _classifier(plan, x, conv):
    """Conv2d(c, n_segcls) (+bias) producing the low-res class logits (yolo.py:66,117,141,162)."""
    return emit_conv(plan, x, conv, None, L.ACT_NONE)[0]


def _up(plan, x, scale):
    out = plan.new(x.n, x.h * scale, x.w * scale, x.c)
    plan.add(E.BilinearOp(plan, x, out))
    return out


def _emit_dropout(plan, m, x):
    if plan.training and m.p > 0:
        out = plan.new(x.n, x.h, x.w, x.c)
        plan.add(E.DropoutOp(plan, x, out, float(m.p)))
        return out
    return x


class SegMaskPSP(PlannedModule):
    """yolo.py:149-186: m8/m16/m32 1x1 -> cat 3*c_hid -> RFB2 -> PyramidPooling -> FFM(k3) -> 1x1 cls -> x8."""

    def __init__(self, n_segcls=19, n=1, c_hid=256, shortcut=False, ch=()):
        super().__init__()
        self.c_in8, self.c_in16, self.c_in32 = ch[0], ch[1], ch[2]
        self.c_out = n_segcls
        self.out = nn.Sequential(RFB2(c_hid * 3, c_hid, d=[2, 3], map_reduce=6),
                                 PyramidPooling(c_hid, k=[1, 2, 3, 6]),
                                 FFM(c_hid * 2, c_hid, k=3, is_cat=False),
                                 nn.Conv2d(c_hid, self.c_out, kernel_size=1, padding=0),
                                 nn.Upsample(scale_factor=8, mode='bilinear', align_corners=True))
        self.m8 = nn.Sequential(Conv(self.c_in8, c_hid, k=1))
        self.m32 = nn.Sequential(Conv(self.c_in32, c_hid, k=1),
                                 nn.Upsample(scale_factor=4, mode='bilinear', align_corners=True))
        self.m16 = nn.Sequential(Conv(self.c_in16, c_hid, k=1),
                                 nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True))

    def emit(self, plan, xs):
        f8 = self.m8[0].emit(plan, xs[0])
        f16 = _up(plan, self.m16[0].emit(plan, xs[1]), 2)
        f32 = _up(plan, self.m32[0].emit(plan, xs[2]), 4)
        y = self.out[0].emit(plan, plan.cat([f8, f16, f32]))
        y = self.out[1].emit(plan, y)
        y = self.out[2].emit(plan, y)
        return E.SegHandle(_classifier(plan, y, self.out[3]), 8)


class SegMaskBase(PlannedModule):
    """yolo.py:129-146: C3 -> C3SPP -> Dropout -> Conv2d k3 (no bias) -> x8."""

    def __init__(self, n_segcls=19, n=1, c_hid=256, shortcut=False, ch=()):
        super().__init__()
        self.c_in = ch[0]
        self.c_out = n_segcls
        self.m = nn.Sequential(C3(c1=self.c_in, c2=c_hid, n=n, shortcut=shortcut, g=1, e=0.5),
                               C3SPP(c1=c_hid, c2=int(c_hid * 1.5), k=(5, 9, 13), g=1, e=0.5),
                               nn.Dropout(0.1, True),
                               nn.Conv2d(int(c_hid * 1.5), self.c_out, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1),
                                         groups=1, bias=False),
                               nn.Upsample(scale_factor=8, mode='bilinear', align_corners=True))

    def emit(self, plan, xs):
        y = self.m[1].emit(plan, self.m[0].emit(plan, xs[0]))
        y = _emit_dropout(plan, self.m[2], y)
        return E.SegHandle(_classifier(plan, y, self.m[3]), 8)


class SegMaskLab(PlannedModule):
    """yolo.py:93-124: detail (1/8) || ASPP encoder (1/16, x2 up) -> FFM(k1) -> Conv k3 -> 1x1 cls -> x8."""

    def __init__(self, n_segcls=19, n=1, c_hid=256, shortcut=False, ch=()):
        super().__init__()
        self.c_detail, self.c_in16 = ch[0], ch[1]
        self.c_out = n_segcls
        self.detail = nn.Sequential(Conv(self.c_detail, 48, k=1), Conv(48, 48, k=3))
        self.encoder = nn.Sequential(Conv(self.c_in16, c_hid * 2, k=1),
                                     ASPP(c_hid * 2, 256, d=[3, 6, 9], has_globel=False, map_reduce=5 - n),
                                     nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True))
        self.decoder = nn.Sequential(FFM(256 + 48, 256, k=1, is_cat=True),
                                     Conv(256, c_hid, k=3),
                                     nn.Conv2d(c_hid, self.c_out, kernel_size=1, padding=0),
                                     nn.Upsample(scale_factor=8, mode='bilinear', align_corners=True))

    def emit(self, plan, xs):
        f16 = _up(plan, self.encoder[1].emit(plan, self.encoder[0].emit(plan, xs[1])), 2)
        f8 = self.detail[1].emit(plan, self.detail[0].emit(plan, xs[0]))
        y = self.decoder[1].emit(plan, self.decoder[0].emit(plan, [f8, f16]))
        return E.SegHandle(_classifier(plan, y, self.decoder[2]), 8)


class SegMaskBiSe(PlannedModule):
    """yolo.py:30-86: RFB2 on 1/32 (+global) and 1/16, refine+up, FFM(k3) -> cls -> x8; train mode adds aux16/aux32."""

    def __init__(self, n_segcls=19, n=1, c_hid=256, shortcut=False, ch=()):
        super().__init__()
        self.c_in8, self.c_in16, self.c_in32 = ch[0], ch[1], ch[2]
        self.c_out = n_segcls
        self.m8 = nn.Sequential(Conv(self.c_in8, 128, k=1, s=1))
        self.m16 = nn.Sequential(RFB2(self.c_in16, 128, map_reduce=4, d=[2, 3], has_globel=False))
        self.m32 = nn.Sequential(RFB2(self.c_in32, 128, map_reduce=8, d=[2, 3], has_globel=True))
        self.up16 = nn.Sequential(Conv(128, 128, 3), nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True))
        self.up32 = nn.Sequential(Conv(128, 128, 3), nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True))
        self.out = nn.Sequential(FFM(256, 256, k=3), nn.Dropout(0.1),
                                 nn.Conv2d(256, self.c_out, kernel_size=1, padding=0),
                                 nn.Upsample(scale_factor=8, mode='bilinear', align_corners=True))
        self.aux16 = nn.Sequential(Conv(128, 128, 3), nn.Conv2d(128, self.c_out, kernel_size=1),
                                   nn.Upsample(scale_factor=8, mode='bilinear', align_corners=True))
        self.aux32 = nn.Sequential(Conv(128, 128, 3), nn.Conv2d(128, self.c_out, kernel_size=1),
                                   nn.Upsample(scale_factor=16, mode='bilinear', align_corners=True))

    def emit(self, plan, xs):
        feat3 = _up(plan, self.up32[0].emit(plan, self.m32[0].emit(plan, xs[2])), 2)
        m16 = self.m16[0].emit(plan, xs[1])
        s = plan.new(m16.n, m16.h, m16.w, m16.c)
        plan.add(E.AddOp(plan, m16, feat3, s))
        feat2 = _up(plan, self.up16[0].emit(plan, s), 2)
        y = self.out[0].emit(plan, [self.m8[0].emit(plan, xs[0]), feat2])
        y = _emit_dropout(plan, self.out[1], y)
        main = E.SegHandle(_classifier(plan, y, self.out[2]), 8)
        if not plan.training:
            return main
        a16 = E.SegHandle(_classifier(plan, self.aux16[0].emit(plan, feat2), self.aux16[1]), 8)
        a32 = E.SegHandle(_classifier(plan, self.aux32[0].emit(plan, feat3), self.aux32[1]), 16)
        return [main, a16, a32]


class Detect(PlannedModule):
    """yolo.py:189-230: per level 1x1 conv(+bias) written straight in [N,na,ny,nx,no]; eval adds the box decode."""
    stride = None
    export = False

    def __init__(self, nc=80, anchors=(), ch=()):
        super().__init__()
        self.nc = nc
        self.no = nc + 5
        self.nl = len(anchors)
        self.na = len(anchors[0]) // 2
        self.grid = [torch.zeros(1)] * self.nl
        a = torch.tensor(anchors).float().view(self.nl, -1, 2)
        self.register_buffer('anchors', a)
        self.register_buffer('anchor_grid', a.clone().view(self.nl, 1, -1, 1, 1, 2))
        self.m = nn.ModuleList(nn.Conv2d(x, self.no * self.na, 1) for x in ch)

    def emit(self, plan, xs):
        dets = []
        for i in range(self.nl):
            _, op = emit_conv(plan, xs[i], self.m[i], None, L.ACT_NONE, det=(self.na, self.no))
            dets.append(E.DetHandle(op))
        if plan.training or self.export:
            return dets
        return E.DecodeHandle(dets, self)


_SEG_HEADS = (SegMaskBiSe, SegMaskLab, SegMaskBase, SegMaskPSP)


def _emit_layer(plan, m, x):
    if hasattr(m, 'emit'):
        return m.emit(plan, x)
    if isinstance(m, nn.Upsample):                         # plain torch module unpickled from a reference checkpoint
        return Upsample(m.size, m.scale_factor, m.mode, m.align_corners).emit(plan, x)
    if isinstance(m, nn.Sequential):
        for s in m:
            x = _emit_layer(plan, s, x)
        return x
    raise NotImplementedError(f'{type(m).__name__} is not on the gfx950 hot path')


class Model(PlannedModule):
    def __init__(self, cfg='yolov5s.yaml', ch=3, nc=None, anchors=None):   # yolo.py:234
        super().__init__()
        if isinstance(cfg, dict):
            self.yaml = cfg
        else:
            import yaml
            self.yaml_file = Path(cfg).name
            with open(cfg) as f:
                self.yaml = yaml.load(f, Loader=yaml.SafeLoader)
        ch = self.yaml['ch'] = self.yaml.get('ch', ch)
        if nc and nc != self.yaml['nc']:
            logger.info(f"Overriding model.yaml nc={self.yaml['nc']} with nc={nc}")
            self.yaml['nc'] = nc
        if anchors:
            logger.info(f'Overriding model.yaml anchors with anchors={anchors}')
            self.yaml['anchors'] = round(anchors)
        self.model, self.save = parse_model(deepcopy(self.yaml), ch=[ch])
        self.save.append(24)                                   # yolo.py:253 (segmentation layer index is hard-coded)
        self.names = [str(i) for i in range(self.yaml['nc'])]
        m = self.model[-1]
        if isinstance(m, Detect):
            # yolo.py:260-265 runs a train-mode forward on zeros(2,ch,256,256) on the CPU to read the strides.  Strides
            # are a property of the graph: take them from a shape-only plan; reproduce the probe's side effect on the
            # BatchNorm buffers (running_var = 0.9*1 + 0.1*var(0) with the ctor-default momentum 0.1, one tracked batch).
            s = 256
            m.stride = torch.tensor([s / d for d in self._probe_det_sizes(ch, s)])
            for mod in self.modules():
                if isinstance(mod, nn.BatchNorm2d):
                    mod.running_var.mul_(0.9)
                    mod.num_batches_tracked.add_(1)
            m.anchors /= m.stride.view(-1, 1, 1)
            _check_anchor_order(m)
            self.stride = m.stride
            self._initialize_biases()
        initialize_weights(self)
        self.info()

    def _probe_det_sizes(self, ch, s):
        plan = E.Plan(torch.device('cpu'), torch.float32, True)
        dets, _ = self._emit_graph(plan, E.ImageInput(0, (2, ch, s, s)))
        return [d.op.out.h for d in dets]

    # ---- graph ------------------------------------------------------------------------------------------
    def _emit_graph(self, plan, x):
        """forward_once (yolo.py:293-316) as plan ops.  Eval plans issue the segmentation head (the layer before Detect, yolo.py:253;
        read by nobody but the model output) as soon as its inputs exist and tag its ops for the plan's side stream: it then runs
        beside the rest of the neck and Detect instead of behind them (detect.py frames are batch 1: most launches of both chains
        fill a fraction of the chip).  Training plans keep the reference's order."""
        n = len(self.model)
        order, seg_i = list(range(n)), n - 2
        f = self.model[seg_i].f if n >= 3 else None
        branch = (not plan.training and E.EVAL_BRANCH) and isinstance(f, (list, tuple)) and all(0 <= j < seg_i for j in f) and \
            not any(seg_i in [(g if g >= 0 else i + g) for g in (m.f if isinstance(m.f, (list, tuple)) else [m.f])]      # relative `from`
                    for i, m in enumerate(self.model[seg_i + 1:], seg_i + 1))                                           # indices resolved
        if branch:
            hoist = max(f)
            order = list(range(hoist + 1)) + [seg_i] + [i for i in range(hoist + 1, n) if i != seg_i]
        outs = {-1: x}
        for i in order:
            m = self.model[i]
            at = lambda j: outs[i + j] if j < 0 else outs[j]   # noqa: E731 -- negative `from` indices are relative to the layer (yolo.py:296)
            xin = at(m.f) if isinstance(m.f, int) else [at(j) for j in m.f]
            plan.cur_branch = 'seg' if (branch and i == seg_i) else None
            outs[i] = _emit_layer(plan, m, xin)
        plan.cur_branch = None
        return outs[n - 1], outs[n - 2]

    def emit(self, plan, x):
        det, seg = self._emit_graph(plan, x)
        return [det, seg]

    def forward(self, x, augment=False, profile=False):        # yolo.py:273-291
        if augment:
            raise NotImplementedError('test-time augmentation (yolo.py:274-289) is outside the hot path (SURVEY §8f rank 4)')
        return super().forward(x)

    def forward_once(self, x, profile=False):
        return super().forward(x)

    # ---- reference helpers --------------------------------------------------------------------------------
    def _initialize_biases(self, cf=None):                     # yolo.py:318-326
        m = self.model[-1]
        for mi, s in zip(m.m, m.stride):
            b = mi.bias.view(m.na, -1)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)
            b.data[:, 5:] += math.log(0.6 / (m.nc - 0.99)) if cf is None else torch.log(cf / cf.sum())
            mi.bias = torch.nn.Parameter(b.view(-1), requires_grad=True)

    def fuse(self):                                            # yolo.py:339-347
        print('Fusing layers... ')
        for m in self.model.modules():
            if type(m) is Conv and hasattr(m, 'bn'):
                m.conv = fuse_conv_and_bn(m.conv, m.bn)
                delattr(m, 'bn')
                m.forward = m.fuseforward
        self.invalidate_plans()
        self.info()
        return self

    def info(self, verbose=False, img_size=640):
        model_info(self, verbose, img_size)


def _check_anchor_order(m):                                    # utils/autoanchor.py:12-20
    a = m.anchor_grid.prod(-1).view(-1)
    da = a[-1] - a[0]
    ds = m.stride[-1] - m.stride[0]
    if da.sign() != ds.sign():
        print('Reversing anchor order')
        m.anchors[:] = m.anchors.flip(0)
        m.anchor_grid[:] = m.anchor_grid.flip(0)


def parse_model(d, ch):                                        # yolo.py:373-429
    logger.info('\n%3s%18s%3s%10s  %-40s%-30s' % ('', 'from', 'n', 'params', 'module', 'arguments'))
    anchors, nc, gd, gw, n_segcls = d['anchors'], d['nc'], d['depth_multiple'], d['width_multiple'], d['n_segcls']
    na = (len(anchors[0]) // 2) if isinstance(anchors, list) else anchors
    no = na * (nc + 5)
    layers, save, c2 = [], [], ch[-1]
    scope = dict(globals())
    scope['nn'] = type('nn_ns', (), {'Upsample': Upsample, 'BatchNorm2d': nn.BatchNorm2d})   # nn.Upsample -> planned variant
    for i, (f, n, m, args) in enumerate(d['backbone'] + d['head']):
        m = eval(m, scope) if isinstance(m, str) else m
        for j, a in enumerate(args):
            try:
                args[j] = eval(a, scope, {'nc': nc, 'anchors': anchors, 'n_segcls': n_segcls}) if isinstance(a, str) else a
            except Exception:
                pass
        n = max(round(n * gd), 1) if n > 1 else n
        if m in [Conv, SPP, Focus, C3, ASPP]:
            c1, c2 = ch[f], args[0]
            if c2 != no:
                c2 = make_divisible(c2 * gw, 8)
            args = [c1, c2, *args[1:]]
            if m is C3:
                args.insert(2, n)
                n = 1
        elif m is Concat:
            c2 = sum([ch[x] for x in f])
        elif m is Detect:
            args.append([ch[x] for x in f])
            if isinstance(args[1], int):
                args[1] = [list(range(args[1] * 2))] * len(f)
        elif m in _SEG_HEADS:
            args[1] = max(round(args[1] * gd), 1) if args[1] > 1 else args[1]
            args[2] = make_divisible(args[2] * gw, 8)
            args.append([ch[x] for x in f])
        else:
            c2 = ch[f]
        m_ = nn.Sequential(*[m(*args) for _ in range(n)]) if n > 1 else m(*args)
        t = str(m)[8:-2].replace('__main__.', '')
        np_ = sum([x.numel() for x in m_.parameters()])
        m_.i, m_.f, m_.type, m_.np = i, f, t, np_
        logger.info('%3s%18s%3s%10.0f  %-40s%-30s' % (i, f, n, np_, t, args))
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(m_)
        if i == 0:
            ch = []
        ch.append(c2)
    return nn.Sequential(*layers), sorted(save)
