"""state_dict key/shape enumeration for the `*_seg.yaml` configs (TEST INFRASTRUCTURE ONLY).

Restates the channel arithmetic of parse_model (/root/reference/models/yolo.py:373-429) and the
constructors of models/common.py + the SegMask heads, so tests can build a synthetic state_dict
without importing either the reference or the product, and can check the product's key set.
"""
from collections import OrderedDict

import torch

from .model_ref import make_divisible


class _B:
    def __init__(self):
        self.sd = OrderedDict()

    def conv(self, p, c1, c2, k, bias=False):
        self.sd[p + '.weight'] = (c2, c1, k, k)
        if bias:
            self.sd[p + '.bias'] = (c2,)

    def bn(self, p, c):
        self.sd[p + '.weight'] = (c,)
        self.sd[p + '.bias'] = (c,)
        self.sd[p + '.running_mean'] = (c,)
        self.sd[p + '.running_var'] = (c,)
        self.sd[p + '.num_batches_tracked'] = ()

    def cb(self, p, c1, c2, k=1):            # `Conv` wrapper (common.py:34-46)
        self.conv(p + '.conv', c1, c2, k)
        self.bn(p + '.bn', c2)

    def bare(self, p, c1, c2):               # Sequential(Conv2d k3, BN, SiLU)
        self.conv(p + '.0', c1, c2, 3)
        self.bn(p + '.1', c2)

    def c3(self, p, c1, c2, n):
        c_ = int(c2 * 0.5)
        self.cb(p + '.cv1', c1, c_)
        self.cb(p + '.cv2', c1, c_)
        self.cb(p + '.cv3', 2 * c_, c2)
        for i in range(n):
            self.cb(f'{p}.m.{i}.cv1', c_, c_)
            self.cb(f'{p}.m.{i}.cv2', c_, c_, 3)

    def spp(self, p, c1, c2):
        self.cb(p + '.cv1', c1, c1 // 2)
        self.cb(p + '.cv2', c1 // 2 * 4, c2)

    def c3spp(self, p, c1, c2):
        c_ = int(c1 * 0.5)
        self.cb(p + '.cv1', c1, c_)
        self.cb(p + '.cv2', c1, c_)
        self.cb(p + '.cv3', c_ + int(c_ * 1.5), c2)
        self.spp(p + '.m', c_, int(c_ * 1.5))

    def rfb2(self, p, c1, c2, map_reduce, has_globel):
        ci = c1 // map_reduce
        self.cb(p + '.branch0.0', c1, ci)
        self.cb(p + '.branch0.1', ci, ci, 3)
        self.bare(p + '.branch1', ci, ci)
        self.bare(p + '.branch2', ci, ci)
        self.cb(p + '.branch3.0', c1, ci)
        if has_globel:
            self.cb(p + '.branch4.1', ci, ci)
        self.cb(p + '.ConvLinear', (5 if has_globel else 4) * ci, c2)

    def aspp(self, p, c1, c2, map_reduce, has_globel):
        hid = c1 // map_reduce
        self.cb(p + '.branch0.0', c1, hid)
        for i in (1, 2, 3):
            self.bare(f'{p}.branch{i}', c1, hid)
        if has_globel:
            self.cb(p + '.branch4.1', c1, hid)
        self.cb(p + '.ConvLinear', (5 if has_globel else 4) * hid, c2)

    def ffm(self, p, c1, c2, k):
        self.cb(p + '.convblk', c1, c2, k)
        self.conv(p + '.channel_attention.1', c2, c2, 1)
        self.conv(p + '.channel_attention.3', c2, c2, 1)


def state_shapes(cfg):
    """OrderedDict name -> shape tuple, in the reference's registration order is NOT guaranteed (use as a set)."""
    b = _B()
    gd, gw, nc, nseg = cfg['depth_multiple'], cfg['width_multiple'], cfg['nc'], cfg['n_segcls']
    na = len(cfg['anchors'][0]) // 2
    ch = []
    c_prev = 3
    for i, (f, n, m, args) in enumerate(cfg['backbone'] + cfg['head']):
        p = f'model.{i}'
        n = max(round(n * gd), 1) if n > 1 else n
        cin = (c_prev if f == -1 else ch[f]) if isinstance(f, int) else [c_prev if j == -1 else ch[j] for j in f]
        if m in ('Focus', 'Conv', 'C3', 'SPP'):
            c2 = make_divisible(args[0] * gw, 8)
            if m == 'Focus':
                b.cb(p + '.conv', cin * 4, c2, args[1])
            elif m == 'Conv':
                b.cb(p, cin, c2, args[1])
            elif m == 'C3':
                b.c3(p, cin, c2, n)
            else:
                b.spp(p, cin, c2)
        elif m == 'nn.Upsample':
            c2 = cin
        elif m == 'Concat':
            c2 = sum(cin)
        elif m == 'Detect':
            for j, c in enumerate(cin):
                b.conv(f'{p}.m.{j}', c, na * (nc + 5), 1, bias=True)
            b.sd[p + '.anchors'] = (len(cin), na, 2)
            b.sd[p + '.anchor_grid'] = (len(cin), 1, na, 1, 1, 2)
            c2 = None
        elif m.startswith('SegMask'):
            nn_ = max(round(args[1] * gd), 1) if args[1] > 1 else args[1]
            chid = make_divisible(args[2] * gw, 8)
            c2 = nseg
            if m == 'SegMaskPSP':
                b.rfb2(p + '.out.0', chid * 3, chid, 6, False)
                for j in (1, 2, 3, 4):
                    b.cb(f'{p}.out.1.conv{j}', chid, chid // 4)
                b.ffm(p + '.out.2', chid * 2, chid, 3)
                b.conv(p + '.out.3', chid, nseg, 1, bias=True)
                b.cb(p + '.m8.0', cin[0], chid)
                b.cb(p + '.m32.0', cin[2], chid)
                b.cb(p + '.m16.0', cin[1], chid)
            elif m == 'SegMaskBase':
                b.c3(p + '.m.0', cin[0], chid, nn_)
                b.c3spp(p + '.m.1', chid, int(chid * 1.5))
                b.conv(p + '.m.3', int(chid * 1.5), nseg, 3)
            elif m == 'SegMaskLab':
                b.cb(p + '.detail.0', cin[0], 48)
                b.cb(p + '.detail.1', 48, 48, 3)
                b.cb(p + '.encoder.0', cin[1], chid * 2)
                b.aspp(p + '.encoder.1', chid * 2, 256, 5 - nn_, False)
                b.ffm(p + '.decoder.0', 256 + 48, 256, 1)
                b.cb(p + '.decoder.1', 256, chid, 3)
                b.conv(p + '.decoder.2', chid, nseg, 1, bias=True)
            elif m == 'SegMaskBiSe':
                b.cb(p + '.m8.0', cin[0], 128)
                b.rfb2(p + '.m16.0', cin[1], 128, 4, False)
                b.rfb2(p + '.m32.0', cin[2], 128, 8, True)
                b.cb(p + '.up16.0', 128, 128, 3)
                b.cb(p + '.up32.0', 128, 128, 3)
                b.ffm(p + '.out.0', 256, 256, 3)
                b.conv(p + '.out.2', 256, nseg, 1, bias=True)
                b.cb(p + '.aux16.0', 128, 128, 3)
                b.conv(p + '.aux16.1', 128, nseg, 1, bias=True)
                b.cb(p + '.aux32.0', 128, 128, 3)
                b.conv(p + '.aux32.1', 128, nseg, 1, bias=True)
        else:
            raise NotImplementedError(m)
        ch.append(c2)
        c_prev = c2
    return b.sd


def anchors_buffers(cfg, strides=(8., 16., 32.)):
    """Detect buffers as the reference leaves them after Model.__init__ (yolo.py:200-202, 262):
    `anchors` in grid units, `anchor_grid` in pixels."""
    a = torch.tensor(cfg['anchors'], dtype=torch.float32).view(len(cfg['anchors']), -1, 2)
    return a / torch.tensor(strides).view(-1, 1, 1), a.clone().view(len(cfg['anchors']), 1, -1, 1, 1, 2)


def template_state_dict(cfg):
    sd = OrderedDict()
    for k, shp in state_shapes(cfg).items():
        sd[k] = torch.zeros(shp, dtype=torch.long if k.endswith('num_batches_tracked') else torch.float32)
    anc, ag = anchors_buffers(cfg)
    for k in sd:
        if k.endswith('.anchors'):
            sd[k] = anc
        elif k.endswith('.anchor_grid'):
            sd[k] = ag
    return sd
