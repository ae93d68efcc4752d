"""Block library of the multiyolov5 hot path on MI355X -- host-side mirror of the reference's `models/common.py`.

Same class names, constructor signatures, attribute names and `state_dict` keys as the reference
(/root/reference/models/common.py; file:line cited per class), so `parse_model`'s `eval(name)` and pickled
checkpoints resolve.  None of these modules calls ATen for compute: `forward()` builds (once per input
signature) and runs a static plan of libmyolo kernel launches (multiyolov5_amd/engine.py); the nn.Conv2d /
nn.BatchNorm2d members are parameter containers only.
"""
import os

import torch
import torch.nn as nn

from .. import _lib as L
from .. import engine as E
from ..runtime import PlannedModule

SILU = L.ACT_SILU


def autopad(k, p=None):  # common.py:22-26
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


def _act_code(act):
    if isinstance(act, nn.SiLU):
        return L.ACT_SILU
    if isinstance(act, nn.Sigmoid):
        return L.ACT_SIGMOID
    if isinstance(act, nn.Identity):
        return L.ACT_NONE
    raise NotImplementedError(f'activation {type(act).__name__} has no gfx950 epilogue (SiLU / Sigmoid / Identity only)')


def conv_out_hw(h, w, k, s, p, d):
    return (h + 2 * p - d * (k - 1) - 1) // s + 1, (w + 2 * p - d * (k - 1) - 1) // s + 1


def emit_conv(plan, x, conv, bn=None, act=L.ACT_NONE, res=None, out=None, det=None):
    """append one Conv2d(+BN)(+act)(+res) to the plan; `conv` is an nn.Conv2d used as a parameter container."""
    k, s, d = conv.kernel_size[0], conv.stride[0], conv.dilation[0]
    if conv.groups != 1 or conv.kernel_size[0] != conv.kernel_size[1] or conv.padding[0] != d * (k // 2):
        raise NotImplementedError('only groups=1, square, "same"-padded convolutions are on the hot path (SURVEY §2.2 K1)')
    ho, wo = conv_out_hw(x.h, x.w, k, s, conv.padding[0], d)
    if out is None:
        out = plan.new(x.n, ho, wo, conv.out_channels)
    op = plan.add(E.ConvOp(plan, x, out, conv.weight, bn=bn, bias=conv.bias, k=k, s=s, d=d, act=act, res=res, det=det))
    return out, op


C3_MERGE = os.environ.get('MYOLO_C3_MERGE', '1') != '0'


def emit_csp_entry(plan, x, cv1, cv2, c_inner, place_inner=True):
    """C3 / C3SPP entry (common.py:137,150): cv1(x) and cv2(x) are two 1x1 Conv+BN+SiLU of the SAME input -> ONE convolution with the
    stacked weights [cv2; cv1], one BatchNorm pass per direction with two parameter sets, one dgrad (K = 2c_): ~5 launches fewer per
    block on the backward's dependent chain.  One buffer [inner (c_inner) | cv2 (c_) | cv1 (c_)]: the stacked conv writes the last
    2c_ channels, the inner branch reads the cv1 slice and its result is placed in front, so cv3 reads channels [0, c_inner + c_) in
    the reference's concat order.  Returns (cv1 output slice, place(inner_result) -> cv3 input) -- or (cv1 output slice, buffer) with
    place_inner=False -- or None when the pair cannot merge."""
    a, b = cv1.conv, cv2.conv
    bn1, bn2 = (cv1.bn if hasattr(cv1, 'bn') else None), (cv2.bn if hasattr(cv2, 'bn') else None)
    seg = E.SEG[plan.dtype]
    same = (a.kernel_size, a.stride, a.padding, a.dilation, a.groups, a.bias is None) == \
           (b.kernel_size, b.stride, b.padding, b.dilation, b.groups, b.bias is None)
    if not (C3_MERGE and same and a.kernel_size == (1, 1) and a.stride == (1, 1) and a.groups == 1 and (bn1 is None) == (bn2 is None)
            and type(cv1.act) is type(cv2.act) and a.out_channels % seg == 0 and b.out_channels % seg == 0 and c_inner % seg == 0
            and (bn1 is None or (bn1.eps, bn1.momentum) == (bn2.eps, bn2.momentum))):
        return None
    c1o, c2o = a.out_channels, b.out_channels
    buf = plan.new_buf(x.n, x.h, x.w, c_inner + c2o + c1o)
    both = plan.new(x.n, x.h, x.w, c2o + c1o)
    both.place(buf, c_inner)
    plan.add(E.ConvOp(plan, x, both, b.weight, bn=bn2, bias=b.bias, k=1, s=1, d=1, act=_act_code(cv1.act),
                      weight2=a.weight, bn2=bn1, bias2=a.bias))
    first = plan.new(x.n, x.h, x.w, c1o)
    first.place(buf, c_inner + c2o)
    if not place_inner:                 # (RFB2: the caller lays several tensors out in front itself)
        return first, buf

    def place(inner):
        assert inner.buf is None and inner.c == c_inner and (inner.n, inner.h, inner.w) == (x.n, x.h, x.w)
        inner.place(buf, 0)
        cat = plan.new(x.n, x.h, x.w, c_inner + c2o)
        cat.place(buf, 0)
        return cat
    return first, place


class Conv(PlannedModule):
    """Conv2d(bias=False) + BatchNorm2d + SiLU (common.py:34-46); `fuseforward` after Model.fuse() (45-46)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p), groups=g, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = nn.SiLU() if act is True else (act if isinstance(act, nn.Module) else nn.Identity())

    def emit(self, plan, x, res=None):
        bn = self.bn if hasattr(self, 'bn') else None       # fused: bn deleted, conv carries the bias
        return emit_conv(plan, x, self.conv, bn, _act_code(self.act), res)[0]

    def fuseforward(self, x):
        return self.forward(x)


class Bottleneck(PlannedModule):
    """x + cv2(cv1(x)) (common.py:95-105); the add rides in cv2's epilogue."""

    def __init__(self, c1, c2, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_, c2, 3, 1, g=g)
        self.add = shortcut and c1 == c2

    def emit(self, plan, x):
        bn1 = self.cv1.bn if hasattr(self.cv1, 'bn') else None
        bn2 = self.cv2.bn if hasattr(self.cv2, 'bn') else None
        t, op1 = emit_conv(plan, x, self.cv1.conv, bn1, _act_code(self.cv1.act))
        y, op2 = emit_conv(plan, t, self.cv2.conv, bn2, _act_code(self.cv2.act), res=x if self.add else None)
        if not plan.training:
            op2.pair_first = op1          # cv1's output has this one reader: eval plans run both layers in one launch (myolo_conv_pair)
        return y


class C3(PlannedModule):
    """cv3(cat(m(cv1(x)), cv2(x))) (common.py:127-139); the cat is free (slice writes)."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*[Bottleneck(c_, c_, shortcut, g, e=1.0) for _ in range(n)])

    def emit(self, plan, x):
        merged = emit_csp_entry(plan, x, self.cv1, self.cv2, self.cv1.conv.out_channels) if len(self.m) else None
        if merged is not None:
            a, place = merged
            for b in self.m:
                a = b.emit(plan, a)
            return self.cv3.emit(plan, place(a))
        a = self.cv1.emit(plan, x)
        for b in self.m:
            a = b.emit(plan, a)
        return self.cv3.emit(plan, plan.cat([a, self.cv2.emit(plan, x)]))


class SPP(PlannedModule):
    """cv1 -> [x, maxpool5, maxpool9, maxpool13] -> cv2 (common.py:163-174); one kernel does the three pools."""

    def __init__(self, c1, c2, k=(5, 9, 13)):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * (len(k) + 1), c2, 1, 1)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=x, stride=1, padding=x // 2) for x in k])
        self.k = tuple(k)

    def emit(self, plan, x):
        if tuple(getattr(self, 'k', (5, 9, 13))) != (5, 9, 13):
            raise NotImplementedError('SPP kernel sizes other than (5,9,13)')
        x = self.cv1.emit(plan, x)
        outs = [plan.new(x.n, x.h, x.w, x.c) for _ in range(3)]
        plan.add(E.SppPoolOp(plan, x, outs))
        return self.cv2.emit(plan, plan.cat([x] + outs))


class C3SPP(PlannedModule):
    """C3 whose inner block is SPP (common.py:142-152)."""

    def __init__(self, c1, c2, k=(5, 9, 13), g=1, e=0.5):
        super().__init__()
        c_ = int(c1 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(c_ + int(c_ * 1.5), c2, 1)
        self.m = SPP(c_, int(c_ * 1.5), k=k)

    def emit(self, plan, x):
        merged = emit_csp_entry(plan, x, self.cv1, self.cv2, self.m.cv2.conv.out_channels)
        if merged is not None:
            a, place = merged
            return self.cv3.emit(plan, place(self.m.emit(plan, a)))
        a = self.m.emit(plan, self.cv1.emit(plan, x))
        return self.cv3.emit(plan, plan.cat([a, self.cv2.emit(plan, x)]))


class Focus(PlannedModule):
    """space-to-depth + Conv (common.py:542-551); the slicing/cat is fused with the image cast (focus_pack)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        self.conv = Conv(c1 * 4, c2, k, s, p, g, act)

    def emit(self, plan, x):
        if not isinstance(x, E.ImageInput):
            raise NotImplementedError('Focus consumes the raw NCHW image (c1=3) on this path')
        n, c, h, w = x.shape
        if c != 3 or h % 2 or w % 2:
            raise NotImplementedError('Focus: 3-channel image with even height/width expected')
        t = plan.new(n, h // 2, w // 2, 16, requires_grad=False)
        plan.add(E.FocusPackOp(plan, x.slot, t))
        return self.conv.emit(plan, t)


class Concat(PlannedModule):
    """torch.cat along channels (common.py:582-589) -- producers write into slices, no copy kernel."""

    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def emit(self, plan, xs):
        if self.d != 1:
            raise NotImplementedError('Concat along a non-channel dimension')
        return plan.cat(list(xs))


class Upsample(PlannedModule):
    """nn.Upsample(None, 2, 'nearest') of the PANet (yaml:31,36) / bilinear align_corners=True elsewhere."""

    def __init__(self, size=None, scale_factor=None, mode='nearest', align_corners=None):
        super().__init__()
        self.size, self.scale_factor, self.mode, self.align_corners = size, scale_factor, mode, align_corners

    def __reduce_ex__(self, protocol):
        """pickled / deep-copied as the torch.nn.Upsample the reference's yaml names (checkpoints written here must only name classes
        the reference has, train.py:485); models.yolo._emit_layer runs a plain nn.Upsample through this class again"""
        m = nn.Upsample(self.size, self.scale_factor, self.mode, self.align_corners)
        for k in ('i', 'f', 'type', 'np', 'training'):
            if k in self.__dict__:
                setattr(m, k, self.__dict__[k])
        return (nn.Upsample, (self.size, self.scale_factor, self.mode, self.align_corners), dict(m.__dict__))

    def emit(self, plan, x):
        s = int(self.scale_factor)
        out = plan.new(x.n, x.h * s, x.w * s, x.c)
        if self.mode == 'nearest':
            if s != 2:
                raise NotImplementedError('nearest upsample other than x2')
            plan.add(E.CopyUpOp(plan, x, out, 2))
        elif self.mode == 'bilinear' and self.align_corners:
            plan.add(E.BilinearOp(plan, x, out))
        else:
            raise NotImplementedError(f'Upsample mode {self.mode} align_corners={self.align_corners}')
        return out


def emit_bilinear(plan, x, h, w, group=None):
    out = plan.new(x.n, h, w, x.c)
    plan.add(E.BilinearOp(plan, x, out, group))
    return out


def emit_avgpool(plan, x, k, group=None):
    out = plan.new(x.n, k, k, x.c)
    plan.add(E.AvgPoolOp(plan, x, out, group))
    return out


def emit_broadcast(plan, g, h, w):
    """F.interpolate(1x1 -> HxW, 'nearest') == broadcast (common.py:273,509)."""
    out = plan.new(g.n, h, w, g.c)
    plan.add(E.BilinearOp(plan, g, out))     # 1x1 source: bilinear align_corners degenerates to a broadcast
    return out


def _bare_conv_bn_act(c1, c2, d):
    """nn.Sequential(Conv2d(k3,dilated,bias=False), BatchNorm2d, SiLU) exactly as written inline in RFB/ASPP (common.py:481-490):
    plain torch containers (so that a pickled model names only classes the reference also has); `_emit_bare` runs them."""
    return nn.Sequential(nn.Conv2d(c1, c2, kernel_size=3, stride=1, padding=d, dilation=d, bias=False),
                         nn.BatchNorm2d(c2), nn.SiLU())


def _emit_bare(plan, seq, x):
    return emit_conv(plan, x, seq[0], seq[1], _act_code(seq[2]))[0]


def _emit_seq(plan, seq, x):
    """nn.Sequential of `Conv` wrappers and inline Conv2d, BatchNorm2d, act triples"""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Conv2d):
            x = emit_conv(plan, x, m, mods[i + 1], _act_code(mods[i + 2]))[0]
            i += 3
        else:
            x = m.emit(plan, x)
            i += 1
    return x


def _global_branch(c1, c2):
    return nn.Sequential(nn.AdaptiveAvgPool2d(1), Conv(c1, c2, k=1))


def _emit_global(plan, seq, x):
    return seq[1].emit(plan, emit_avgpool(plan, x, 1))


class RFB2(PlannedModule):
    """cascaded dilated block (common.py:470-511)."""

    def __init__(self, in_planes, out_planes, map_reduce=4, d=[2, 3], has_globel=False):
        super().__init__()
        self.out_channels = out_planes
        self.has_globel = has_globel
        ip = in_planes // map_reduce
        self.branch0 = nn.Sequential(Conv(in_planes, ip, k=1, s=1), Conv(ip, ip, k=3, s=1))
        self.branch1 = _bare_conv_bn_act(ip, ip, d[0])
        self.branch2 = _bare_conv_bn_act(ip, ip, d[1])
        self.branch3 = nn.Sequential(Conv(in_planes, ip, k=1, s=1))
        if self.has_globel:
            self.branch4 = _global_branch(ip, ip)
        self.ConvLinear = Conv(int(5 * ip) if has_globel else int(4 * ip), out_planes, k=1, s=1)

    def emit(self, plan, x, res=None):
        # branch3 and branch0[0] are two 1x1 Conv+BN+SiLU of the same input: one launch (see emit_csp_entry); the buffer is the block's
        # concat [x0 | x1 | x2 | x3] with branch0's hidden map appended behind x3, the stacked conv writes [x3 | hidden]
        b3, b00 = self.branch3[0], self.branch0[0]
        merged = None if self.has_globel else emit_csp_entry(plan, x, b00, b3, 3 * b3.conv.out_channels, place_inner=False)
        if merged is not None:
            hidden, buf = merged
            ip = b3.conv.out_channels
            x0 = self.branch0[1].emit(plan, hidden)
            x1 = _emit_bare(plan, self.branch1, x0)
            x2 = _emit_bare(plan, self.branch2, x1)
            for i, t in enumerate((x0, x1, x2)):
                assert t.buf is None and t.c == ip
                t.place(buf, i * ip)
            cat = plan.new(x.n, x.h, x.w, 4 * ip)
            cat.place(buf, 0)
            return self.ConvLinear.emit(plan, cat)
        x3 = _emit_seq(plan, self.branch3, x)
        x0 = _emit_seq(plan, self.branch0, x)
        x1 = _emit_bare(plan, self.branch1, x0)
        x2 = _emit_bare(plan, self.branch2, x1)
        parts = [x0, x1, x2, x3]
        if self.has_globel:
            parts.append(emit_broadcast(plan, _emit_global(plan, self.branch4, x2), x.h, x.w))
        return self.ConvLinear.emit(plan, plan.cat(parts))


class RFB1(PlannedModule):
    """parallel-branch variant (common.py:416-466): the commented alternative encoder of SegMaskLab (yolo.py:110); branch3 carries
    the only 5x5 (25-tap) convolution of the block library."""

    def __init__(self, in_planes, out_planes, map_reduce=4, d=[3, 5, 7], has_globel=False):
        super().__init__()
        self.out_channels = out_planes
        self.has_globel = has_globel
        ip = in_planes // map_reduce
        self.branch0 = nn.Sequential(Conv(in_planes, ip, k=1, s=1), Conv(ip, ip, k=3, s=1))
        self.branch1 = nn.Sequential(Conv(in_planes, ip, k=1, s=1), Conv(ip, ip, k=3, s=1), *_bare_conv_bn_act(ip, ip, d[0]))
        self.branch2 = nn.Sequential(Conv(in_planes, ip, k=1, s=1), Conv(ip, ip, k=3, s=1), *_bare_conv_bn_act(ip, ip, d[1]))
        self.branch3 = nn.Sequential(Conv(in_planes, ip, k=1, s=1), Conv(ip, ip, k=5, s=1), *_bare_conv_bn_act(ip, ip, d[2]))
        if self.has_globel:
            self.branch4 = _global_branch(in_planes, ip)
        self.Fusion = Conv(int(5 * ip) if has_globel else int(4 * ip), out_planes, k=1, s=1)

    def emit(self, plan, x):
        parts = [_emit_seq(plan, b, x) for b in (self.branch0, self.branch1, self.branch2, self.branch3)]
        if self.has_globel:
            parts.append(emit_broadcast(plan, _emit_global(plan, self.branch4, x), x.h, x.w))
        return self.Fusion.emit(plan, plan.cat(parts))


class ASPP(PlannedModule):
    """1x1 || three dilated 3x3 (|| global) -> 1x1 (common.py:233-275)."""

    def __init__(self, in_planes, out_planes, d=[3, 6, 9], has_globel=True, map_reduce=4):
        super().__init__()
        self.has_globel = has_globel
        self.hid = in_planes // map_reduce
        self.branch0 = nn.Sequential(Conv(in_planes, self.hid, k=1, s=1))
        self.branch1 = _bare_conv_bn_act(in_planes, self.hid, d[0])
        self.branch2 = _bare_conv_bn_act(in_planes, self.hid, d[1])
        self.branch3 = _bare_conv_bn_act(in_planes, self.hid, d[2])
        if self.has_globel:
            self.branch4 = _global_branch(in_planes, self.hid)
        self.ConvLinear = Conv(int(5 * self.hid) if has_globel else int(4 * self.hid), out_planes, k=1, s=1)

    def emit(self, plan, x):
        parts = [_emit_seq(plan, b, x) for b in (self.branch0, self.branch1, self.branch2, self.branch3)]
        if self.has_globel:
            parts.append(emit_broadcast(plan, _emit_global(plan, self.branch4, x), x.h, x.w))
        return self.ConvLinear.emit(plan, plan.cat(parts))


def _conv_then_bare(c1, c2, d):
    """nn.Sequential(Conv(c1,hid,1), Conv2d(k3,dilated,bias=False), BatchNorm2d, SiLU) of ASPPs (common.py:288-305)"""
    return nn.Sequential(Conv(c1, c2, k=1), nn.Conv2d(c2, c2, kernel_size=3, stride=1, padding=d, dilation=d, bias=False),
                         nn.BatchNorm2d(c2), nn.SiLU())


class ASPPs(PlannedModule):
    """ASPP with a private 1x1 channel reduction in front of every branch (common.py:278-324)."""

    def __init__(self, in_planes, out_planes, d=[3, 6, 9], has_globel=True, map_reduce=4):
        super().__init__()
        self.has_globel = has_globel
        self.hid = in_planes // map_reduce
        self.branch0 = nn.Sequential(Conv(in_planes, self.hid, k=1), Conv(self.hid, self.hid, k=3, s=1))
        self.branch1 = _conv_then_bare(in_planes, self.hid, d[0])
        self.branch2 = _conv_then_bare(in_planes, self.hid, d[1])
        self.branch3 = _conv_then_bare(in_planes, self.hid, d[2])
        if self.has_globel:
            self.branch4 = _global_branch(in_planes, self.hid)
        self.ConvLinear = Conv(int(5 * self.hid) if has_globel else int(4 * self.hid), out_planes, k=1, s=1)

    def emit(self, plan, x):
        parts = [_emit_seq(plan, b, x) for b in (self.branch0, self.branch1, self.branch2, self.branch3)]
        if self.has_globel:
            parts.append(emit_broadcast(plan, _emit_global(plan, self.branch4, x), x.h, x.w))
        return self.ConvLinear.emit(plan, plan.cat(parts))


class PyramidPooling(PlannedModule):
    """AdaptiveAvgPool(1,2,3,6) -> 1x1 -> bilinear up -> cat with x (common.py:514-539)."""

    def __init__(self, in_channels, k=[1, 2, 3, 6]):
        super().__init__()
        self.pool1 = nn.AdaptiveAvgPool2d(k[0])
        self.pool2 = nn.AdaptiveAvgPool2d(k[1])
        self.pool3 = nn.AdaptiveAvgPool2d(k[2])
        self.pool4 = nn.AdaptiveAvgPool2d(k[3])
        oc = in_channels // 4
        self.conv1 = Conv(in_channels, oc, k=1)
        self.conv2 = Conv(in_channels, oc, k=1)
        self.conv3 = Conv(in_channels, oc, k=1)
        self.conv4 = Conv(in_channels, oc, k=1)
        self.k = list(k)

    def emit(self, plan, x):
        feats = [x]
        group, ups = [], []                                   # the four pools / the four upsamples share one backward pass each (engine)
        for k, conv in zip(self.k, (self.conv1, self.conv2, self.conv3, self.conv4)):
            feats.append(emit_bilinear(plan, conv.emit(plan, emit_avgpool(plan, x, k, group)), x.h, x.w, ups))
        return plan.cat(feats)


class FFM(PlannedModule):
    """feature fusion: f = Conv_k(cat); out = f*sigmoid(W2 silu(W1 GAP(f))) + f (common.py:210-230)."""

    def __init__(self, in_chan, out_chan, reduction=1, is_cat=True, k=1):
        super().__init__()
        self.convblk = Conv(in_chan, out_chan, k=k, s=1, p=None)
        self.channel_attention = nn.Sequential(
            nn.AdaptiveAvgPool2d(1),
            nn.Conv2d(out_chan, out_chan // reduction, kernel_size=1, stride=1, padding=0, bias=False),
            nn.SiLU(inplace=True),
            nn.Conv2d(out_chan // reduction, out_chan, kernel_size=1, stride=1, padding=0, bias=False),
            nn.Sigmoid())
        self.is_cat = is_cat

    def emit(self, plan, xs):
        x = plan.cat(list(xs)) if self.is_cat else xs
        feat = self.convblk.emit(plan, x)
        a = emit_avgpool(plan, feat, 1)
        ca = self.channel_attention
        a = emit_conv(plan, a, ca[1], None, L.ACT_SILU)[0]
        a = emit_conv(plan, a, ca[3], None, L.ACT_SIGMOID)[0]
        out = plan.new(feat.n, feat.h, feat.w, feat.c)
        plan.add(E.GateOp(plan, feat, a, out))
        return out


def _emit_se_bn(plan, pooled, conv_mod, act):
    """`Conv(c, c, k=1, act=False)` (+ the nn.Sigmoid that follows it) on the pooled [n,1,1,c] vector (common.py:183,188,200):
    the sigmoid rides in the BatchNorm kernel's activation slot (train) / the conv epilogue (fused eval)."""
    bn = conv_mod.bn if hasattr(conv_mod, 'bn') else None
    return emit_conv(plan, pooled, conv_mod.conv, bn, act)[0]


class Attention(PlannedModule):
    """SE gate x * W(x) (common.py:177-192)."""

    def __init__(self, chan, reduction=1):
        super().__init__()
        if reduction > 1:
            self.W = nn.Sequential(nn.AdaptiveAvgPool2d(1), Conv(chan, chan // reduction, k=1, s=1),
                                   Conv(chan // reduction, chan, k=1, s=1, act=False), nn.Sigmoid())
        else:
            self.W = nn.Sequential(nn.AdaptiveAvgPool2d(1), Conv(chan, chan, k=1, s=1, act=False), nn.Sigmoid())

    def emit(self, plan, x):
        a = emit_avgpool(plan, x, 1)
        convs = [m for m in self.W if isinstance(m, Conv)]
        for m in convs[:-1]:
            a = m.emit(plan, a)
        a = _emit_se_bn(plan, a, convs[-1], L.ACT_SIGMOID)
        out = plan.new(x.n, x.h, x.w, x.c)
        plan.add(E.GateOp(plan, x, a, out, residual=False))
        return out


class ARM(PlannedModule):
    """AttentionRefinementModule: feat = Conv3x3(x); feat * sigmoid(BN(W GAP(feat))) (common.py:195-207)."""

    def __init__(self, in_chan, out_chan, *args, **kwargs):
        super().__init__()
        self.conv = Conv(in_chan, out_chan, k=3, s=1, p=None)
        self.channel_attention = nn.Sequential(nn.AdaptiveAvgPool2d(1), Conv(out_chan, out_chan, k=1, s=1, act=False),
                                               nn.Sigmoid())

    def emit(self, plan, x):
        feat = self.conv.emit(plan, x)
        a = _emit_se_bn(plan, emit_avgpool(plan, feat, 1), self.channel_attention[1], L.ACT_SIGMOID)
        out = plan.new(feat.n, feat.h, feat.w, feat.c)
        plan.add(E.GateOp(plan, feat, a, out, residual=False))
        return out
