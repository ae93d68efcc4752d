"""CPU fp32 restatement of the reference's model graph (TEST INFRASTRUCTURE ONLY).

A functional PyTorch-CPU re-expression of what `Model.forward_once`
(/root/reference/models/yolo.py:293-316) computes for the `*_seg.yaml` configs,
driven by a plain `state_dict` (same keys as the reference: `model.{i}.…`) and the
yaml dict.  No nn.Module is built; every block is a function of (ctx, prefix, x).

Each function cites the reference lines it follows.  Pinned against the reference
itself by `oracle/make_golden.py` -> `tests/golden/*.npz` (see oracle/__init__.py).
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3        # utils/torch_utils.py:150
BN_MOMENTUM = 0.03   # utils/torch_utils.py:151


def make_divisible(x, d):  # utils/general.py:176-178
    return math.ceil(x / d) * d


class Ctx:
    """state_dict + mode.  In training mode BN running stats in `sd` are updated in place."""

    def __init__(self, sd, training, dropout_p=0.1, record=None):
        self.sd = sd
        self.training = training
        self.dropout_p = dropout_p   # Base/BiSe heads (yolo.py:65,140); tests pin with 0.0
        self.record = record         # optional dict: name -> tensor (per-layer taps)
        self.dropout_fn = None       # tests: replay a given keep-mask instead of drawing one (dropout parity is statistical)
        self.maxpool_fn = None       # tests: fn(x, k) replacing F.max_pool2d(x, k, 1, k // 2) -- replays the arg-max choices another run made
                                     # at windows whose top-2 values are a rounding-noise tie (tests/gpu_util.pool_replay)

    def has(self, key):
        return key in self.sd


def _bn(ctx, p, x):
    """nn.BatchNorm2d with eps 1e-3 / momentum 0.03 (torch_utils.py:145-154)."""
    sd = ctx.sd
    if ctx.training:
        sd[p + '.num_batches_tracked'] += 1
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        ctx.training, BN_MOMENTUM, BN_EPS)


def conv_block(ctx, p, x, k=1, s=1, d=1, act=True):
    """`Conv` = Conv2d(bias=False, pad=k//2) + BN + SiLU (common.py:34-46); after `fuse()` the bn keys are gone
    and the conv carries a bias (common.py:45-46, torch_utils.py:182-202)."""
    sd = ctx.sd
    pad = d * (k // 2)
    y = F.conv2d(x, sd[p + '.conv.weight'], sd.get(p + '.conv.bias'), s, pad, d)
    if ctx.has(p + '.bn.weight'):
        y = _bn(ctx, p + '.bn', y)
    return F.silu(y) if act else y


def bare_conv_bn_silu(ctx, p, x, d):
    """nn.Sequential(Conv2d(k3, pad=d, dil=d, bias=False), BatchNorm2d, SiLU) (common.py:481-490, 242-256)."""
    y = F.conv2d(x, ctx.sd[p + '.0.weight'], None, 1, d, d)
    return F.silu(_bn(ctx, p + '.1', y))


def bottleneck(ctx, p, x, shortcut):  # common.py:95-105
    y = conv_block(ctx, p + '.cv2', conv_block(ctx, p + '.cv1', x, 1), 3)
    return x + y if shortcut else y


def c3(ctx, p, x, n, shortcut):  # common.py:127-139
    a = conv_block(ctx, p + '.cv1', x)
    for i in range(n):
        a = bottleneck(ctx, f'{p}.m.{i}', a, shortcut)
    b = conv_block(ctx, p + '.cv2', x)
    return conv_block(ctx, p + '.cv3', torch.cat((a, b), 1))


def spp(ctx, p, x, ks=(5, 9, 13)):  # common.py:163-174
    x = conv_block(ctx, p + '.cv1', x)
    pool = ctx.maxpool_fn if ctx.maxpool_fn is not None else (lambda t, k: F.max_pool2d(t, k, 1, k // 2))
    return conv_block(ctx, p + '.cv2', torch.cat([x] + [pool(x, k) for k in ks], 1))


def c3spp(ctx, p, x):  # common.py:142-152
    a = spp(ctx, p + '.m', conv_block(ctx, p + '.cv1', x))
    b = conv_block(ctx, p + '.cv2', x)
    return conv_block(ctx, p + '.cv3', torch.cat((a, b), 1))


def focus(ctx, p, x):  # common.py:542-551 -- parity order (row,col) = (0,0),(1,0),(0,1),(1,1)
    x = torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1)
    return conv_block(ctx, p + '.conv', x, 3)


def up_bilinear(x, scale=None, size=None):  # every bilinear op uses align_corners=True
    return F.interpolate(x, size=size, scale_factor=scale, mode='bilinear', align_corners=True)


def rfb2(ctx, p, x, d=(2, 3), has_globel=False):  # common.py:470-511
    x3 = conv_block(ctx, p + '.branch3.0', x)
    x0 = conv_block(ctx, p + '.branch0.1', conv_block(ctx, p + '.branch0.0', x), 3)
    x1 = bare_conv_bn_silu(ctx, p + '.branch1', x0, d[0])
    x2 = bare_conv_bn_silu(ctx, p + '.branch2', x1, d[1])
    parts = [x0, x1, x2, x3]
    if has_globel:
        g = conv_block(ctx, p + '.branch4.1', F.adaptive_avg_pool2d(x2, 1))
        parts.append(g.expand(-1, -1, x.shape[2], x.shape[3]))   # nearest 1x1 -> HxW == broadcast
    return conv_block(ctx, p + '.ConvLinear', torch.cat(parts, 1))


def rfb1(ctx, p, x, d=(3, 5, 7), has_globel=False):  # common.py:416-466 (parallel branches; branch3 has the 5x5 Conv)
    def branch(q, k2, dil):
        y = conv_block(ctx, q + '.1', conv_block(ctx, q + '.0', x), k2)
        if dil is None:
            return y
        y = F.conv2d(y, ctx.sd[q + '.2.weight'], None, 1, dil, dil)          # inline Conv2d, BatchNorm2d, SiLU (430-432)
        return F.silu(_bn(ctx, q + '.3', y))
    parts = [branch(p + '.branch0', 3, None), branch(p + '.branch1', 3, d[0]), branch(p + '.branch2', 3, d[1]),
             branch(p + '.branch3', 5, d[2])]
    if has_globel:
        g = conv_block(ctx, p + '.branch4.1', F.adaptive_avg_pool2d(x, 1))
        parts.append(g.expand(-1, -1, x.shape[2], x.shape[3]))               # F.interpolate(1x1 -> HxW, 'nearest') (464)
    return conv_block(ctx, p + '.Fusion', torch.cat(parts, 1))


def attention(ctx, p, x, reduction=1):  # common.py:177-192: x * sigmoid(Conv(act=False)(... GAP(x)))
    a = F.adaptive_avg_pool2d(x, 1)
    if reduction > 1:
        a = conv_block(ctx, p + '.W.1', a)
        a = conv_block(ctx, p + '.W.2', a, act=False)
    else:
        a = conv_block(ctx, p + '.W.1', a, act=False)
    return x * torch.sigmoid(a)


def arm(ctx, p, x):  # common.py:195-207
    feat = conv_block(ctx, p + '.conv', x, 3)
    a = conv_block(ctx, p + '.channel_attention.1', F.adaptive_avg_pool2d(feat, 1), act=False)
    return torch.mul(feat, torch.sigmoid(a))


def aspp(ctx, p, x, d=(3, 6, 9), has_globel=True):  # common.py:233-275
    parts = [conv_block(ctx, p + '.branch0.0', x)]
    for i in range(3):
        parts.append(bare_conv_bn_silu(ctx, f'{p}.branch{i + 1}', x, d[i]))
    if has_globel:
        g = conv_block(ctx, p + '.branch4.1', F.adaptive_avg_pool2d(x, 1))
        parts.append(g.expand(-1, -1, x.shape[2], x.shape[3]))
    return conv_block(ctx, p + '.ConvLinear', torch.cat(parts, 1))


def aspps(ctx, p, x, d=(3, 6, 9), has_globel=True):  # common.py:278-324
    parts = [conv_block(ctx, p + '.branch0.1', conv_block(ctx, p + '.branch0.0', x), 3)]
    for i in range(3):
        h = conv_block(ctx, f'{p}.branch{i + 1}.0', x)
        y = F.conv2d(h, ctx.sd[f'{p}.branch{i + 1}.1.weight'], None, 1, d[i], d[i])
        parts.append(F.silu(_bn(ctx, f'{p}.branch{i + 1}.2', y)))
    if has_globel:
        g = conv_block(ctx, p + '.branch4.1', F.adaptive_avg_pool2d(x, 1))
        parts.append(g.expand(-1, -1, x.shape[2], x.shape[3]))
    return conv_block(ctx, p + '.ConvLinear', torch.cat(parts, 1))


def pyramid_pooling(ctx, p, x, ks=(1, 2, 3, 6)):  # common.py:514-539
    h, w = x.shape[2:]
    feats = [x]
    for i, k in enumerate(ks):
        feats.append(up_bilinear(conv_block(ctx, f'{p}.conv{i + 1}', F.adaptive_avg_pool2d(x, k)), size=(h, w)))
    return torch.cat(feats, 1)


def ffm(ctx, p, x, k):  # common.py:210-230 (x already concatenated)
    feat = conv_block(ctx, p + '.convblk', x, k)
    a = F.adaptive_avg_pool2d(feat, 1)
    a = F.silu(F.conv2d(a, ctx.sd[p + '.channel_attention.1.weight']))
    a = torch.sigmoid(F.conv2d(a, ctx.sd[p + '.channel_attention.3.weight']))
    return feat * a + feat


def _dropout(ctx, x):
    if ctx.dropout_fn is not None and ctx.training:
        return ctx.dropout_fn(x)
    return F.dropout(x, ctx.dropout_p, ctx.training) if ctx.dropout_p > 0 else x


def seg_psp(ctx, p, xs):  # yolo.py:149-186
    f8 = conv_block(ctx, p + '.m8.0', xs[0])
    f16 = up_bilinear(conv_block(ctx, p + '.m16.0', xs[1]), 2)
    f32 = up_bilinear(conv_block(ctx, p + '.m32.0', xs[2]), 4)
    y = rfb2(ctx, p + '.out.0', torch.cat([f8, f16, f32], 1))
    y = pyramid_pooling(ctx, p + '.out.1', y)
    y = ffm(ctx, p + '.out.2', y, 3)
    low = F.conv2d(y, ctx.sd[p + '.out.3.weight'], ctx.sd[p + '.out.3.bias'])
    if ctx.record is not None:
        ctx.record['seg_lowres'] = low
    return up_bilinear(low, 8)


def seg_base(ctx, p, xs, n):  # yolo.py:129-146
    y = c3(ctx, p + '.m.0', xs[0], n, False)
    y = c3spp(ctx, p + '.m.1', y)
    y = _dropout(ctx, y)
    low = F.conv2d(y, ctx.sd[p + '.m.3.weight'], None, 1, 1)
    if ctx.record is not None:
        ctx.record['seg_lowres'] = low
    return up_bilinear(low, 8)


def seg_lab(ctx, p, xs):  # yolo.py:93-124
    e = conv_block(ctx, p + '.encoder.0', xs[1])
    e = up_bilinear(aspp(ctx, p + '.encoder.1', e, (3, 6, 9), False), 2)
    dt = conv_block(ctx, p + '.detail.1', conv_block(ctx, p + '.detail.0', xs[0]), 3)
    y = ffm(ctx, p + '.decoder.0', torch.cat([dt, e], 1), 1)
    y = conv_block(ctx, p + '.decoder.1', y, 3)
    low = F.conv2d(y, ctx.sd[p + '.decoder.2.weight'], ctx.sd[p + '.decoder.2.bias'])
    if ctx.record is not None:
        ctx.record['seg_lowres'] = low
    return up_bilinear(low, 8)


def seg_bise(ctx, p, xs):  # yolo.py:30-86
    f3 = rfb2(ctx, p + '.m32.0', xs[2], (2, 3), True)
    f3 = up_bilinear(conv_block(ctx, p + '.up32.0', f3, 3), 2)
    f2 = rfb2(ctx, p + '.m16.0', xs[1], (2, 3), False) + f3
    f2 = up_bilinear(conv_block(ctx, p + '.up16.0', f2, 3), 2)
    f1 = conv_block(ctx, p + '.m8.0', xs[0])
    y = _dropout(ctx, ffm(ctx, p + '.out.0', torch.cat([f1, f2], 1), 3))
    low = F.conv2d(y, ctx.sd[p + '.out.2.weight'], ctx.sd[p + '.out.2.bias'])
    if ctx.record is not None:
        ctx.record['seg_lowres'] = low
    main = up_bilinear(low, 8)
    if not ctx.training:
        return main

    def aux(q, feat, scale):
        a = conv_block(ctx, q + '.0', feat, 3)
        return up_bilinear(F.conv2d(a, ctx.sd[q + '.1.weight'], ctx.sd[q + '.1.bias']), scale)
    return [main, aux(p + '.aux16', f2, 8), aux(p + '.aux32', f3, 16)]


def detect(ctx, p, xs, nc, stride):  # yolo.py:206-230
    sd = ctx.sd
    no, na = nc + 5, sd[p + '.anchors'].shape[1]
    raw, z = [], []
    for i, x in enumerate(xs):
        y = F.conv2d(x, sd[f'{p}.m.{i}.weight'], sd[f'{p}.m.{i}.bias'])
        bs, _, ny, nx = y.shape
        y = y.view(bs, na, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
        raw.append(y)
        if not ctx.training:
            gy, gx = torch.meshgrid(torch.arange(ny, device=y.device), torch.arange(nx, device=y.device), indexing='ij')
            grid = torch.stack((gx, gy), 2).view(1, 1, ny, nx, 2).to(y.dtype)
            s = y.sigmoid()
            xy = (s[..., 0:2] * 2. - 0.5 + grid) * stride[i]
            wh = (s[..., 2:4] * 2) ** 2 * sd[p + '.anchor_grid'][i]
            z.append(torch.cat((xy, wh, s[..., 4:]), -1).view(bs, -1, no))
    return raw if ctx.training else (torch.cat(z, 1), raw)


def forward(cfg, sd, x, training, dropout_p=0.1, record=None, dropout_fn=None, maxpool_fn=None):
    """Model.forward_once (yolo.py:293-316): returns [det_out, seg_out].  dropout_fn: tests replay a given keep-mask (Ctx.dropout_fn);
    maxpool_fn: tests replay given arg-max choices of the SPP pools (Ctx.maxpool_fn)."""
    ctx = Ctx(sd, training, dropout_p, record)
    ctx.dropout_fn = dropout_fn
    ctx.maxpool_fn = maxpool_fn
    gd, gw, nc = cfg['depth_multiple'], cfg['width_multiple'], cfg['nc']
    ys = []
    for i, (f, n, m, args) in enumerate(cfg['backbone'] + cfg['head']):
        p = f'model.{i}'
        n = max(round(n * gd), 1) if n > 1 else n          # yolo.py:388
        if isinstance(f, int):
            xin = x if f == -1 else ys[f]
        else:
            xin = [x if j == -1 else ys[j] for j in f]
        if m == 'Focus':
            x = focus(ctx, p, xin)
        elif m == 'Conv':
            x = conv_block(ctx, p, xin, *args[1:])
        elif m == 'C3':
            x = c3(ctx, p, xin, n, args[1] if len(args) > 1 else True)
        elif m == 'SPP':
            x = spp(ctx, p, xin, args[1])
        elif m == 'nn.Upsample':
            x = F.interpolate(xin, scale_factor=args[1], mode=args[2])
        elif m == 'Concat':
            x = torch.cat(xin, 1)
        elif m == 'SegMaskPSP':
            x = seg_psp(ctx, p, xin)
        elif m == 'SegMaskBase':
            nn_ = max(round(args[1] * gd), 1) if args[1] > 1 else args[1]   # yolo.py:408
            x = seg_base(ctx, p, xin, nn_)
        elif m == 'SegMaskLab':
            x = seg_lab(ctx, p, xin)
        elif m == 'SegMaskBiSe':
            x = seg_bise(ctx, p, xin)
        elif m == 'Detect':
            stride = [8., 16., 32.]
            x = detect(ctx, p, xin, nc, stride)
        else:
            raise NotImplementedError(m)
        ys.append(x)
        if record is not None and m not in ('Detect',) and torch.is_tensor(x):
            record[f'layer{i}'] = x
    return [ys[-1], ys[-2]]


def fuse_state_dict(sd):
    """Model.fuse() (yolo.py:339-347) + fuse_conv_and_bn (torch_utils.py:182-202) on a state_dict:
    only `Conv` wrappers ('.conv.weight' + '.bn.*') are folded; bare Conv2d+BN pairs stay."""
    out = {}
    for k, v in sd.items():
        if k.endswith('.conv.weight') and (k[:-len('.conv.weight')] + '.bn.weight') in sd:
            p = k[:-len('.conv.weight')]
            g, b = sd[p + '.bn.weight'], sd[p + '.bn.bias']
            mu, var = sd[p + '.bn.running_mean'], sd[p + '.bn.running_var']
            scale = g / torch.sqrt(BN_EPS + var)
            out[k] = (scale.view(-1, 1) * v.reshape(v.shape[0], -1)).view(v.shape)
            out[p + '.conv.bias'] = b - g * mu / torch.sqrt(var + BN_EPS)
        elif '.bn.' in k and (k.split('.bn.')[0] + '.conv.weight') in sd:
            continue
        else:
            out[k] = v
    return out
