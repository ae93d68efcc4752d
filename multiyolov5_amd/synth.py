"""Synthetic Cityscapes-shaped inputs and random-init weights for bench.py / smoke: there is no network
for datasets or checkpoints.  numpy RandomState streams keyed by name, so every rank / box generates the same data."""
import zlib

import numpy as np
import torch


def _rs(key, seed):
    return np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def images(b, h, w, seed=1):
    """[B,3,H,W] f32 in [0,1): both reference loaders deliver [0,1] without mean/std (train.py:342)."""
    return torch.from_numpy(_rs('images', seed).uniform(0, 1, (b, 3, h, w)).astype(np.float32))


def det_targets(b, per_img=8, nc=10, seed=1):
    """[nt,6] rows (img, cls, x, y, w, h), normalised; small Cityscapes-like boxes."""
    rs = _rs('targets', seed)
    nt = per_img * b
    t = np.zeros((nt, 6), np.float32)
    t[:, 0] = rs.randint(0, b, nt)
    t[:, 1] = rs.randint(0, nc, nt)
    t[:, 2:4] = rs.uniform(0.1, 0.9, (nt, 2))
    t[:, 4:6] = rs.uniform(0.02, 0.22, (nt, 2))
    return torch.from_numpy(t)


def seg_targets(b, h, w, ncls=19, seed=1, blocky=8):
    """i64 [B,H,W] in {-1,0..18} (train ids, -1 = ignore), blocky tiles."""
    rs = _rs('segmask', seed)
    hh, ww = (h + blocky - 1) // blocky, (w + blocky - 1) // blocky
    coarse = rs.randint(-1, ncls, (b, hh, ww))
    m = np.repeat(np.repeat(coarse, blocky, 1), blocky, 2)[:, :h, :w]
    return torch.from_numpy(np.ascontiguousarray(m, dtype=np.int64))


def nms_pred(b, a, nc=10, seed=3, img_w=1024, img_h=512, clusters=40):
    """[B,A,5+nc] decoded predictions (xywh px, obj, cls) clustered so that suppression happens; distinct scores."""
    rs = _rs('nms', seed)
    p = np.zeros((b, a, 5 + nc), np.float32)
    for i in range(b):
        cx, cy = rs.uniform(50, img_w - 50, clusters), rs.uniform(50, img_h - 50, clusters)
        cw, ch = rs.uniform(20, 200, clusters), rs.uniform(20, 200, clusters)
        k = rs.randint(0, clusters, a)
        p[i, :, 0] = cx[k] + rs.normal(0, 6, a)
        p[i, :, 1] = cy[k] + rs.normal(0, 6, a)
        p[i, :, 2] = cw[k] * rs.uniform(0.8, 1.25, a)
        p[i, :, 3] = ch[k] * rs.uniform(0.8, 1.25, a)
        p[i, :, 4] = 1 / (1 + np.exp(-rs.normal(-4, 2, a)))
        cl = rs.normal(-2, 1.5, (a, nc))
        cl[np.arange(a), k % nc] += 4
        p[i, :, 5:] = 1 / (1 + np.exp(-cl))
    return torch.from_numpy(p)


def randomize_(model, seed=0):
    """random-init weights of the architecture with non-trivial BatchNorm affine / running statistics (a fresh BN is the
    identity and would hide normalisation cost paths); Detect biases keep the reference's prior (yolo.py:318-326)."""
    with torch.no_grad():
        for k, t in model.state_dict().items():
            rs = _rs(k, seed)
            if k.endswith('num_batches_tracked') or 'anchor' in k or k.startswith('model.25.m.') and k.endswith('bias'):
                continue
            if k.endswith('running_var'):
                v = rs.uniform(0.5, 1.5, tuple(t.shape))
            elif k.endswith('running_mean'):
                v = rs.normal(0, 0.1, tuple(t.shape))
            elif t.dim() == 4:
                fan_in = t.shape[1] * t.shape[2] * t.shape[3]
                v = rs.uniform(-1, 1, tuple(t.shape)) * np.sqrt(3.0 / fan_in)
            elif k.endswith('.weight'):
                v = rs.uniform(0.5, 1.5, tuple(t.shape))
            else:
                v = rs.normal(0, 0.1, tuple(t.shape))
            t.copy_(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)))
    return model
