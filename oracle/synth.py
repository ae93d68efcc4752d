"""Deterministic synthetic weights / inputs shared by the golden generator and the tests
(TEST INFRASTRUCTURE ONLY).  numpy's legacy RandomState stream is frozen across numpy
versions, so `f(key, shape, seed)` is reproducible on any box without shipping weights.
"""
import zlib

import numpy as np
import torch


def _rs(key, seed):
    return np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def synth_tensor(key, shape, seed=0):
    """Value distribution by parameter kind (fresh-init BN is identity and hides bugs,
    so BN affine/running stats are randomised too)."""
    rs = _rs(key, seed)
    shape = tuple(shape)
    if key.endswith('num_batches_tracked'):
        return torch.tensor(1, dtype=torch.long)
    if key.endswith('running_var'):
        v = rs.uniform(0.5, 1.5, shape)
    elif key.endswith('running_mean'):
        v = rs.normal(0, 0.1, shape)
    elif key.endswith('anchors') or key.endswith('anchor_grid'):
        raise KeyError('anchors are structural, not synthesised')
    elif len(shape) == 4:                                    # conv weight OIHW: unit-gain fan-in init
        fan_in = shape[1] * shape[2] * shape[3]
        v = rs.uniform(-1, 1, shape) * np.sqrt(3.0 / fan_in) * 1.4
    elif key.endswith('.weight'):                            # BN gamma
        v = rs.uniform(0.5, 1.5, shape)
    elif key.endswith('.bias'):
        v = rs.normal(0, 0.1, shape)
    else:
        v = rs.normal(0, 1, shape)
    return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))


def synth_state_dict(template, seed=0):
    """template: dict name -> tensor (shapes/dtypes taken from it). anchors / anchor_grid are copied."""
    out = {}
    for k in template:
        t = template[k]
        if k.endswith('anchors') or k.endswith('anchor_grid'):
            out[k] = t.clone()
        else:
            out[k] = synth_tensor(k, t.shape, seed)
    # Detect biases: keep the reference's prior-style offsets so obj/cls logits are realistic (yolo.py:318-326)
    return out


def synth_images(b, h, w, seed=1):
    """[B,3,H,W] f32 in [0,1) (train.py:342 / SegmentationDataset.py:465-466 deliver [0,1] without mean/std)."""
    return torch.from_numpy(_rs('images', seed).uniform(0, 1, (b, 3, h, w)).astype(np.float32))


def synth_det_targets(b, per_img=8, nc=10, seed=1):
    """[nt,6] rows (img, cls, x, y, w, h) normalised -- SURVEY §8(d)."""
    rs = _rs('targets', seed)
    nt = per_img * b
    t = np.zeros((nt, 6), np.float32)
    t[:, 0] = rs.randint(0, b, nt)
    t[:, 1] = rs.randint(0, nc, nt)
    t[:, 2:4] = rs.uniform(0.1, 0.9, (nt, 2))
    t[:, 4:6] = rs.uniform(0.02, 0.22, (nt, 2))
    return torch.from_numpy(t)


def synth_seg_targets(b, h, w, ncls=19, seed=1, blocky=8):
    """i64 [B,H,W] in {-1,0..18}; blocky tiles so OHEM sees easy + hard regions."""
    rs = _rs('segmask', seed)
    hh, ww = (h + blocky - 1) // blocky, (w + blocky - 1) // blocky
    coarse = rs.randint(-1, ncls, (b, hh, ww))
    m = np.repeat(np.repeat(coarse, blocky, 1), blocky, 2)[:, :h, :w]
    return torch.from_numpy(np.ascontiguousarray(m, dtype=np.int64))


def synth_nms_pred(b, a, nc=10, seed=3, img_w=1024, img_h=512, clusters=40):
    """[B,A,5+nc] decoded predictions (xywh pixels, obj, cls probs) with clustered boxes and distinct
    scores so suppression happens and ties do not (SURVEY §8(d), App. B NMS)."""
    rs = _rs('nms', seed)
    p = np.zeros((b, a, 5 + nc), np.float32)
    for i in range(b):
        cx = rs.uniform(50, img_w - 50, clusters)
        cy = rs.uniform(50, img_h - 50, clusters)
        cw = rs.uniform(20, 200, clusters)
        ch = rs.uniform(20, 200, clusters)
        k = rs.randint(0, clusters, a)
        p[i, :, 0] = cx[k] + rs.normal(0, 6, a)
        p[i, :, 1] = cy[k] + rs.normal(0, 6, a)
        p[i, :, 2] = cw[k] * rs.uniform(0.8, 1.25, a)
        p[i, :, 3] = ch[k] * rs.uniform(0.8, 1.25, a)
        obj_logit = rs.normal(-4, 2, a)
        p[i, :, 4] = 1 / (1 + np.exp(-obj_logit))
        cls_logit = rs.normal(-2, 1.5, (a, nc))
        cls_logit[np.arange(a), k % nc] += 4          # cluster-consistent dominant class
        p[i, :, 5:] = 1 / (1 + np.exp(-cls_logit))
    return torch.from_numpy(p)
