"""Training-time augmentation on the device -- host side.

Mirrors the reference's per-sample augmentation entry points (SegmentationDataset.py:118-151 `_sync_transform`, :25-45
`range_and_prob` / `get_long_size`, :166-183 `_class_to_index`): the random numbers are drawn here with Python's `random` in the
reference's call order (same seed -> same parameters), the pixels are moved by libmyolo (csrc/augment.hip).  Inputs are uint8
tensors already on the GPU (a decoded image is uploaded once, 3 bytes per pixel); there is no CPU path.
"""
import ctypes as C
import math
import random as _random
from functools import lru_cache

import numpy as np
import torch

from .. import _lib as L

PIL_BITS = 22                                            # Resample.c PRECISION_BITS


# ---- Pillow's resampling tables (Resample.c precompute_coeffs / normalize_coeffs_8bpc, BILINEAR = triangle filter, support 1) ---------
@lru_cache(64)
def pil_bilinear_tables(in_size, out_size):
    """(bounds int32 [out,2], coefficients int32 [out,ksize]) of Image.resize(..., BILINEAR) along one axis; identity when the
    axis is not resized (ImagingResampleInner skips that pass)."""
    if in_size == out_size:
        b = np.stack([np.arange(out_size), np.ones(out_size, np.int64)], 1).astype(np.int32)
        return b, np.full((out_size, 1), 1 << PIL_BITS, np.int32)
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size          # box coordinates are C floats
    fscale = max(scale, 1.0)
    support = 1.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / fscale
    xx = np.arange(out_size, dtype=np.float64)
    center = 0.0 + (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)           # (int) truncation of non-negative values
    neg = (center - support + 0.5) < 0
    xmin = np.where(neg, 0, xmin)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    j = np.arange(ksize, dtype=np.float64)[None, :]
    arg = np.abs((j + xmin[:, None] - center[:, None] + 0.5) * ss)
    w = np.where(arg < 1.0, 1.0 - arg, 0.0)
    w = np.where(j < xmax[:, None], w, 0.0)
    ww = np.zeros(out_size)
    for k in range(ksize):                                                     # sequential sum, as the C loop accumulates
        ww = ww + w[:, k]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww[:, None] != 0.0, ww[:, None], 1.0), w)
    kk = np.where(w < 0, (-0.5 + w * (1 << PIL_BITS)), (0.5 + w * (1 << PIL_BITS))).astype(np.int64).astype(np.int32)   # (int) truncates
    return np.stack([xmin, xmax], 1).astype(np.int32), kk


@lru_cache(64)
def pil_nearest_table(in_size, out_size):
    """source index of every output index for Image.resize(..., NEAREST) (Geometry.c ImagingScaleAffine: xo = a*0.5, then `xo += a`
    per pixel -- the running double sum is reproduced, not the closed form)"""
    a = float(np.float32(in_size) - np.float32(0.0)) / out_size
    steps = np.full(out_size, a)
    steps[0] = 0.0 + a * 0.5
    xo = np.add.accumulate(steps)                                              # sequential double additions
    return np.where(xo < 0.0, -1, xo.astype(np.int64)).astype(np.int32)


# ---- random scale (SegmentationDataset.py:25-45) ----------------------------------------------------------------------------------
@lru_cache(128)
def range_and_prob(base_size, low=0.5, high=3.0, std=25):
    from scipy import stats
    lo = math.ceil((base_size * low) / 32)
    hi = math.ceil((base_size * high) / 32)
    mean = math.ceil(base_size / 32) - 4
    x = np.array(list(range(lo, hi + 1)))
    p = stats.norm.pdf(x, mean, std)
    p = p / p.sum()
    return x, np.cumsum(p)


def get_long_size(base_size, low=0.5, high=3.0, std=40, rng=_random):
    x, cum_p = range_and_prob(base_size, low, high, std)
    return int(rng.choices(population=x, cum_weights=cum_p, k=1)[0] * 32)


def city_label_lut():
    """CitySegmentation._class_to_index (SegmentationDataset.py:166-183) as a 256-entry table: 255 (padding) -> id 0 -> -1, id v ->
    _key[v + 1]; ids the reference asserts against (34..254) map to -1"""
    key = np.array([-1, -1, -1, -1, -1, -1, -1, -1, 0, 1, -1, -1, 2, 3, 4, -1, -1, -1, 5, -1, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
                    -1, -1, 16, 17, 18], np.int64)
    lut = np.full(256, -1, np.int64)
    lut[:34] = key[1:35]
    lut[255] = key[1]
    return lut


def draw_sync_params(w, h, base_size, crop_size, low=0.65, high=3.0, std=25, rng=_random):
    """the random decisions of `_sync_transform` (SegmentationDataset.py:118-151) for a (w, h) image, in its call order:
    random.random() (mirror) -> get_long_size (one random.choices) -> random.randint x 2 (crop origin)"""
    flip = rng.random() < 0.5
    wc, hc = crop_size
    long_size = get_long_size(base_size, low, high, std, rng)
    if h > w:
        oh = long_size
        ow = int(1.0 * w * long_size / h + 0.5)
    else:
        ow = long_size
        oh = int(1.0 * h * long_size / w + 0.5)
    pw, ph = max(ow, wc), max(oh, hc)                           # size after ImageOps.expand
    x1 = rng.randint(0, pw - wc)
    y1 = rng.randint(0, ph - hc)
    return {'flip': bool(flip), 'ow': int(ow), 'oh': int(oh), 'x1': int(x1), 'y1': int(y1), 'wc': int(wc), 'hc': int(hc)}


def testval_size(w, h, base_size):
    """_testval_img_transform's target size (SegmentationDataset.py:80-94): long side -> base_size rounded up to a multiple of 32,
    the other side scaled (truncated) and rounded up to a multiple of 32"""
    div = lambda v: int(math.ceil(v / 32) * 32)                 # general.make_divisible(x, 32)
    outlong = div(base_size)
    if w > h:
        ow = outlong
        oh = div(int(1.0 * h * ow / w))
    else:
        oh = outlong
        ow = div(int(1.0 * w * oh / h))
    return ow, oh


_DEV_TABLES = {}


def _dev(arr, device, key):
    k = (str(device), key)
    t = _DEV_TABLES.get(k)
    if t is None:
        if len(_DEV_TABLES) > 256:
            _DEV_TABLES.clear()
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(device)
        _DEV_TABLES[k] = t
    return t


def seg_sync_transform(img, mask, params, lab_lut=None):
    """img: uint8 [H0,W0,3] (RGB, as `Image.open(...).convert('RGB')`) and mask: uint8 [H0,W0] on the GPU; params from
    draw_sync_params.  Returns (crop uint8 [hc,wc,3], label int64 [hc,wc]) = `_sync_transform` + `_mask_transform`."""
    return _seg_sync(img, mask, params, lab_lut, want_img=True)


def _seg_sync(img, mask, params, lab_lut, want_img):
    L.require_gpu(img)
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3 or not img.is_contiguous():
        raise L.MyoloError('image must be a contiguous uint8 [H,W,3] tensor')
    H0, W0 = int(img.shape[0]), int(img.shape[1])
    if mask is not None and (mask.dtype != torch.uint8 or tuple(mask.shape) != (H0, W0) or not mask.is_contiguous()):
        raise L.MyoloError('label map must be a contiguous uint8 [H,W] tensor matching the image')
    dev = img.device
    ow, oh, wc, hc = params['ow'], params['oh'], params['wc'], params['hc']
    hb, hk = pil_bilinear_tables(W0, ow)
    vb, vk = pil_bilinear_tables(H0, oh)
    d = L.SegSyncDesc()
    keep = [_dev(hb, dev, ('hb', W0, ow)), _dev(hk, dev, ('hk', W0, ow)), _dev(vb, dev, ('vb', H0, oh)), _dev(vk, dev, ('vk', H0, oh))]
    d.img, d.H0, d.W0, d.flip, d.ow, d.oh = img.data_ptr(), H0, W0, int(params['flip']), ow, oh
    d.hb, d.hk, d.vb, d.vk, d.ksh, d.ksv = keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(), hk.shape[1], vk.shape[1]
    d.x1, d.y1, d.wc, d.hc = params['x1'], params['y1'], wc, hc
    out = torch.empty(hc, wc, 3, dtype=torch.uint8, device=dev) if want_img else None
    d.out_img = out.data_ptr() if want_img else None
    lab = None
    if mask is not None:
        lut = city_label_lut() if lab_lut is None else np.asarray(lab_lut, np.int64)
        keep += [_dev(pil_nearest_table(W0, ow), dev, ('xin', W0, ow)), _dev(pil_nearest_table(H0, oh), dev, ('yin', H0, oh)),
                 _dev(lut, dev, ('lut', lut.tobytes()))]
        lab = torch.empty(hc, wc, dtype=torch.int64, device=dev)
        d.mask, d.xin, d.yin, d.lab_lut, d.out_lab = mask.data_ptr(), keep[4].data_ptr(), keep[5].data_ptr(), keep[6].data_ptr(), lab.data_ptr()
    L.check(L.lib().myolo_seg_sync_transform(C.byref(d), L.stream_ptr()), 'myolo_seg_sync_transform')
    return out, lab


def seg_testval_transform(img, mask, base_size, lab_lut=None):
    """the validation sample of train.py:228-229 / test.py:71 (mode='testval', SegmentationDataset.py:201-204): the image resized
    (BILINEAR) to testval_size, the label map only through `_mask_transform`.  Returns (uint8 [oh,ow,3], int64 [H0,W0])."""
    H0, W0 = int(img.shape[0]), int(img.shape[1])
    ow, oh = testval_size(W0, H0, base_size)
    a, _ = seg_sync_transform(img, None, {'flip': False, 'ow': ow, 'oh': oh, 'x1': 0, 'y1': 0, 'wc': ow, 'hc': oh})
    lab = None
    if mask is not None:
        lab = seg_label_map(img, mask, lab_lut)
    return a, lab


def seg_label_map(img, mask, lab_lut=None):
    """CitySegmentation._mask_transform (SegmentationDataset.py:225-228) of an un-resized label map: id -> train id, int64"""
    H0, W0 = int(mask.shape[0]), int(mask.shape[1])
    p = {'flip': False, 'ow': W0, 'oh': H0, 'x1': 0, 'y1': 0, 'wc': W0, 'hc': H0}
    return _seg_sync(img, mask, p, lab_lut, want_img=False)[1]


# ---- ColorJitter + ToTensor (get_citys_loader: ColorJitter(brightness=0.45, contrast=0.45, saturation=0.45, hue=0.15)) -----------------
def draw_color_jitter_params(brightness=0.45, contrast=0.45, saturation=0.45, hue=0.15, generator=None):
    """torchvision.transforms.ColorJitter.get_params (torch-RNG versions, >= 0.9): torch.randperm(4), then one uniform_ each for
    brightness, contrast, saturation, hue.  (0.8.x shuffled with Python's `random`; torchvision is not installed here, the version
    the reference ran with is not pinned by requirements.txt -- the order of the draws is "parity unpinned".)"""
    order = torch.randperm(4, generator=generator).tolist()
    u = lambda lo, hi: float(torch.empty(1).uniform_(lo, hi, generator=generator))
    b = u(max(0.0, 1 - brightness), 1 + brightness)
    c = u(max(0.0, 1 - contrast), 1 + contrast)
    s = u(max(0.0, 1 - saturation), 1 + saturation)
    h = u(-hue, hue)
    return {'order': order, 'brightness': b, 'contrast': c, 'saturation': s, 'hue': h}


_LUTS = {}


def _unit_lut(device, dtype):
    """ToTensor's 256 values: uint8 -> float32 `.div(255)` (then cast).  Computed by torch ON THE CPU, where the reference's DataLoader
    workers run ToTensor: the GPU division by a scalar multiplies by the reciprocal and differs in the last bit for some values."""
    k = (str(device), dtype)
    t = _LUTS.get(k)
    if t is None:
        t = (torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255)).to(dtype).to(device)
        _LUTS[k] = t
    return t


def color_jitter(img, params, dtype=torch.float32, return_uint8=False):
    """img: uint8 [h,w,3] RGB on the GPU.  Returns the jittered image as ToTensor would hand it to the model ([3,h,w] `dtype` in [0,1])
    or, with return_uint8, the uint8 [h,w,3] PIL-equivalent image."""
    L.require_gpu(img)
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3 or not img.is_contiguous():
        raise L.MyoloError('image must be a contiguous uint8 [H,W,3] tensor')
    h, w = int(img.shape[0]), int(img.shape[1])
    order = (C.c_int32 * 4)(*[int(o) for o in params['order']])
    hue_u8 = int(params['hue'] * 255) % 256                 # functional_pil.adjust_hue: np.uint8(hue_factor * 255) added with wrap-around
    scratch = torch.empty(1, dtype=torch.int64, device=img.device)
    out_u8 = torch.empty(h, w, 3, dtype=torch.uint8, device=img.device) if return_uint8 else None
    out_f = None if return_uint8 else torch.empty(3, h, w, dtype=dtype, device=img.device)
    L.check(L.lib().myolo_color_jitter(L.ptr(img), h, w, order, C.c_float(params['brightness']), C.c_float(params['contrast']),
                                       C.c_float(params['saturation']), hue_u8, L.ptr(scratch), L.ptr(out_u8) if return_uint8 else None,
                                       None if return_uint8 else L.ptr(out_f), L.DT.get(dtype, 0) if not return_uint8 else 0,
                                       None if return_uint8 else L.ptr(_unit_lut(img.device, dtype)), L.stream_ptr()), 'myolo_color_jitter')
    return out_u8 if return_uint8 else out_f


# ---- detection samples: mosaic + random_perspective + augment_hsv + flips (utils/datasets.py:518-593, 646-658, 672-724, 851-937) --------
def resize_u8(img, rw, rh):
    """cv2.resize(img, (rw, rh), interpolation=cv2.INTER_LINEAR) of load_image (datasets.py:638-640) on the device"""
    L.require_gpu(img)
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3 or not img.is_contiguous():
        raise L.MyoloError('image must be a contiguous uint8 [H,W,3] tensor')
    out = torch.empty(int(rh), int(rw), 3, dtype=torch.uint8, device=img.device)
    L.check(L.lib().myolo_resize_u8(L.ptr(img), int(img.shape[0]), int(img.shape[1]), int(rh), int(rw), L.ptr(out), L.stream_ptr()),
            'myolo_resize_u8')
    return out


def load_image_dev(img0, img_size):
    """load_image (datasets.py:629-641) for a training (augment=True) dataset: long side -> img_size, always INTER_LINEAR"""
    h0, w0 = int(img0.shape[0]), int(img0.shape[1])
    r = img_size / max(h0, w0)
    if r != 1:
        img0 = resize_u8(img0, int(w0 * r), int(h0 * r))
    return img0, (h0, w0), (int(img0.shape[0]), int(img0.shape[1]))


def _boxes_norm_to_pixels(b, w, h, padw, padh):
    """general.xywhn2xyxy: normalised centre boxes -> pixel corner boxes shifted by the paste offset (dtype of `b` is kept)"""
    out = np.copy(b)
    half_w, half_h = b[:, 2] / 2, b[:, 3] / 2
    out[:, 0] = w * (b[:, 0] - half_w) + padw
    out[:, 1] = h * (b[:, 1] - half_h) + padh
    out[:, 2] = w * (b[:, 0] + half_w) + padw
    out[:, 3] = h * (b[:, 1] + half_h) + padh
    return out


def mosaic_layout(s, yc, xc, hw4):
    """the four paste windows of load_mosaic (datasets.py:684-699): per image (x1a, y1a, x2a, y2a) on the 2s x 2s canvas and the
    offsets (padw, padh) = canvas - source coordinates"""
    lay = []
    for i, (h, w) in enumerate(hw4):
        if i == 0:                                                   # top left: bottom-right corner of the image at the centre
            x1a, y1a, x2a, y2a = max(xc - w, 0), max(yc - h, 0), xc, yc
            x1b, y1b = w - (x2a - x1a), h - (y2a - y1a)
        elif i == 1:                                                 # top right
            x1a, y1a, x2a, y2a = xc, max(yc - h, 0), min(xc + w, s * 2), yc
            x1b, y1b = 0, h - (y2a - y1a)
        elif i == 2:                                                 # bottom left
            x1a, y1a, x2a, y2a = max(xc - w, 0), yc, xc, min(s * 2, yc + h)
            x1b, y1b = w - (x2a - x1a), 0
        else:                                                        # bottom right
            x1a, y1a, x2a, y2a = xc, yc, min(xc + w, s * 2), min(s * 2, yc + h)
            x1b, y1b = 0, 0
        lay.append((x1a, y1a, x2a, y2a, x1a - x1b, y1a - y1b))
    return lay


def draw_perspective_matrix(shape_hw, hyp, border, rng=_random):
    """random_perspective's matrix (datasets.py:857-888) with its `random` calls in order; returns (M 3x3, scale s, (width, height))"""
    height, width = shape_hw[0] + border[0] * 2, shape_hw[1] + border[1] * 2
    Cm = np.eye(3)
    Cm[0, 2], Cm[1, 2] = -shape_hw[1] / 2, -shape_hw[0] / 2
    Pm = np.eye(3)
    Pm[2, 0] = rng.uniform(-hyp['perspective'], hyp['perspective'])
    Pm[2, 1] = rng.uniform(-hyp['perspective'], hyp['perspective'])
    a = rng.uniform(-hyp['degrees'], hyp['degrees'])
    sc = rng.uniform(1 - hyp['scale'], 1 + hyp['scale'])
    Rm = np.eye(3)
    ang = a * np.pi / 180.0                                           # cv2.getRotationMatrix2D(angle=a, center=(0, 0), scale=sc)
    alpha, beta = math.cos(ang) * sc, math.sin(ang) * sc
    Rm[0] = [alpha, beta, 0.0]
    Rm[1] = [-beta, alpha, 0.0]
    Sm = np.eye(3)
    Sm[0, 1] = math.tan(rng.uniform(-hyp['shear'], hyp['shear']) * math.pi / 180)
    Sm[1, 0] = math.tan(rng.uniform(-hyp['shear'], hyp['shear']) * math.pi / 180)
    Tm = np.eye(3)
    Tm[0, 2] = rng.uniform(0.5 - hyp['translate'], 0.5 + hyp['translate']) * width
    Tm[1, 2] = rng.uniform(0.5 - hyp['translate'], 0.5 + hyp['translate']) * height
    return Tm @ Sm @ Rm @ Pm @ Cm, sc, (width, height)


def warp_boxes(targets, M, sc, width, height):
    """random_perspective's label half for box labels (datasets.py:903-925): corners through M, new axis-aligned boxes clipped to the
    output, box_candidates filter (wh > 2 px, area ratio > 0.1, aspect ratio < 20)"""
    n = len(targets)
    if not n:
        return targets
    corners = np.ones((n * 4, 3))
    corners[:, :2] = targets[:, [1, 2, 3, 4, 1, 4, 3, 2]].reshape(n * 4, 2)
    corners = (corners @ M.T)[:, :2].reshape(n, 8)
    xs, ys = corners[:, [0, 2, 4, 6]], corners[:, [1, 3, 5, 7]]
    new = np.stack([xs.min(1), ys.min(1), xs.max(1), ys.max(1)], 1)
    new[:, [0, 2]] = new[:, [0, 2]].clip(0, width)
    new[:, [1, 3]] = new[:, [1, 3]].clip(0, height)
    old = targets[:, 1:5].T * sc
    w1, h1 = old[2] - old[0], old[3] - old[1]
    w2, h2 = new[:, 2] - new[:, 0], new[:, 3] - new[:, 1]
    eps = 1e-16
    ar = np.maximum(w2 / (h2 + eps), h2 / (w2 + eps))
    keep = (w2 > 2) & (h2 > 2) & (w2 * h2 / (w1 * h1 + eps) > 0.10) & (ar < 20)
    targets = targets[keep]
    targets[:, 1:5] = new[keep]
    return targets


def hsv_luts(hyp, nprng=np.random):
    """augment_hsv's three tables (datasets.py:647-654): one np.random.uniform(-1, 1, 3) draw"""
    r = nprng.uniform(-1, 1, 3) * [hyp['hsv_h'], hyp['hsv_s'], hyp['hsv_v']] + 1
    x = np.arange(0, 256, dtype=np.int16)
    return np.stack([((x * r[0]) % 180).astype(np.uint8), np.clip(x * r[1], 0, 255).astype(np.uint8), np.clip(x * r[2], 0, 255).astype(np.uint8)])


def _invert_for_warp(M):
    """the inversion cv::warpAffine applies to its 2x3 argument, in its operation order (double)"""
    m = np.array(M[:2], np.float64).copy()
    D = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[1, 1] * D, m[0, 0] * D
    m[0, 0] = A11
    m[0, 1] *= -D
    m[1, 0] *= -D
    m[1, 1] = A22
    b1 = -m[0, 0] * m[0, 2] - m[0, 1] * m[1, 2]
    b2 = -m[1, 0] * m[0, 2] - m[1, 1] * m[1, 2]
    m[0, 2], m[1, 2] = b1, b2
    return m


def _finish_sample(lab, width, height, hyp, rng, nprng):
    """the tail of __getitem__ shared by both branches (datasets.py:558-584): augment_hsv draw, pixel xyxy -> normalised xywh, flips"""
    lut = hsv_luts(hyp, nprng)
    nL = len(lab)
    if nL:                                                            # general.xyxy2xywh, then normalise by the image size
        x1, y1, x2, y2 = lab[:, 1].copy(), lab[:, 2].copy(), lab[:, 3].copy(), lab[:, 4].copy()
        lab[:, 1], lab[:, 2], lab[:, 3], lab[:, 4] = (x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1
        lab[:, [2, 4]] /= height
        lab[:, [1, 3]] /= width
    flipud = rng.random() < hyp['flipud']
    if flipud and nL:
        lab[:, 2] = 1 - lab[:, 2]
    fliplr = rng.random() < hyp['fliplr']
    if fliplr and nL:
        lab[:, 1] = 1 - lab[:, 1]
    out_lab = torch.zeros((nL, 6))
    if nL:
        out_lab[:, 1:] = torch.from_numpy(lab)
    return lut, flipud, fliplr, out_lab


def _launch_warp(srcs, canvas_wh, M, warp, out_wh, lut, fliplr, flipud):
    """srcs: [(uint8 [h,w,3] tensor, x1a, y1a, x2a, y2a, padw, padh)] -> uint8 [3,oh,ow] RGB"""
    dev = srcs[0][0].device
    d = L.MosaicDesc()
    d.nsrc, d.cw, d.ch, d.warp, d.ow, d.oh = len(srcs), canvas_wh[0], canvas_wh[1], int(warp), out_wh[0], out_wh[1]
    for k, (im, x1a, y1a, x2a, y2a, padw, padh) in enumerate(srcs):
        L.require_gpu(im)
        if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3 or not im.is_contiguous():
            raise L.MyoloError('images must be contiguous uint8 [h,w,3] tensors')
        if x2a <= x1a or y2a <= y1a:                                  # empty window (the centre lies outside the canvas on that side)
            x1a = y1a = x2a = y2a = 0
            padw = padh = 0
        sk = d.src[k]
        sk.img, sk.h, sk.w = im.data_ptr(), int(im.shape[0]), int(im.shape[1])
        sk.x1a, sk.y1a, sk.x2a, sk.y2a, sk.padw, sk.padh = x1a, y1a, x2a, y2a, padw, padh
    Mi = _invert_for_warp(M) if warp else np.array([[1.0, 0, 0], [0, 1.0, 0]])
    for k in range(6):
        d.M[k] = float(Mi.reshape(-1)[k])
    lut_d = torch.from_numpy(np.ascontiguousarray(lut)).to(dev)
    out = torch.empty(3, out_wh[1], out_wh[0], dtype=torch.uint8, device=dev)
    d.hsv_lut, d.fliplr, d.flipud, d.fill, d.out_chw = lut_d.data_ptr(), int(fliplr), int(flipud), 114, out.data_ptr()
    L.check(L.lib().myolo_mosaic_warp(C.byref(d), L.stream_ptr()), 'myolo_mosaic_warp')
    return out


def _single_train_sample(index, images, labels, img_size, hyp, rng, nprng):
    """the non-mosaic training branch of __getitem__ (datasets.py:536-556): letterbox(img, img_size, auto=False, scaleup=True) of the
    (already long-side-resized) image -> random_perspective with border (0, 0) -> augment_hsv -> flips"""
    from .datasets import letterbox_params
    im = images(index)
    h, w = int(im.shape[0]), int(im.shape[1])
    new_unpad, ratio, (dw, dh), (top, bottom, left, right) = letterbox_params((h, w), img_size, auto=False, scaleFill=False, scaleup=True)
    if tuple(new_unpad) != (w, h):
        im = resize_u8(im, new_unpad[0], new_unpad[1])                # (only when the cached image is not at the target long side)
    H, W = new_unpad[1] + top + bottom, new_unpad[0] + left + right
    lab = labels(index).copy()
    if lab.size:
        lab[:, 1:] = _boxes_norm_to_pixels(lab[:, 1:], ratio[0] * w, ratio[1] * h, dw, dh)
    M, sc, (width, height) = draw_perspective_matrix((H, W), hyp, (0, 0), rng)
    if hyp['perspective']:
        raise NotImplementedError('perspective != 0 (cv2.warpPerspective) is not on the device path; hyp.scratch.yaml uses 0')
    warp = bool((M != np.eye(3)).any())
    lab = warp_boxes(lab, M, sc, width, height)
    lut, flipud, fliplr, out_lab = _finish_sample(lab, W, H, hyp, rng, nprng)
    out = _launch_warp([(im, left, top, left + new_unpad[0], top + new_unpad[1], left, top)], (W, H), M, warp, (width, height), lut, fliplr, flipud)
    return out, out_lab


def mosaic_train_sample(index, images, labels, indices, img_size, hyp, rng=_random, nprng=np.random):
    """LoadImagesAndLabels.__getitem__ for a training dataset with mosaic (datasets.py:518-593).
    images: index -> uint8 [h,w,3] BGR tensor on the GPU, already `load_image`d (long side = img_size; see load_image_dev);
    labels: index -> float32 [n,5] (cls, normalised xywh).  Returns (uint8 [3,s,s] RGB tensor on the GPU, labels_out float32 [nL,6])."""
    s = img_size
    if not (rng.random() < hyp['mosaic']):
        return _single_train_sample(index, images, labels, img_size, hyp, rng, nprng)
    border = [-s // 2, -s // 2]
    yc, xc = [int(rng.uniform(-x, 2 * s + x)) for x in border]
    idx4 = [index] + rng.choices(indices, k=3)
    imgs = [images(i) for i in idx4]
    hw4 = [(int(im.shape[0]), int(im.shape[1])) for im in imgs]
    lay = mosaic_layout(s, yc, xc, hw4)
    parts = []
    for i, (h, w) in zip(idx4, hw4):
        x1a, y1a, x2a, y2a, padw, padh = lay[len(parts)]
        lb = labels(i).copy()
        if lb.size:
            lb[:, 1:] = _boxes_norm_to_pixels(lb[:, 1:], w, h, padw, padh)
        parts.append(lb)
    lab4 = np.concatenate(parts, 0)
    np.clip(lab4[:, 1:], 0, 2 * s, out=lab4[:, 1:])
    M, sc, (width, height) = draw_perspective_matrix((2 * s, 2 * s), hyp, border, rng)
    if hyp['perspective']:
        raise NotImplementedError('perspective != 0 (cv2.warpPerspective) is not on the device path; hyp.scratch.yaml uses 0')
    lab4 = warp_boxes(lab4, M, sc, width, height)
    if rng.random() < hyp['mixup']:
        raise NotImplementedError('mixup (hyp.scratch.yaml: 0.0) is not on the device path')
    lut, flipud, fliplr, out_lab = _finish_sample(lab4, width, height, hyp, rng, nprng)
    srcs = [(im, *lay[k]) for k, im in enumerate(imgs)]
    out = _launch_warp(srcs, (2 * s, 2 * s), M, True, (width, height), lut, fliplr, flipud)
    return out, out_lab
