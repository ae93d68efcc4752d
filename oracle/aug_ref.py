"""CPU restatement (TEST INFRASTRUCTURE) of the reference's segmentation training augmentation, numpy only:

* `sync_transform` -- SegmentationDataset.py:118-151 `_sync_transform` (mirror, random rescale, pad, random crop) with the random
  decisions passed in, + :225-228 `_mask_transform` / :166-183 `_class_to_index`.
* `pil_resize_bilinear` / `pil_resize_nearest` -- Pillow's 8-bit `Image.resize` (third-party: Pillow, 12.2.0 in this image; the
  reference calls it at SegmentationDataset.py:134-135), restated from src/libImaging/Resample.c (precompute_coeffs,
  normalize_coeffs_8bpc, ImagingResampleHorizontal/Vertical_8bpc: triangle filter widened by the down-scale factor, 2^-22 fixed
  point, uint8 intermediate between the passes) and Geometry.c (ImagingScaleAffine for NEAREST).  Pinned: tests/golden/augment_seg.npz
  was written by running the reference's own `_sync_transform` with the real Pillow (oracle/make_golden.py augment_case), and
  tests/test_augment_cpu.py checks this restatement against it bit for bit.
* `draw_params` -- the order of the `random` calls of `_sync_transform` (pinned by the golden's generator-state probe).
* `color_jitter` -- torchvision.transforms.ColorJitter on a PIL image (get_citys_loader, SegmentationDataset.py:462-466).  torchvision
  is NOT installed here (requirements.txt:11 `torchvision>=0.8.1`): its functional_pil.py is restated (adjust_brightness / contrast /
  saturation = PIL ImageEnhance, adjust_hue = HSV round trip with a wrapping uint8 add); the Pillow arithmetic underneath (Blend.c,
  Convert.c rgb2hsv / hsv2rgb, ImageStat mean) is pinned against the real Pillow: tests/golden/augment_jitter.npz was produced with
  PIL's own ImageEnhance / convert('HSV') calls (make_golden.py jitter_case), and the HSV conversions were checked over all 2^24
  colours.  The ORDER in which ColorJitter draws its random numbers is version-dependent and unpinned (see utils/augment.py).
"""
import math

import numpy as np

PB = 22


def _coeffs(in_size, out_size):
    scale = float(np.float32(in_size) - np.float32(0)) / out_size
    fs = max(scale, 1.0)
    support = 1.0 * fs
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / fs
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [0.0] * ksize
        ww = 0.0
        for x in range(xmax):
            v = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - v if v < 1.0 else 0.0
            ww += w[x]
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PB)) if w[x] < 0 else int(0.5 + w[x] * (1 << PB))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def pil_resize_bilinear(img, ow, oh):
    """Image.fromarray(img).resize((ow, oh), Image.BILINEAR) for uint8 [h,w,c]"""
    h, w = img.shape[:2]
    out = img
    if ow != w:
        b, k = _coeffs(w, ow)
        tmp = np.zeros((h, ow) + img.shape[2:], np.uint8)
        for xx in range(ow):
            xmin, n = b[xx]
            acc = (out[:, xmin:xmin + n].astype(np.int64) * k[xx, :n].reshape((1, n) + (1,) * (img.ndim - 2))).sum(1) + (1 << (PB - 1))
            tmp[:, xx] = np.clip(acc >> PB, 0, 255)
        out = tmp
    if oh != h:
        b, k = _coeffs(h, oh)
        res = np.zeros((oh,) + out.shape[1:], np.uint8)
        for yy in range(oh):
            ymin, n = b[yy]
            acc = (out[ymin:ymin + n].astype(np.int64) * k[yy, :n].reshape((n,) + (1,) * (img.ndim - 1))).sum(0) + (1 << (PB - 1))
            res[yy] = np.clip(acc >> PB, 0, 255)
        out = res
    return out


def _nearest_tab(in_size, out_size):
    a = float(np.float32(in_size) - np.float32(0)) / out_size
    xo = 0.0 + a * 0.5
    tab = np.zeros(out_size, np.int64)
    for x in range(out_size):
        tab[x] = -1 if xo < 0 else int(xo)
        xo += a
    return tab


def pil_resize_nearest(m, ow, oh):
    h, w = m.shape[:2]
    return m[_nearest_tab(h, oh)][:, _nearest_tab(w, ow)]


CITY_KEY = np.array([-1, -1, -1, -1, -1, -1, -1, -1, 0, 1, -1, -1, 2, 3, 4, -1, -1, -1, 5, -1, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
                     -1, -1, 16, 17, 18])
CITY_MAPPING = np.array(range(-1, len(CITY_KEY) - 1)).astype('int32')


def class_to_index(mask):
    """SegmentationDataset.py:166-173"""
    mask = mask.astype('int32').copy()
    mask[mask == 255] = 0
    index = np.digitize(mask.ravel(), CITY_MAPPING, right=True)
    return CITY_KEY[index].reshape(mask.shape).astype(np.int64)


def range_and_prob(base_size, low, high, std):
    from scipy import stats
    lo = math.ceil((base_size * low) / 32)
    hi = math.ceil((base_size * high) / 32)
    mean = math.ceil(base_size / 32) - 4
    x = np.array(list(range(lo, hi + 1)))
    p = stats.norm.pdf(x, mean, std)
    p = p / p.sum()
    return x, np.cumsum(p)


def draw_params(rng, w, h, base_size, crop_size, low=0.65, high=3, std=25):
    """the `random` calls of _sync_transform in order (118-149)"""
    flip = rng.random() < 0.5
    wc, hc = crop_size
    x, cum_p = range_and_prob(base_size, low, high, std)
    long_size = rng.choices(population=x, cum_weights=cum_p, k=1)[0] * 32
    if h > w:
        oh = long_size
        ow = int(1.0 * w * long_size / h + 0.5)
    else:
        ow = long_size
        oh = int(1.0 * h * long_size / w + 0.5)
    pw, ph = max(ow, wc), max(oh, hc)
    x1 = rng.randint(0, pw - wc)
    y1 = rng.randint(0, ph - hc)
    return dict(flip=flip, ow=int(ow), oh=int(oh), x1=x1, y1=y1, wc=wc, hc=hc)


def sync_transform(img, mask, p):
    if p['flip']:
        img, mask = img[:, ::-1], mask[:, ::-1]
    img = pil_resize_bilinear(np.ascontiguousarray(img), p['ow'], p['oh'])
    mask = pil_resize_nearest(np.ascontiguousarray(mask), p['ow'], p['oh'])
    ph, pw = max(p['oh'], p['hc']), max(p['ow'], p['wc'])
    im2 = np.zeros((ph, pw, 3), np.uint8)
    m2 = np.full((ph, pw), 255, np.uint8)
    im2[:p['oh'], :p['ow']] = img
    m2[:p['oh'], :p['ow']] = mask
    y1, x1 = p['y1'], p['x1']
    return im2[y1:y1 + p['hc'], x1:x1 + p['wc']], class_to_index(m2[y1:y1 + p['hc'], x1:x1 + p['wc']])


# ---- ColorJitter on a PIL image (torchvision functional_pil.py + Pillow ImageEnhance.py / Blend.c / Convert.c), restated ---------------
def to_l(x):
    x = x.astype(np.int64)
    return ((x[..., 0] * 19595 + x[..., 1] * 38470 + x[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(deg, img, f):
    f32 = np.float32(f)
    d, i = deg.astype(np.int32), img.astype(np.int32)
    t = (d.astype(np.float32) + f32 * (i - d).astype(np.float32)).astype(np.float32)
    if 0.0 <= f <= 1.0:
        return t.astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def rgb2hsv(a):
    r, g, b = (a[..., k].astype(np.int32) for k in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(np.float32)
    crs = np.where(cr == 0, np.float32(1), cr)
    mx = np.where(maxc == 0, 1, maxc).astype(np.float32)
    s = (cr / mx).astype(np.float32)
    rc, gc, bc = (((maxc - c).astype(np.float32) / crs).astype(np.float32) for c in (r, g, b))
    h = np.where(r == maxc, (bc - gc).astype(np.float32).astype(np.float64),
                 np.where(g == maxc, 2.0 + rc.astype(np.float64) - bc.astype(np.float64), 4.0 + gc.astype(np.float64) - rc.astype(np.float64)))
    h = h.astype(np.float32).astype(np.float64)
    h = np.fmod(h / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    gray = minc == maxc
    return np.stack([np.where(gray, 0, uh), np.where(gray, 0, us), maxc], -1).astype(np.uint8)


def _cround(x):
    return np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5))


def hsv2rgb(a):
    h, s, v = a[..., 0].astype(np.float32), a[..., 1], a[..., 2]
    hd = h.astype(np.float64) * 6.0 / 255.0
    i = np.floor(hd).astype(np.int64)
    f = (hd - i.astype(np.float32).astype(np.float64)).astype(np.float32)
    fs = (s.astype(np.float32).astype(np.float64) / 255.0).astype(np.float32)
    vf = v.astype(np.float32).astype(np.float64)
    p = _cround(vf * (1.0 - fs.astype(np.float64)))
    q = _cround(vf * (1.0 - (fs * f).astype(np.float32).astype(np.float64)))
    t = _cround(vf * (1.0 - (fs.astype(np.float64) * (1.0 - f.astype(np.float64)))))
    p, q, t = [np.clip(x, 0, 255).astype(np.uint8) for x in (p, q, t)]
    k = i % 6
    r = np.choose(k, [v, q, p, p, t, v]); g = np.choose(k, [t, v, v, q, p, p]); b = np.choose(k, [p, p, t, v, v, q])
    z = s == 0
    return np.stack([np.where(z, v, r), np.where(z, v, g), np.where(z, v, b)], -1).astype(np.uint8)


def color_jitter(img, order, brightness, contrast, saturation, hue):
    """ColorJitter.forward on a uint8 RGB array: adjustments in `order` (0 brightness, 1 contrast, 2 saturation, 3 hue)"""
    for op in order:
        if op == 0:
            img = blend(np.zeros_like(img), img, brightness)
        elif op == 1:
            Lm = to_l(img)
            mean = int(Lm.astype(np.float64).sum() / Lm.size + 0.5)
            img = blend(np.full_like(img, mean), img, contrast)
        elif op == 2:
            img = blend(np.repeat(to_l(img)[..., None], 3, 2), img, saturation)
        elif op == 3:
            hsv = rgb2hsv(img)
            hsv[..., 0] = (hsv[..., 0].astype(np.int64) + int(hue * 255)) % 256
            img = hsv2rgb(hsv)
    return img


# ---- detection augmentation: the OpenCV pieces utils/datasets.py calls, restated (cv2 is NOT installed here -> "parity unpinned";
# ---- make_golden.py serves them to the REAL reference code as its `cv2`, so everything around them -- random call order, mosaic
# ---- geometry, label arithmetic, flips -- is the reference's own) -------------------------------------------------------------------
def cv_get_rotation_matrix_2d(center, angle, scale):
    """cv2.getRotationMatrix2D (imgwarp.cpp): angle in degrees, counter-clockwise"""
    a = angle * np.pi / 180.0
    alpha, beta = math.cos(a) * scale, math.sin(a) * scale
    cx, cy = center
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]], np.float64)


def cv_invert_affine_for_warp(M):
    """the inversion cv::warpAffine applies to its 2x3 argument (no WARP_INVERSE_MAP)"""
    M = np.array(M, np.float64).reshape(2, 3).copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0] = A11
    M[0, 1] *= -D
    M[1, 0] *= -D
    M[1, 1] = A22
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2], M[1, 2] = b1, b2
    return M


def cv_warp_affine_u8(img, M, dsize, border=114):
    """cv2.warpAffine(img, M, dsize, flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=(border,)*3) for uint8 HxWx3: fixed-point
    source coordinates (AB_BITS 10, 1/32 pixel), bilinear weights (32-a)(32-b)*32 (sum 2^15), result (sum + 2^14) >> 15"""
    w, h = dsize
    Mi = cv_invert_affine_for_warp(M)
    xs, ys = np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64)
    adx = np.rint(Mi[0, 0] * xs * 1024.0).astype(np.int64)
    bdx = np.rint(Mi[1, 0] * xs * 1024.0).astype(np.int64)
    X0 = np.rint((Mi[0, 1] * ys + Mi[0, 2]) * 1024.0).astype(np.int64) + 16
    Y0 = np.rint((Mi[1, 1] * ys + Mi[1, 2]) * 1024.0).astype(np.int64) + 16
    X = (X0[:, None] + adx[None, :]) >> 5
    Y = (Y0[:, None] + bdx[None, :]) >> 5
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    fa, fb = Y & 31, X & 31
    H, W = img.shape[:2]
    src = img.astype(np.int64)

    def px(yy, xx):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return np.where(ok[..., None], v, border)
    w00, w01, w10, w11 = (32 - fa) * (32 - fb) * 32, (32 - fa) * fb * 32, fa * (32 - fb) * 32, fa * fb * 32
    acc = px(sy, sx) * w00[..., None] + px(sy, sx + 1) * w01[..., None] + px(sy + 1, sx) * w10[..., None] + px(sy + 1, sx + 1) * w11[..., None]
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


_SDIV = np.array([0] + [int(np.rint((255 << 12) / (1.0 * i))) for i in range(1, 256)], np.int64)
_HDIV = np.array([0] + [int(np.rint((180 << 12) / (6.0 * i))) for i in range(1, 256)], np.int64)


def cv_bgr2hsv_u8(img):
    """cv2.cvtColor(img, cv2.COLOR_BGR2HSV) for uint8 (color_hsv RGB2HSV_b, hrange 180, integer tables)"""
    b, g, r = (img[..., k].astype(np.int64) for k in range(3))
    v = np.maximum(b, np.maximum(g, r))
    vmin = np.minimum(b, np.minimum(g, r))
    diff = v - vmin
    vr = np.where(v == r, -1, 0)
    vg = np.where(v == g, -1, 0)
    s = (diff * _SDIV[v] + (1 << 11)) >> 12
    h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))))
    h = (h * _HDIV[diff] + (1 << 11)) >> 12
    h = h + np.where(h < 0, 180, 0)
    return np.stack([np.clip(h, 0, 255), s & 255, v], -1).astype(np.uint8)


def cv_hsv2bgr_u8(hsv):
    """cv2.cvtColor(hsv, cv2.COLOR_HSV2BGR) for uint8 (HSV2RGB_b: float32 sector formula, saturate_cast<uchar>(x * 255))"""
    f = np.float32
    h = hsv[..., 0].astype(f)
    s = (hsv[..., 1].astype(f) * f(1.0 / 255.0)).astype(f)
    v = (hsv[..., 2].astype(f) * f(1.0 / 255.0)).astype(f)
    hh = (h * f(6.0 / 180.0)).astype(f)
    hh = np.fmod(hh, f(6.0)).astype(f)
    sector = np.floor(hh).astype(np.int64)
    hh = (hh - sector.astype(f)).astype(f)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    hh = np.where(bad, f(0), hh).astype(f)
    one = f(1.0)
    t0 = v
    t1 = (v * (one - s)).astype(f)
    t2 = (v * (one - (s * hh).astype(f)).astype(f)).astype(f)
    t3 = (v * (one - (s * (one - hh).astype(f)).astype(f)).astype(f)).astype(f)
    tab = np.stack([t0, t1, t2, t3], -1)
    sd = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])
    idx = sd[sector]
    out = np.take_along_axis(tab, idx, -1)
    out = np.where((hsv[..., 1] == 0)[..., None], v[..., None], out)
    return np.clip(np.rint((out * f(255.0)).astype(f)), 0, 255).astype(np.uint8)


def install_cv2_stub(cv2):
    """give the stub `cv2` module of oracle/ref_shim the functions utils/datasets.py's augmentation path calls"""
    from . import frame_ref
    cv2.INTER_LINEAR, cv2.INTER_AREA = 1, 3
    cv2.COLOR_BGR2HSV, cv2.COLOR_HSV2BGR = 40, 54

    def resize(img, dsize, interpolation=1):
        assert interpolation == 1
        return frame_ref.cv_resize_linear_u8(img, dsize)

    def warp_affine(img, M, dsize, borderValue=(0, 0, 0)):
        return cv_warp_affine_u8(img, M, dsize, int(borderValue[0]))

    def cvt(img, code, dst=None):
        out = cv_bgr2hsv_u8(img) if code == cv2.COLOR_BGR2HSV else cv_hsv2bgr_u8(img)
        if dst is not None:
            dst[...] = out
            return dst
        return out
    cv2.resize, cv2.warpAffine, cv2.cvtColor = resize, warp_affine, cvt
    cv2.BORDER_CONSTANT = 0

    def copy_make_border(img, top, bottom, left, right, border_type, value=(0, 0, 0)):
        out = np.empty((img.shape[0] + top + bottom, img.shape[1] + left + right, img.shape[2]), img.dtype)
        out[...] = np.array(value, img.dtype)
        out[top:top + img.shape[0], left:left + img.shape[1]] = img
        return out
    cv2.copyMakeBorder = copy_make_border
    cv2.getRotationMatrix2D = lambda angle, center, scale: cv_get_rotation_matrix_2d(center, angle, scale)
    cv2.split = lambda m: [np.ascontiguousarray(m[..., k]) for k in range(m.shape[2])]
    cv2.merge = lambda planes: np.stack(planes, -1)
    cv2.LUT = lambda src, lut: lut[src]
    return cv2
