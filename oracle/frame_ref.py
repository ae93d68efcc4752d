"""CPU restatement (TEST INFRASTRUCTURE) of detect.py's per-frame host steps, numpy only:

* `frame_to_input` -- utils/datasets.py:818-848 `letterbox` (border branch; pinned by tests/golden/letterbox.npz, written by running the
  reference's own letterbox with cv2.copyMakeBorder BORDER_CONSTANT served by a numpy constant fill -- cv2 is not installed), datasets.py:185
  `img[:, :, ::-1].transpose(2, 0, 1)`, detect.py:135-139 `torch.from_numpy(img).half()/float(); img /= 255.0; unsqueeze(0)`
  (this part IS the reference's own torch code, executed here on CPU).
* `seg_overlay` -- detect.py:69-72 `label2image` (numpy fancy indexing, as the reference) + `[:, :, ::-1]` + cv2.addWeighted restated from
  its documented definition dst = saturate(round(src1*alpha + src2*beta + gamma)) in float32 with round-half-even ("parity
  unpinned": cv2 absent; exact .5 ties are the only place an implementation could differ).
"""
import numpy as np
import torch


def cv_resize_linear_u8(img, dsize):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) for uint8 HxWxC, restated from OpenCV's resize.cpp ("parity unpinned":
    cv2 is not installed): float32 source coordinate fx = (dx + 0.5) * scale - 0.5 (scale in double), coefficients rounded half-to-even
    to 1/2048 fixed point, horizontal pass in int32, vertical pass (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2; x
    borders reset (fx = 0), y rows clamped; an exact 2x down-scale takes resize()'s INTER_AREA-fast branch ((a + b + c + d + 2) >> 2)."""
    h0, w0 = img.shape[:2]
    rw, rh = int(dsize[0]), int(dsize[1])
    src = img.astype(np.int64)
    if w0 == 2 * rw and h0 == 2 * rh:
        return ((src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    fx = ((np.arange(rw, dtype=np.float64) + 0.5) * (w0 / rw) - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    fx = fx - sx.astype(np.float32)
    lo, hi = sx < 0, sx >= w0 - 1
    fx = np.where(lo | hi, np.float32(0), fx)
    sx = np.where(lo, 0, np.where(hi, w0 - 1, sx))
    x1 = np.minimum(sx + 1, w0 - 1)
    a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(fx * np.float32(2048)).astype(np.int64)
    fy = ((np.arange(rh, dtype=np.float64) + 0.5) * (h0 / rh) - 0.5).astype(np.float32)
    sy = np.floor(fy).astype(np.int64)
    fy = fy - sy.astype(np.float32)
    y0, y1 = np.clip(sy, 0, h0 - 1), np.clip(sy + 1, 0, h0 - 1)
    b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int64)
    b1 = np.rint(fy * np.float32(2048)).astype(np.int64)
    hrow = src[:, sx] * a0[None, :, None] + src[:, x1] * a1[None, :, None]            # [h0, rw, C] int
    r0, r1 = hrow[y0], hrow[y1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox(img, new_shape=(640, 640), color=(114, 114, 114), auto=True, scaleFill=False, scaleup=True, stride=32):
    shape = img.shape[:2]
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    ratio = r, r
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    elif scaleFill:
        dw, dh = 0.0, 0.0
        new_unpad = (new_shape[1], new_shape[0])
        ratio = new_shape[1] / shape[1], new_shape[0] / shape[0]
    dw /= 2
    dh /= 2
    if shape[::-1] != new_unpad:
        img = cv_resize_linear_u8(img, new_unpad)               # datasets.py:843-844
        shape = img.shape[:2]
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    out = np.empty((shape[0] + top + bottom, shape[1] + left + right, 3), np.uint8)
    out[...] = np.array(color, np.uint8)
    out[top:top + shape[0], left:left + shape[1]] = img
    return out, ratio, (dw, dh)


def frame_to_input(im0, new_shape=640, stride=32, auto=True, half=True):
    img, ratio, pad = letterbox(im0, new_shape, stride=stride, auto=auto)
    img = np.ascontiguousarray(img[:, :, ::-1].transpose(2, 0, 1))
    t = torch.from_numpy(img)
    t = t.half() if half else t.float()
    t /= 255.0
    return t.unsqueeze(0), ratio, pad


def seg_overlay(labels, im0, colormap, alpha=0.4, beta=0.6, gamma=0.0):
    cm = np.array(colormap, dtype='uint8')
    mask = cm[labels.astype('int32'), :][:, :, ::-1]
    t = (mask.astype(np.float32) * np.float32(alpha) + im0.astype(np.float32) * np.float32(beta)) + np.float32(gamma)
    dst = np.clip(np.rint(t), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(mask), dst
