"""Frame -> model input on the device: the per-frame half of the reference's utils/datasets.py (`letterbox` 818-848, the
`img[:, :, ::-1].transpose(2, 0, 1)` of LoadImages.__next__ 185) fused with detect.py:135-137 (uint8 -> half/float -> /255).

The geometry (`letterbox_params`) is the reference's arithmetic verbatim in meaning; the pixels are moved by ONE libmyolo
kernel (`myolo_frame_pack`): the uint8 frame crosses PCIe once (3 bytes/pixel instead of the 6-12 of a host-normalised tensor)
and no intermediate HWC / float image exists.  Frames that letterbox resamples (`cv2.resize(img, new_unpad, INTER_LINEAR)`,
datasets.py:843-844) go through `myolo_frame_resize_pack`: OpenCV's 8-bit fixed-point bilinear restated in the kernel (cv2 is not
installed here, so that resampler is "parity unpinned" like copyMakeBorder; oracle/frame_ref.py holds the same restatement in numpy).
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib as L


def letterbox_params(shape, new_shape=(640, 640), auto=True, scaleFill=False, scaleup=True, stride=32):
    """datasets.py:818-846 without the pixels: ((new_unpad_w, new_unpad_h), ratio (w, h), (dw, dh), (top, bottom, left, right))."""
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
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return new_unpad, ratio, (dw, dh), (top, bottom, left, right)


_LUTS = {}


def _lut(device, dtype):
    """the 256 possible values of `x.to(dtype) / 255.0` (detect.py:136-137), computed by torch on the device itself"""
    key = (str(device), dtype)
    t = _LUTS.get(key)
    if t is None:
        t = torch.arange(256, device=device, dtype=torch.uint8).to(dtype)
        t /= 255.0
        _LUTS[key] = t
    return t


def frame_to_input(im0, new_shape=640, color=(114, 114, 114), auto=True, scaleFill=False, scaleup=True, stride=32, half=True,
                   bgr=True, out=None):
    """im0: uint8 [h,w,3] tensor on the GPU (a cv2 BGR frame uploaded as is).  Returns (img [1,3,H,W] fp16|fp32 in [0,1], RGB
    planes, ratio, (dw, dh)) = what `letterbox` + LoadImages + detect.py:135-139 hand to the model, ready for `scale_coords`."""
    L.require_gpu(im0)
    if im0.dtype != torch.uint8 or im0.dim() != 3 or im0.shape[2] != 3 or not im0.is_contiguous():
        raise L.MyoloError('frame must be a contiguous uint8 [h,w,3] tensor')
    if len(set(color)) != 1:
        raise NotImplementedError('letterbox border: one grey level (the reference always pads with 114)')
    h0, w0 = int(im0.shape[0]), int(im0.shape[1])
    new_unpad, ratio, (dw, dh), (top, bottom, left, right) = letterbox_params((h0, w0), new_shape, auto, scaleFill, scaleup, stride)
    rw, rh = int(new_unpad[0]), int(new_unpad[1])
    H, W = rh + top + bottom, rw + left + right
    dtype = torch.float16 if half else torch.float32
    if out is None:
        out = torch.empty(1, 3, H, W, dtype=dtype, device=im0.device)
    elif tuple(out.shape) != (1, 3, H, W) or out.dtype != dtype or not out.is_contiguous():
        raise L.MyoloError('out must be a contiguous [1,3,H,W] tensor of the requested dtype')
    if (w0, h0) == (rw, rh):
        L.check(L.lib().myolo_frame_pack(L.ptr(im0), h0, w0, int(bool(bgr)), top, left, H, W, int(color[0]), L.ptr(out), L.DT[dtype],
                                         L.ptr(_lut(im0.device, dtype)), L.stream_ptr()), 'myolo_frame_pack')
    else:
        L.check(L.lib().myolo_frame_resize_pack(L.ptr(im0), h0, w0, rh, rw, int(bool(bgr)), top, left, H, W, int(color[0]), L.ptr(out),
                                                L.DT[dtype], L.ptr(_lut(im0.device, dtype)), L.stream_ptr()), 'myolo_frame_resize_pack')
    return out, ratio, (dw, dh)
