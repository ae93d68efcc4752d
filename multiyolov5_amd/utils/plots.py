"""Segmentation overlay on the device: detect.py:69-72 `label2image` + detect.py:193-194 (`[:, :, ::-1]`, `cv2.addWeighted(mask,
0.4, im0, 0.6, 0)`), one libmyolo kernel (`myolo_seg_blend`) on the label map that `utils.general.seg_argmax` leaves in HBM --
only the finished BGR frame has to cross PCIe."""
import ctypes as C

import torch

from .. import _lib as L

# the Cityscapes train-id palette (RGB), detect.py:19-39
Cityscapes_COLORMAP = [
    [128, 64, 128], [244, 35, 232], [70, 70, 70], [102, 102, 156], [190, 153, 153], [153, 153, 153], [250, 170, 30], [220, 220, 0],
    [107, 142, 35], [152, 251, 152], [0, 130, 180], [220, 20, 60], [255, 0, 0], [0, 0, 142], [0, 0, 70], [0, 60, 100], [0, 80, 100],
    [0, 0, 230], [119, 11, 32],
]

_CMAPS = {}


def _cmap(colormap, device):
    key = (str(device), tuple(map(tuple, colormap)))
    t = _CMAPS.get(key)
    if t is None:
        t = _CMAPS[key] = torch.tensor(colormap, dtype=torch.uint8, device=device).contiguous()
    return t


def seg_overlay(labels, im0=None, colormap=Cityscapes_COLORMAP, alpha=0.4, beta=0.6, gamma=0.0, bgr=True, want_mask=True):
    """labels [h,w] (uint8 | int64, e.g. `seg_argmax(seg, h0, w0)[0]`), im0 uint8 [h,w,3] on the GPU.  Returns (mask, dst):
    mask = label2image(labels, colormap)[:, :, ::-1] (BGR when `bgr`), dst = cv2.addWeighted(mask, alpha, im0, beta, gamma);
    dst is None without im0, mask is None with want_mask=False."""
    L.require_gpu(labels)
    if labels.dim() != 2 or labels.dtype not in (torch.uint8, torch.int64) or not labels.is_contiguous():
        raise L.MyoloError('labels must be a contiguous [h,w] uint8/int64 tensor')
    h, w = labels.shape
    cm = _cmap(colormap, labels.device)
    mask = torch.empty(h, w, 3, dtype=torch.uint8, device=labels.device) if want_mask else None
    dst = None
    if im0 is not None:
        L.require_gpu(im0)
        if im0.dtype != torch.uint8 or tuple(im0.shape) != (h, w, 3) or not im0.is_contiguous():
            raise L.MyoloError('im0 must be a contiguous uint8 [h,w,3] tensor matching the label map')
        dst = torch.empty_like(im0)
    if mask is None and dst is None:
        raise L.MyoloError('nothing to compute: pass im0 or want_mask=True')
    L.check(L.lib().myolo_seg_blend(L.ptr(labels), L.DT[labels.dtype], L.ptr(im0), h, w, L.ptr(cm), cm.shape[0], int(bool(bgr)),
                                    C.c_float(alpha), C.c_float(beta), C.c_float(gamma), L.ptr(mask), L.ptr(dst), L.stream_ptr()),
            'myolo_seg_blend')
    return mask, dst


def label2image(pred, COLORMAP=Cityscapes_COLORMAP):
    """detect.py:69-72 on a device label map: RGB image [h,w,3] uint8."""
    return seg_overlay(pred, None, COLORMAP, bgr=False)[0]
