"""Segmentation evaluation counters on MI355X -- host-side mirror of the reference's utils/metrics.py:234-275
(`batch_pix_accuracy`, `batch_intersection_union`), the two functions test.py:31-65 `seg_validation` calls per batch.

The reference arg-maxes the logits on the device, copies both label maps to the host and histograms them with numpy; here the
resize + arg-max is one kernel (myolo_seg_argmax, never materialising resized logits) and the counters are one kernel
(myolo_seg_metrics); only 2+3*nclass integers cross PCIe.  Same return values (numpy int64, as np.sum / np.histogram give)."""
import ctypes as C

import numpy as np
import torch

from .. import _lib as L
from .general import seg_argmax


def _labels(output):
    from ..runtime import LazyResized
    if isinstance(output, LazyResized):         # test.py:38's resized prediction: the fused resize + arg-max, cached on the view
        return output.argmax(1)
    if output.dim() == 4:                       # logits [N,C,H,W]: `_, predict = torch.max(output, 1)` (metrics.py:240,259)
        return seg_argmax(output, out_dtype=torch.uint8)
    L.require_gpu(output)
    if output.dtype not in (torch.uint8, torch.int64):
        raise L.MyoloError('labels must be uint8 or int64')
    return output.contiguous()


def seg_counts(output, target, nclass):
    """(pixel_correct, pixel_labeled, area_inter[nclass], area_union[nclass]) in one pass."""
    pred = _labels(output)
    L.require_gpu(target)
    target = target.to(torch.int64).contiguous()
    if pred.shape != target.shape:
        raise L.MyoloError(f'prediction {tuple(pred.shape)} and target {tuple(target.shape)} differ')
    counts = torch.empty(2 + 3 * nclass, dtype=torch.int64, device=pred.device)
    L.check(L.lib().myolo_seg_metrics(L.ptr(pred), L.DT[pred.dtype], L.ptr(target), pred.numel(), int(nclass), L.ptr(counts),
                                      L.stream_ptr()), 'myolo_seg_metrics')
    c = counts.cpu().numpy()
    inter, predc, lab = c[2:2 + nclass], c[2 + nclass:2 + 2 * nclass], c[2 + 2 * nclass:]
    return c[0], c[1], inter, predc + lab - inter


def batch_pix_accuracy(output, target):          # metrics.py:234-249
    correct, labeled, _, _ = seg_counts(output, target, 1 if output.dim() != 4 else output.shape[1])
    assert correct <= labeled, 'Correct area should be smaller than Labeled'
    return correct, labeled


def batch_intersection_union(output, target, nclass):   # metrics.py:252-275
    _, _, inter, union = seg_counts(output, target, nclass)
    assert (inter <= union).all(), 'Intersection area should be smaller than Union area'
    return inter, union


def match_predictions(predn, labels, iouv):
    """test.py:230-262 for one image on the device: predn [n,6] = (xyxy, conf, cls) in native image space and NMS order, labels [m,5] =
    (cls, xyxy) native; iouv [niou] (test.py:98).  Returns `correct` bool [n, niou] (the per-image input of `ap_per_class`, test.py:
    265,270).  The reference walks the candidates with one `.item()` per detection; here it is one launch without a sync."""
    L.require_gpu(predn)
    n, m = int(predn.shape[0]), int(labels.shape[0])
    iouv = iouv.to(device=predn.device, dtype=torch.float32).contiguous()
    niou = int(iouv.numel())
    correct = torch.zeros(n, niou, dtype=torch.uint8, device=predn.device)
    if n == 0 or m == 0:
        return correct.bool()
    if predn.shape[1] != 6 or labels.shape[1] != 5:
        raise L.MyoloError('predn must be [n,6] (xyxy, conf, cls) and labels [m,5] (cls, xyxy)')
    p = predn.to(torch.float32).contiguous()
    l = labels.to(device=predn.device, dtype=torch.float32).contiguous()
    ws = torch.empty(n * 8 + (m + 15) // 16 * 16, dtype=torch.uint8, device=predn.device)
    L.check(L.lib().myolo_match_predictions(L.ptr(p), n, L.ptr(l), m, L.ptr(iouv), niou, L.ptr(correct), L.ptr(ws), ws.numel(),
                                            L.stream_ptr()), 'myolo_match_predictions')
    return correct.bool()

