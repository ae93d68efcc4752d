"""CPU restatement of inference post-processing (TEST INFRASTRUCTURE ONLY).

* `greedy_nms`  -- torchvision.ops.nms (not in /root/reference: requirements.txt:11 `torchvision>=0.8.1`,
  call site utils/general.py:493).  Published algorithm: visit boxes by descending score; keep a box iff no
  previously kept box has IoU > thr with it (strict '>', IoU = inter/(a+b-inter), no +1); return kept indices
  in descending-score order.  PARITY UNPINNED: the reference has no test vector for it.
* `non_max_suppression` -- utils/general.py:421-509 (single-label and multi_label paths) around it.
* `seg_argmax` -- detect.py:191-193 (bilinear align_corners resize to (H0,W0), argmax over classes).
"""
import numpy as np


def greedy_nms(boxes, scores, thr):
    boxes = np.asarray(boxes, np.float32)
    order = np.argsort(-np.asarray(scores, np.float32), kind='stable')
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    keep, suppressed = [], np.zeros(len(boxes), bool)
    for idx in order:
        if suppressed[idx]:
            continue
        keep.append(idx)
        xx1 = np.maximum(boxes[idx, 0], boxes[:, 0]); yy1 = np.maximum(boxes[idx, 1], boxes[:, 1])
        xx2 = np.minimum(boxes[idx, 2], boxes[:, 2]); yy2 = np.minimum(boxes[idx, 3], boxes[:, 3])
        inter = np.maximum(xx2 - xx1, 0).astype(np.float32) * np.maximum(yy2 - yy1, 0).astype(np.float32)
        iou = inter / (area[idx] + area - inter)
        suppressed |= iou > thr
    return np.asarray(keep, np.int64)


def non_max_suppression(pred, conf_thres=0.25, iou_thres=0.45, multi_label=False, max_wh=4096, max_det=300,
                        max_nms=30000, classes=None, agnostic=False, half=False):
    """pred: [B,A,5+nc] (xywh, obj, cls).  Returns list of [n,6] (xyxy, conf, cls) float32.
    half=True: detect.py --half -- the reference runs general.py:421-509 on an fp16 prediction tensor: the threshold compares, cls*obj
    (462) and xywh2xyxy (265-272) are fp16 arithmetic (every operation rounded to fp16, as torch's CPU half kernels and numpy's float16
    do), `torch.cat((box, conf, j.float()))` (473) promotes the rows to float32 and everything behind it (class offsets, torchvision
    nms) is float32.  Pinned by the goldens the reference wrote on pred.half() (tests/test_oracle_golden.py)."""
    ft = np.float16 if half else np.float32
    pred = np.asarray(pred, ft)
    thr = ft(conf_thres)
    nc = pred.shape[2] - 5
    multi_label = multi_label and nc > 1
    out = []
    for x in pred:
        x = x[x[:, 4] > thr]                                          # general.py:430,446 (obj threshold)
        if not len(x):
            out.append(np.zeros((0, 6), np.float32)); continue
        x = x.copy()
        x[:, 5:] *= x[:, 4:5]                                         # 462
        two = ft(2)
        box = np.stack((x[:, 0] - x[:, 2] / two, x[:, 1] - x[:, 3] / two,
                        x[:, 0] + x[:, 2] / two, x[:, 1] + x[:, 3] / two), 1).astype(ft)   # 265-272
        if multi_label:                                               # 468-470
            i, j = np.nonzero(x[:, 5:] > thr)
            x = np.concatenate((box[i].astype(np.float32), x[i, j + 5, None].astype(np.float32), j[:, None].astype(np.float32)), 1)
        else:                                                         # 472-473
            j = x[:, 5:].argmax(1)
            conf = x[np.arange(len(x)), 5 + j]
            x = np.concatenate((box.astype(np.float32), conf[:, None].astype(np.float32), j[:, None].astype(np.float32)), 1)[conf > thr]
        if classes is not None:                                       # 476-477 class filter
            x = x[np.isin(x[:, 5].astype(np.int64), np.asarray(classes, np.int64))]
        if not len(x):
            out.append(np.zeros((0, 6), np.float32)); continue
        if len(x) > max_nms:                                          # 487-488
            x = x[np.argsort(-x[:, 4], kind='stable')[:max_nms]]
        c = x[:, 5:6] * (0 if agnostic else max_wh)                   # 491-492 class offset
        keep = greedy_nms(x[:, :4] + c, x[:, 4], iou_thres)[:max_det]  # 493-495
        out.append(x[keep].astype(np.float32))
    return out


def bilinear_ac(x, oh, ow):
    """F.interpolate(mode='bilinear', align_corners=True) on [C,H,W] float32 (ATen upsample_bilinear2d semantics:
    scale=(in-1)/(out-1), src=scale*dst, lambda from floor, neighbour index clamped)."""
    c, h, w = x.shape
    sy = (h - 1) / (oh - 1) if oh > 1 else 0.0
    sx = (w - 1) / (ow - 1) if ow > 1 else 0.0
    fy = (np.arange(oh, dtype=np.float32) * np.float32(sy)).astype(np.float32)
    fx = (np.arange(ow, dtype=np.float32) * np.float32(sx)).astype(np.float32)
    y0 = np.floor(fy).astype(np.int64); x0 = np.floor(fx).astype(np.int64)
    y1 = np.minimum(y0 + 1, h - 1); x1 = np.minimum(x0 + 1, w - 1)
    ly = (fy - y0).astype(np.float32)[None, :, None]; lx = (fx - x0).astype(np.float32)[None, None, :]
    a = x[:, y0][:, :, x0]; b = x[:, y0][:, :, x1]; cc = x[:, y1][:, :, x0]; d = x[:, y1][:, :, x1]
    return ((1 - ly) * ((1 - lx) * a + lx * b) + ly * ((1 - lx) * cc + lx * d)).astype(np.float32)


def seg_argmax(low, oh, ow):
    """low: [C,h,w] logits -> [oh,ow] int64 labels (first max wins, as torch.max)."""
    return bilinear_ac(np.asarray(low, np.float32), oh, ow).argmax(0)
