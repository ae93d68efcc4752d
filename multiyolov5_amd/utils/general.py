"""Box math / NMS entry points mirrored from the reference's utils/general.py (make_divisible 176-178,
xywh2xyxy 265-272, scale_coords 319-332, clip_coords 335-340, non_max_suppression 421-509)."""
import math

import torch


def make_divisible(x, divisor):
    return math.ceil(x / divisor) * divisor


def xywh2xyxy(x):
    y = x.clone()
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def clip_coords(boxes, img_shape):
    boxes[:, 0].clamp_(0, img_shape[1])
    boxes[:, 1].clamp_(0, img_shape[0])
    boxes[:, 2].clamp_(0, img_shape[1])
    boxes[:, 3].clamp_(0, img_shape[0])


def scale_coords(img1_shape, coords, img0_shape, ratio_pad=None):
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain = ratio_pad[0][0]
        pad = ratio_pad[1]
    coords[:, [0, 2]] -= pad[0]
    coords[:, [1, 3]] -= pad[1]
    coords[:, :4] /= gain
    clip_coords(coords, img0_shape)
    return coords


# ---------------------------------------------------------------------------------------------------------------
# inference post-processing on MI355X (libmyolo csrc/nms.hip, csrc/seg_out.hip); no CPU path
import ctypes as _C

from .. import _lib as _L


_NMS_SCRATCH = {}
_NMS_TINY = {}
_NMS_CACHE_MAX_BYTES = 64 << 20


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        labels=()):
    """general.py:421-509: list (one per image) of [n,6] tensors (x1, y1, x2, y2, conf, cls), descending conf, n <= 300.
    One batched launch sequence for all images; the only host synchronisation is reading the per-image keep counts.
    fp16 predictions: as in the reference, cls*obj, xywh->xyxy and the threshold compares are rounded to fp16, the rows come
    back as float32 (torch.cat with `j.float()` promotes them, general.py:470-473) and the IoU test runs in fp32."""
    if labels is not None and len(labels):
        raise NotImplementedError('labels= (auto-labelling priors, general.py:448-456) is not on the gfx950 hot path')
    class_mask = 0
    if classes is not None:                                       # general.py:476-477 (detect.py --classes)
        cl = [int(c) for c in (classes.tolist() if torch.is_tensor(classes) else classes)]
        if any(c < 0 or c >= 64 for c in cl) or prediction.shape[2] - 5 > 64:
            raise _L.MyoloError('classes= filter supports class ids 0..63')
        for c in cl:
            class_mask |= 1 << c
        if not cl:                                                # an empty filter keeps nothing
            return [torch.zeros((0, 6), device=prediction.device, dtype=torch.float32) for _ in range(prediction.shape[0])]
    _L.require_gpu(prediction)
    if prediction.dim() != 3 or prediction.dtype not in (torch.float16, torch.float32):
        raise _L.MyoloError('prediction must be a [B,A,5+nc] fp16/fp32 tensor')
    pred = prediction.contiguous()
    B, A, no = pred.shape
    nc = no - 5
    max_wh, max_det, max_nms = 4096.0, 300, 30000                 # general.py:433-435
    multi = bool(multi_label and nc > 1)                          # general.py:438
    cap = A * nc if multi else A
    dev = pred.device
    # scratch of the launch sequence: reused across calls with the same geometry on the same stream (stream order makes the reuse safe;
    # eight torch.empty per frame were ~25 us of host time on the detect.py path).  `out` -- the rows the caller keeps -- is fresh.
    use_ws = bool(multi or conf_thres < 0.05)
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, B, A, no, multi, use_ws)
    # only detect.py-sized sets are kept (<= 64 MB of candidate rows); test.py's conf 0.001 + multi_label workspaces are
    # gigabytes per geometry and rect validation changes A almost every batch -- those come from (and go back to) torch's caching allocator
    # (the bound counts EVERYTHING a cached set pins -- candidate rows, indices, the sorted rows, the counting-sort workspace and the
    #  bit-matrix workspace of the short-list path (~9 MB per image) -- not the candidate rows alone)
    mws_est = int(_L.lib().myolo_nms_ws_bytes(B, cap)) if (not use_ws and B <= 64) else 0
    set_bytes = B * cap * (6 + 1) * 4 + B * max_nms * 6 * 4 + (B * (3 * 65536 + cap) * 4 if use_ws else 0) + mws_est
    cacheable = set_bytes <= _NMS_CACHE_MAX_BYTES
    sc = _NMS_SCRATCH.get(key) if cacheable else None
    if sc is None:
        if len(_NMS_SCRATCH) > 16:
            _NMS_SCRATCH.clear()
        # the tiny host-side objects (pinned keep-count buffer, event, the two count vectors) are cached per (device, stream, B) whatever the
        # geometry: a pin_memory() per call is a synchronous hipHostMalloc on test.py's non-cacheable path
        tiny = _NMS_TINY.get(key[:3])
        if tiny is None:
            if len(_NMS_TINY) > 64:
                _NMS_TINY.clear()
            tiny = _NMS_TINY[key[:3]] = {'counts': torch.empty(B, dtype=torch.int32, device=dev), 'nkeep': torch.empty(B, dtype=torch.int32, device=dev),
                                         'host': torch.empty(B, dtype=torch.int32).pin_memory(), 'ev': torch.cuda.Event()}
        sc = {'counts': tiny['counts'], 'cand': torch.empty(B * cap * 6, dtype=torch.float32, device=dev),
              'cidx': torch.empty(B * cap, dtype=torch.int32, device=dev), 'srt': torch.empty(B * max_nms * 6, dtype=torch.float32, device=dev),
              'nkeep': tiny['nkeep'], 'host': tiny['host'],
              'ev': tiny['ev'],
              # long candidate lists (test.py: conf 0.001 + multi_label, 1e5 per image): counting sort instead of the O(n^2) rank kernel
              'ws': torch.empty(B * (3 * 65536 + cap), dtype=torch.int32, device=dev) if use_ws else None, 'mws': None, 'mws_bytes': 0}
        # short single-label lists (detect.py): rank / bit matrix on the whole device + a one-wave scan (csrc/nms.hip); 9 MB per image
        if sc['ws'] is None and B <= 64:
            sc['mws_bytes'] = mws_est
            sc['mws'] = torch.empty(sc['mws_bytes'], dtype=torch.uint8, device=dev)
        if cacheable:
            _NMS_SCRATCH[key] = sc
    counts, cand, cidx, srt, nkeep, ws, mws, mws_bytes = (sc[k] for k in ('counts', 'cand', 'cidx', 'srt', 'nkeep', 'ws', 'mws', 'mws_bytes'))
    out = torch.empty(B, max_det, 6, dtype=torch.float32, device=dev)
    _L.check(_L.lib().myolo_nms(_L.ptr(pred), _L.DT[pred.dtype], B, A, no, _C.c_float(conf_thres), _C.c_float(iou_thres),
                                int(multi), int(bool(agnostic)), _C.c_float(max_wh), max_nms, max_det, cap, _L.ptr(counts),
                                _L.ptr(cand), _L.ptr(cidx), _L.ptr(srt), _L.ptr(out), _L.ptr(nkeep), class_mask,
                                _L.ptr(ws) if ws is not None else None, _L.ptr(mws) if mws is not None else None, mws_bytes,
                                _L.stream_ptr()),
             'myolo_nms')
    # the one sync (the reference syncs per image, 446-495): the keep counts through pinned memory, the host POLLS the event instead of
    # sleeping in a blocking copy (the interrupt wake-up was ~40 us per frame, profiles/r3k_infer_timeline.md)
    host, ev = sc['host'], sc['ev']
    host.copy_(nkeep, non_blocking=True)
    ev.record()
    spins = 0
    while not ev.query():
        spins += 1
        if spins > 200000:                                        # (a long-running queue in front: stop burning the core)
            ev.synchronize()
            break
    n = host.tolist()
    return [out[i, :n[i]] for i in range(B)]


def seg_argmax(seg, h0=None, w0=None, out_dtype=torch.int64):
    """detect.py:191-193 fused: bilinear(align_corners=True) resize of the class logits to (h0, w0) + argmax over classes ->
    labels [N,h0,w0].  `seg` is the model's segmentation output [N,C,H,W]; when (h0,w0) == (H,W) the resize is taken
    straight from the head's low-res logits (the x8 upsample of yolo.py:163 is folded in, bit-identical)."""
    _L.require_gpu(seg)
    n, c, H, W = seg.shape
    h0, w0 = int(h0 or H), int(w0 or W)
    low = getattr(seg, '_myolo_low', None)
    holder = getattr(seg, '_myolo_holder', None)
    if holder is not None:
        holder.wait_branch()                                      # eval: the segmentation head may still be running on the plan's side stream
    if low is not None and (h0, w0) == (H, W):
        src = low                                                 # [N,h,w,C] view of the plan's NHWC buffer
    else:
        src = seg.permute(0, 2, 3, 1)
        if src.stride(3) != 1:
            src = src.contiguous()
    if src.dtype not in (torch.float16, torch.float32) or c > 32:
        raise _L.MyoloError('seg_argmax: fp16/fp32 logits with at most 32 classes')
    d = _L.Tensor(src.data_ptr(), src.shape[0], src.shape[1], src.shape[2], src.shape[3], src.stride(0), src.stride(1),
                  src.stride(2), _L.DT[src.dtype], 0)
    labels = torch.empty(n, h0, w0, dtype=out_dtype, device=seg.device)
    _L.check(_L.lib().myolo_seg_argmax(_C.byref(d), _L.ptr(labels), _L.DT[out_dtype], h0, w0, _L.stream_ptr()),
             'myolo_seg_argmax')
    return labels
