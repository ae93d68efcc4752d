"""-m gpu: inference post-processing through the reference's API (utils.general.non_max_suppression) and the fused
resize+argmax, against the golden vectors produced by the reference's own non_max_suppression (tests/golden/nms.npz, greedy
kernel restated -- torchvision is absent, oracle/nms_ref.py) and against the CPU oracle at full size."""
import os

import numpy as np
import pytest
import torch

from oracle import nms_ref, synth
from tests.util import CFG, TAGS, golden, synth_sd

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _same(got, ref, name):
    assert got.shape == ref.shape, f'{name}: kept {got.shape[0]} rows, reference {ref.shape[0]}'
    # box / class indices bit-exact: identical rows in identical order
    np.testing.assert_array_equal(got[:, 5], ref[:, 5], err_msg=name + ' classes')
    np.testing.assert_array_equal(got, ref, err_msg=name)


NMS_MODES = {'single': dict(conf_thres=0.25, iou_thres=0.45), 'multi': dict(conf_thres=0.001, iou_thres=0.6, multi_label=True),
             'classes': dict(conf_thres=0.25, iou_thres=0.45, classes=[2, 5, 7]), 'agnostic': dict(conf_thres=0.25, iou_thres=0.45, agnostic=True),
             'multi_classes': dict(conf_thres=0.05, iou_thres=0.6, multi_label=True, classes=[0, 9])}


@pytest.mark.parametrize('mode', list(NMS_MODES))
def test_nms_matches_reference_golden(mode):
    from multiyolov5_amd.utils.general import non_max_suppression
    g = golden('nms')
    pred = synth.synth_nms_pred(2, 3000, 10, seed=3).to(DEV)
    kw = NMS_MODES[mode]
    out = non_max_suppression(pred, **kw)
    assert len(out) == 2
    for i, o in enumerate(out):
        _same(o.cpu().numpy(), g[f'{mode}_{i}'], f'nms/{mode}/{i}')


@pytest.mark.parametrize('A,wh', [(32256, (1024, 512)), (129024, (2048, 1024))])
def test_nms_full_size_vs_oracle(A, wh):
    from multiyolov5_amd.utils.general import non_max_suppression
    pred = synth.synth_nms_pred(2, A, 10, seed=5, img_w=wh[0], img_h=wh[1])
    ref = nms_ref.non_max_suppression(pred.numpy(), 0.25, 0.45)
    out = non_max_suppression(pred.to(DEV), 0.25, 0.45)
    for i in range(2):
        _same(out[i].cpu().numpy(), ref[i], f'nms/full{A}/{i}')
        assert out[i].shape[0] <= 300 and (np.diff(out[i][:, 4].cpu().numpy()) <= 0).all()
    # idempotence: feeding the survivors back (as xywh + obj=1) keeps all of them
    o = out[0].float()
    x = torch.zeros(1, o.shape[0], 15, device=DEV)
    x[0, :, 0], x[0, :, 1] = (o[:, 0] + o[:, 2]) / 2, (o[:, 1] + o[:, 3]) / 2
    x[0, :, 2], x[0, :, 3] = o[:, 2] - o[:, 0], o[:, 3] - o[:, 1]
    x[0, :, 4] = 1.0
    x[0, torch.arange(o.shape[0]), 5 + o[:, 5].long()] = o[:, 4]
    again = non_max_suppression(x, 0.25, 0.45)[0]
    assert again.shape[0] == o.shape[0]


@pytest.mark.parametrize('mode,kw', [('single_f16', dict(conf_thres=0.25, iou_thres=0.45)),
                                     ('multi_f16', dict(conf_thres=0.001, iou_thres=0.6, multi_label=True))])
def test_nms_fp16_predictions_match_reference_golden(mode, kw):
    """detect.py --half (BASELINE config 5): the reference's non_max_suppression run on fp16 predictions (CPU) -- cls*obj, xywh2xyxy and
    the threshold compares in fp16, float32 rows out; kept rows bit-exact"""
    from multiyolov5_amd.utils.general import non_max_suppression
    g = golden('nms')
    pred = synth.synth_nms_pred(2, 3000, 10, seed=3).half().to(DEV)
    out = non_max_suppression(pred, **kw)
    for i, o in enumerate(out):
        assert o.dtype == torch.float32
        np.testing.assert_array_equal(o.cpu().numpy(), g[f'{mode}_{i}'])


def test_nms_edge_cases():
    from multiyolov5_amd.utils.general import non_max_suppression
    # nothing above the threshold -> empty [0,6]; fp16 input returns fp16 rows
    pred = torch.zeros(3, 500, 15, device=DEV)
    out = non_max_suppression(pred)
    assert [tuple(o.shape) for o in out] == [(0, 6)] * 3
    p16 = synth.synth_nms_pred(1, 4000, 10, seed=8).to(DEV).half()
    out16 = non_max_suppression(p16, 0.25, 0.45)[0]
    assert out16.dtype == torch.float32                       # the reference's rows are float32 for fp16 input too (torch.cat with j.float())
    # more than max_det survivors: exactly 300 rows, the 300 best
    rs = np.random.RandomState(0)
    far = np.zeros((1, 1000, 15), np.float32)
    far[0, :, 0] = np.arange(1000) * 60 % 2000 + 10
    far[0, :, 1] = np.arange(1000) // 33 * 30 + 10
    far[0, :, 2:4] = 8
    far[0, :, 4] = 0.9
    far[0, :, 5] = rs.uniform(0.5, 1.0, 1000)
    o = non_max_suppression(torch.from_numpy(far).to(DEV), 0.25, 0.45)[0]
    r = nms_ref.non_max_suppression(far, 0.25, 0.45)[0]
    assert o.shape[0] == 300
    np.testing.assert_array_equal(o.cpu().numpy(), r)


@pytest.mark.parametrize('case', ['agnostic', 'nc80', 'wide_span', 'one_class_many_kept', 'classes_filter'])
def test_nms_per_class_segments_match_the_oracle(case):
    """round 3: short lists are suppressed class by class (class on top of the sort key, one scan wave per class, merge by score).
    Against the oracle (global greedy walk in score order, like torchvision's): agnostic mode (one segment), 80 classes (many short
    segments), coordinates spanning more than max_wh (classes are NOT independent any more: the image must go to the lazy scan),
    one class keeping more than max_det boxes while others keep a few, and the `classes=` filter"""
    from multiyolov5_amd.utils.general import non_max_suppression
    kw = dict(conf_thres=0.25, iou_thres=0.45)
    if case == 'nc80':
        pred = synth.synth_nms_pred(2, 6000, 80, seed=11, img_w=1024, img_h=512)
    elif case == 'wide_span':
        pred = synth.synth_nms_pred(2, 5000, 10, seed=12, img_w=1024, img_h=512)
        pred[0, ::7, 0] += 5000.0                                # boxes of image 0 spread over > max_wh (4096) pixels; image 1 stays narrow
        pred[0, 1::7, 2] = 4500.0                                # and some wider than max_wh: they overlap across the class offsets
    elif case == 'one_class_many_kept':
        rs = np.random.RandomState(1)
        a = np.zeros((1, 3000, 15), np.float32)
        a[0, :, 0] = np.arange(3000) * 37 % 2000 + 10
        a[0, :, 1] = np.arange(3000) // 54 * 18 + 10
        a[0, :, 2:4] = 6
        a[0, :, 4] = 0.95
        cls = np.where(np.arange(3000) % 5 == 0, rs.randint(1, 10, 3000), 0)      # 80 % class 0: > 300 disjoint boxes of one class
        a[0, np.arange(3000), 5 + cls] = rs.uniform(0.4, 1.0, 3000)
        pred = torch.from_numpy(a)
    else:
        pred = synth.synth_nms_pred(2, 6000, 10, seed=13, img_w=1024, img_h=512)
    if case == 'agnostic':
        kw['agnostic'] = True
    if case == 'classes_filter':
        kw['classes'] = [1, 3, 8]
    for dt in (torch.float32, torch.float16):                 # fp16: detect.py --half (config 5 feeds NMS fp16 predictions)
        half = dt == torch.float16
        p = pred.half() if half else pred
        ref = nms_ref.non_max_suppression(p.numpy(), half=half, **kw)      # (half=True is pinned to the reference's own fp16 run)
        out = non_max_suppression(p.to(DEV), **kw)
        for i in range(pred.shape[0]):
            assert out[i].dtype == torch.float32
            _same(out[i].cpu().numpy(), ref[i], f'nms/{case}/{dt}/{i}')
            assert out[i].shape[0] <= 300
        if case == 'one_class_many_kept':
            assert out[0].shape[0] == 300


def test_seg_argmax_fused_matches_oracle_and_model_output():
    from multiyolov5_amd.models.yolo import Model
    from multiyolov5_amd.utils.general import seg_argmax
    m = Model(os.path.join(CFG, TAGS['s_psp']))
    m.load_state_dict(synth_sd('s_psp'), strict=True)
    m = m.to(DEV).fuse().eval()
    x = synth.synth_images(2, 64, 128, seed=1)[:1].to(DEV)
    with torch.no_grad():
        (pred, raw), seg = m(x)
    g = golden('model_s_psp')
    lab = seg_argmax(seg)                                   # fused from the low-res logits (no second resize)
    assert lab.dtype == torch.int64 and tuple(lab.shape) == (1, 64, 128)
    # vs the reference's own argmax: identical, except where the ORACLE's top-2 logits are a rounding-noise tie (north_star: bit-exact)
    from oracle import model_ref
    from tests.test_gpu_model import assert_argmax_exact_or_near_tie
    from tests.util import load_cfg
    fsd = model_ref.fuse_state_dict({k: v.clone() for k, v in synth_sd('s_psp').items()})
    with torch.no_grad():
        _, rseg = model_ref.forward(load_cfg('s_psp'), fsd, x.cpu(), training=False)
    assert_argmax_exact_or_near_tie('postproc/seg_argmax', lab.cpu(), torch.from_numpy(g['eval_seg_argmax']).long(), rseg, eps=1e-4)
    np.testing.assert_array_equal(lab.cpu().numpy(), seg.argmax(1).cpu().numpy())   # bit-identical to the materialised logits
    # detect.py:191 resize to a different original size: second-stage bilinear of the full-res logits + argmax, against the oracle's
    # resize of the SAME logits (near-ties judged in the oracle's resized logits)
    lab2 = seg_argmax(seg, 100, 180, out_dtype=torch.uint8)
    segc = seg[0].float().cpu()
    ref2 = nms_ref.seg_argmax(segc.numpy(), 100, 180)
    r2 = torch.from_numpy(nms_ref.bilinear_ac(segc.numpy(), 100, 180))[None]
    assert_argmax_exact_or_near_tie('postproc/seg_argmax_resized', lab2[0].long().cpu(), torch.from_numpy(ref2).long(), r2, eps=1e-5)


class _AtenLog(torch.utils._python_dispatch.TorchDispatchMode):
    """every ATen op that reaches the dispatcher while the mode is active"""

    def __init__(self):
        super().__init__()
        self.ops = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        self.ops.append(str(func))
        return func(*args, **(kwargs or {}))


@pytest.mark.parametrize('half', [False, True], ids=['f32', 'f16'])
def test_unchanged_detect_py_and_test_py_statements_run_the_fused_argmax(half, monkeypatch):
    """detect.py:144-149,191-193 and test.py:34-40 + utils/metrics.py:240,259 AS THE REFERENCE WRITES THEM, under
    dropin.install(): the resize + arg-max must be the library's fused launch (myolo_seg_argmax in the launch trace, no ATen
    upsample_bilinear2d / max in the dispatcher log) and the indices must be the reference's"""
    import torch.nn.functional as F
    from multiyolov5_amd import _lib as L, dropin, runtime as R
    from oracle import metrics_ref
    from tests.test_gpu_model import assert_argmax_exact_or_near_tie
    with dropin.installed():
        from models.yolo import Model                              # the reference's import path (train.py:20, detect.py:34)
        from multiyolov5_amd.utils.general import non_max_suppression
        model = Model(os.path.join(CFG, TAGS['s_psp']))
        model.load_state_dict(synth_sd('s_psp'), strict=True)
        model = model.to(DEV)
        if half:
            model.half()
        model.fuse().eval()
        img = synth.synth_images(2, 64, 128, seed=1).to(DEV)
        img = img.half() if half else img

        def trace_names():
            tr = L.launch_trace()
            L.lib().myolo_trace_start(0)
            return [n for n, c in tr.items() for _ in range(c) if 'seg_argmax' in n or 'seg_up' in n]

        # ---- detect.py:144-149 + 191-193 (im0 = a 100 x 180 source frame, then one of the network's own size)
        for (h0, w0) in ((100, 180), (64, 128)):
            im0 = np.zeros((h0, w0, 3), np.uint8)
            with torch.no_grad():
                out = model(img[:1], augment=False)
                pred = out[0][0]
                seg = out[1]
                pred = non_max_suppression(pred, 0.25, 0.45, classes=None, agnostic=False)
                L.lib().myolo_trace_start(1)
                with _AtenLog() as log:
                    seg = F.interpolate(seg, (im0.shape[0], im0.shape[1]), mode='bilinear', align_corners=True)[0]
                    mask = seg.max(axis=0)[1].cpu().numpy()
                names = trace_names()
            assert mask.shape == (h0, w0) and mask.dtype == np.int64
            assert not any('upsample' in o or 'aten.max' in o or 'argmax' in o for o in log.ops), log.ops
            assert sum('seg_argmax' in n for n in names) == 1, names
            # the ATen route over the same logits (round 5's behaviour) and the oracle's resize + argmax
            monkeypatch.setattr(R, 'LAZY_RESIZE', False)
            with torch.no_grad():
                seg2 = model(img[:1], augment=False)[1]
                full = F.interpolate(seg2, (h0, w0), mode='bilinear', align_corners=True)[0]
                ref_aten = full.max(axis=0)[1].cpu().numpy()
                logits = seg2[0].float().cpu().numpy()
            monkeypatch.setattr(R, 'LAZY_RESIZE', True)
            ref = nms_ref.seg_argmax(logits, h0, w0)
            r2 = torch.from_numpy(nms_ref.bilinear_ac(logits, h0, w0))[None]
            assert_argmax_exact_or_near_tie(f'unchanged/detect_{h0}x{w0}', torch.from_numpy(mask)[None], torch.from_numpy(ref).long()[None], r2,
                                            eps=1e-5 if not half else 2e-3)
            assert (mask != ref_aten).mean() <= (0 if (h0, w0) == (64, 128) else 2e-3)

        # ---- test.py:34-40 + metrics.py:240-249, 259-275 on a batch of two; target at twice the network's resolution
        target = synth.synth_seg_targets(2, 128, 256, 19, seed=5)
        with torch.no_grad():
            outputs = model(img)
            pred = outputs[1]
            target = target.to(DEV, non_blocking=True)
            L.lib().myolo_trace_start(1)
            with _AtenLog() as log:
                pred = F.interpolate(pred, (target.shape[1], target.shape[2]), mode='bilinear', align_corners=True)
                _, predict = torch.max(pred.data, 1)                               # batch_pix_accuracy
                predict = predict.cpu().numpy().astype('int64') + 1
                _, predict2 = torch.max(pred.data, 1)                              # batch_intersection_union
                predict2 = predict2.cpu().numpy().astype('int64') + 1
            names = trace_names()
        assert not any('upsample' in o or 'aten.max' in o or 'argmax' in o for o in log.ops), log.ops
        assert sum('seg_argmax' in n for n in names) == 1, names                  # the second torch.max reuses the label map
        np.testing.assert_array_equal(predict, predict2)
        monkeypatch.setattr(R, 'LAZY_RESIZE', False)
        with torch.no_grad():
            p2 = F.interpolate(model(img)[1], (128, 256), mode='bilinear', align_corners=True)
            ref_pred = torch.max(p2.data, 1)[1].cpu().numpy().astype('int64') + 1
            rc, rl = metrics_ref.batch_pix_accuracy(p2.float().cpu().numpy(), target.cpu().numpy())
        monkeypatch.setattr(R, 'LAZY_RESIZE', True)
        assert (predict != ref_pred).mean() <= 2e-3
        t = target.cpu().numpy().astype('int64') + 1
        correct, labeled = np.sum((predict == t) * (t > 0)), np.sum(t > 0)
        assert labeled == rl and abs(int(correct) - int(rc)) <= 2e-3 * predict.size
        # the mirror's own counters take the cached label map of the view
        from multiyolov5_amd.utils.metrics import batch_pix_accuracy
        c3, l3 = batch_pix_accuracy(pred.data, target)
        assert (int(c3), int(l3)) == (int(correct), int(labeled))


def test_seg_metrics_match_reference_golden_and_oracle():
    """test.py:31-65 counters (batch_pix_accuracy / batch_intersection_union) on the device: bit-exact integers."""
    from multiyolov5_amd.utils.metrics import batch_intersection_union, batch_pix_accuracy
    from oracle import metrics_ref
    g = golden('metrics')
    logits = torch.from_numpy(g['logits']).to(DEV)
    mask = torch.from_numpy(g['mask'].astype(np.int64)).to(DEV)
    c, l = batch_pix_accuracy(logits, mask)
    i, u = batch_intersection_union(logits, mask, 19)
    assert int(c) == int(g['correct']) and int(l) == int(g['labeled'])
    np.testing.assert_array_equal(i, g['inter'])
    np.testing.assert_array_equal(u, g['union'])
    # full-size labels [2,1024,2048], all-ignored image, class 18 present
    rs = np.random.RandomState(3)
    x = torch.from_numpy(rs.normal(0, 1, (2, 19, 128, 256)).astype(np.float32))
    m = synth.synth_seg_targets(2, 128, 256, 19, seed=9)
    m[1] = -1
    c2, l2 = batch_pix_accuracy(x.to(DEV), m.to(DEV))
    i2, u2 = batch_intersection_union(x.to(DEV), m.to(DEV), 19)
    rc, rl = metrics_ref.batch_pix_accuracy(x.numpy(), m.numpy())
    ri, ru = metrics_ref.batch_intersection_union(x.numpy(), m.numpy(), 19)
    assert (int(c2), int(l2)) == (int(rc), int(rl))
    np.testing.assert_array_equal(i2, ri)
    np.testing.assert_array_equal(u2, ru)


@pytest.mark.parametrize('half', [True, False], ids=['f16', 'f32'])
@pytest.mark.parametrize('shape,new_shape', [((1024, 2048), 2048), ((1000, 2048), 2048), ((37, 64), 64), ((64, 50), 64),
                                             # frames letterbox resamples (cv2.resize INTER_LINEAR, datasets.py:843-844):
                                             ((720, 1280), 640), ((1080, 1920), 1024), ((1024, 2048), 1024), ((240, 320), 640),
                                             ((333, 500), 416), ((97, 61), 128)])
def test_frame_to_input_matches_reference_steps(shape, new_shape, half):
    """letterbox border + BGR->RGB + HWC->CHW + /255 (datasets.py:818-848,185; detect.py:135-139) in one kernel: bit-identical to
    the reference's own numpy/torch steps (oracle.frame_ref), incl. odd padding splits"""
    from multiyolov5_amd.utils.datasets import frame_to_input
    from oracle import frame_ref
    rs = np.random.RandomState(shape[0])
    im0 = rs.randint(0, 256, (shape[0], shape[1], 3)).astype(np.uint8)
    ref, rratio, rpad = frame_ref.frame_to_input(im0, new_shape, stride=32, half=half)
    got, ratio, pad = frame_to_input(torch.from_numpy(im0).to(DEV), new_shape, stride=32, half=half)
    assert ratio == rratio and tuple(pad) == tuple(rpad) and got.shape == ref.shape and got.dtype == ref.dtype
    if half:
        assert torch.equal(got.cpu(), ref)
    else:
        # fp32 `img /= 255.0`: torch's GPU kernel multiplies by the reciprocal, its CPU kernel divides -- the last bit can differ.  The
        # product takes its 256 values from torch on the device (what detect.py computes on a GPU); vs the CPU oracle: <= 1 ulp, and
        # exactly the device's own `arange(256)/255` table
        assert float((got.cpu() - ref).abs().max()) <= 2.0 ** -24
        lut = torch.arange(256, device=DEV, dtype=torch.uint8).float()
        lut /= 255.0
        from oracle import frame_ref as fr
        boxed, _, _ = fr.letterbox(im0, new_shape, stride=32)
        idx = torch.from_numpy(np.ascontiguousarray(boxed[:, :, ::-1].transpose(2, 0, 1))).long().to(DEV)
        assert torch.equal(got[0], lut[idx])


def test_frame_to_input_resamples_like_the_8bit_fixed_point_restatement():
    """the resampling branch is really taken (1280x720 -> 640x360 = the exact-2x box average; 1920x1080 -> 1024x576 general bilinear)
    and is bit-identical, pixel for pixel, to oracle.frame_ref.cv_resize_linear_u8"""
    from multiyolov5_amd.utils.datasets import frame_to_input
    from oracle import frame_ref
    for (h, w), ns in (((720, 1280), 640), ((1080, 1920), 1024)):
        im0 = np.random.RandomState(h).randint(0, 256, (h, w, 3)).astype(np.uint8)
        got, ratio, pad = frame_to_input(torch.from_numpy(im0).to(DEV), ns, stride=32, half=True, auto=False)
        assert got.shape == (1, 3, ns, ns) and ratio[0] == ns / w
        rh, rw = int(round(h * ratio[1])), int(round(w * ratio[0]))
        res = frame_ref.cv_resize_linear_u8(im0, (rw, rh))
        top = int(round(pad[1] - 0.1))
        inner = (got[0, :, top:top + rh, :rw] * 255).round().to(torch.uint8).cpu().numpy()       # RGB planes back to bytes
        np.testing.assert_array_equal(inner, res[:, :, ::-1].transpose(2, 0, 1))
        assert float(got[0, :, 0, 0].float().mean()) == pytest.approx(114 / 255, abs=1e-3)       # border rows


@pytest.mark.parametrize('ldt', [torch.uint8, torch.int64], ids=['u8', 'i64'])
def test_seg_overlay_matches_reference_steps(ldt):
    """label2image + [:, :, ::-1] + addWeighted(mask, 0.4, im0, 0.6, 0) (detect.py:69-72,193-194): bit-identical bytes"""
    from multiyolov5_amd.utils.plots import Cityscapes_COLORMAP, label2image, seg_overlay
    from oracle import frame_ref
    rs = np.random.RandomState(5)
    h, w = 203, 517
    labels = rs.randint(0, 19, (h, w))
    im0 = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    rmask, rdst = frame_ref.seg_overlay(labels, im0, Cityscapes_COLORMAP)
    mask, dst = seg_overlay(torch.from_numpy(labels).to(DEV, ldt), torch.from_numpy(im0).to(DEV))
    np.testing.assert_array_equal(mask.cpu().numpy(), rmask)
    np.testing.assert_array_equal(dst.cpu().numpy(), rdst)
    rgb = label2image(torch.from_numpy(labels).to(DEV, ldt))
    np.testing.assert_array_equal(rgb.cpu().numpy(), rmask[:, :, ::-1])


def test_match_predictions_matches_reference_golden():
    """test.py:230-262 (true-positive matrix per image) in one launch: bit-identical to the golden built around the reference's own
    box_iou, incl. duplicate detections competing for one target, class-confused detections and classes without targets"""
    from multiyolov5_amd.utils.metrics import match_predictions
    g = golden('match')
    iouv = torch.from_numpy(g['iouv']).to(DEV)
    for i in range(3):
        p, l = torch.from_numpy(g[f'pred_{i}']).to(DEV), torch.from_numpy(g[f'labels_{i}']).to(DEV)
        c = match_predictions(p, l, iouv)
        assert c.dtype == torch.bool and tuple(c.shape) == (p.shape[0], 10)
        np.testing.assert_array_equal(c.cpu().numpy(), g[f'correct_{i}'], err_msg=f'match/{i}')
    # no predictions / no labels: all-false matrices of the right shape
    assert tuple(match_predictions(torch.zeros(0, 6, device=DEV), l, iouv).shape) == (0, 10)
    assert not match_predictions(p, torch.zeros(0, 5, device=DEV), iouv).any()
