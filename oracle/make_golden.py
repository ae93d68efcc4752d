"""Generate tests/golden/*.npz by running the REAL reference (/root/reference) on CPU.

Build-container only (the reference does not travel to the GPU box); the vectors it writes are committed.
    python -m oracle.make_golden            # regenerate everything
Every case: weights = oracle.synth.synth_state_dict(reference state_dict shapes, seed 0), inputs from
oracle.synth (numpy RandomState streams), fp32, torch CPU.
"""
import os
import sys

import numpy as np
import torch
import yaml

from . import loss_ref, ref_shim, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
CFG = os.path.join(ROOT, 'multiyolov5_amd', 'cfg')
H, W = 64, 128     # golden image size (stride-32 multiple; P5 map 2x4)


def tap(t, k=6):
    """small fingerprint of a tensor: mean, std, abs-max and k strided samples."""
    f = t.detach().float().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, k).long()
    return np.concatenate(([f.mean().item(), f.std().item() if f.numel() > 1 else 0.0, f.abs().max().item()],
                           f[idx].numpy())).astype(np.float32)


def build_ref(ref, cfg_name):
    torch.manual_seed(0)
    m = ref.yolo.Model(os.path.join(CFG, cfg_name))
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    sd = synth.synth_state_dict(sd0, seed=0)
    m.load_state_dict(sd, strict=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0                      # train-mode parity is pinned with dropout off
    return m, sd0


def model_case(ref, cfg_name, tag, with_backward):
    m, sd_init = build_ref(ref, cfg_name)
    out = {}
    # state right after construction (stride probe side effects, yolo.py:261): pinned for the drop-in ctor
    bn0 = [k for k in sd_init if k.endswith('running_var')][0]
    out['init_running_var0'] = sd_init[bn0][:4].numpy()
    out['init_nbt0'] = np.array(sd_init[bn0.replace('running_var', 'num_batches_tracked')].item())
    out['init_anchors'] = sd_init[[k for k in sd_init if k.endswith('.anchors')][0]].numpy()
    out['n_params'] = np.array(sum(p.numel() for p in m.parameters()))

    x = synth.synth_images(2, H, W, seed=1)
    # ---- train-mode forward (batch-stat BN) ----
    m.train()
    feats = {}
    hooks = [mod.register_forward_hook(lambda mod_, i_, o_, idx=i: feats.__setitem__(idx, o_))
             for i, mod in enumerate(m.model)]
    det, seg = m(x)
    for h in hooks:
        h.remove()
    for i, o in feats.items():
        if torch.is_tensor(o):
            out[f'train_layer{i}'] = tap(o)
    for i, d in enumerate(det):
        out[f'train_det{i}'] = d.detach().numpy()
    segs = seg if isinstance(seg, list) else [seg]
    for j, s in enumerate(segs):
        out[f'train_seg{j}_sub'] = s.detach()[:, :, ::4, ::4].numpy()
        out[f'train_seg{j}_argmax'] = s.detach().argmax(1).numpy().astype(np.uint8)
    sd_after = m.state_dict()
    for k in list(sd_after)[:]:
        if k.endswith('running_mean') or k.endswith('running_var'):
            out['train_rs/' + k] = tap(sd_after[k], 4)

    if with_backward:
        nc = 10
        m.nc, m.gr = nc, 1.0
        m.hyp = loss_ref.scaled_hyp(1024, nc, 3)
        targets = synth.synth_det_targets(2, 8, nc, seed=1)
        mask = synth.synth_seg_targets(2, H, W, 19, seed=1)
        cl = ref.loss.ComputeLoss(m)
        loss, items = cl(det, targets)
        if isinstance(seg, list):
            crit = ref.loss.SegmentationLosses(aux=True, aux_num=2, aux_weight=0.1)
            segloss = crit(seg[0], seg[1], seg[2], mask)
        else:
            crit = ref.loss.SegmentationLosses()
            segloss = crit(seg, mask)
        total = loss * 0.6 + segloss * 2 * 0.35         # train.py:366-391 gains (detgain .6, seggain .35, *batch)
        total.backward()
        out['loss_det'] = loss.detach().numpy()
        out['loss_items'] = items.numpy()
        out['loss_seg'] = segloss.detach().numpy().reshape(1)
        for k, p in m.named_parameters():
            g = p.grad.reshape(-1)
            out['grad/' + k] = np.concatenate(([g.norm().item()], g[:5].numpy(), g[-3:].numpy())).astype(np.float32)

    # ---- eval forward, fused (detect.py path: attempt_load -> fuse().eval()) ----
    m2, _ = build_ref(ref, cfg_name)
    m2.fuse().eval()
    with torch.no_grad():
        (pred, raw), seg = m2(x[:1])
    out['eval_pred'] = pred.numpy()
    out['eval_seg_sub'] = seg[:, :, ::4, ::4].numpy()
    out['eval_seg_argmax'] = seg.argmax(1).numpy().astype(np.uint8)
    np.savez_compressed(os.path.join(GOLD, f'model_{tag}.npz'), **out)
    print('wrote', tag, len(out), 'arrays')


BLOCKS = {
    # name: (reference ctor, input shape [N,C,H,W])  -- blocks no shipped yaml instantiates (north_star names them): pinned directly
    'rfb1': (lambda C: C.RFB1(64, 64, map_reduce=4, d=[3, 5, 7], has_globel=False), (2, 64, 12, 16)),
    'rfb1_global': (lambda C: C.RFB1(96, 64, map_reduce=6, d=[3, 5, 7], has_globel=True), (2, 96, 8, 12)),
    'arm': (lambda C: C.ARM(64, 32), (2, 64, 8, 12)),
    'attention': (lambda C: C.Attention(64), (3, 64, 6, 8)),
    'attention_r4': (lambda C: C.Attention(64, reduction=4), (2, 64, 6, 8)),
}


def block_inputs(name):
    """(state_dict, x, r): synthetic weights for the block (oracle.synth streams over the reference class's own state_dict shapes),
    input and the fixed cotangent of the backward."""
    import zlib
    seed = zlib.crc32(name.encode()) % 1000
    rs = np.random.RandomState(100 + seed)
    shape = BLOCKS[name][1]
    x = torch.from_numpy(rs.normal(0, 1, shape).astype(np.float32))
    return x, seed


def blocks_case(ref):
    """RFB1 / ARM / Attention run from the reference's own classes (models/common.py:177-207,416-466): train-mode forward +
    backward (batch-stat BN, all parameter and input gradients) and eval-mode forward."""
    import models.common as rcommon
    out = {}
    for name, (ctor, shape) in BLOCKS.items():
        torch.manual_seed(0)
        m = ctor(rcommon)
        ref.torch_utils.initialize_weights(m)                      # eps 1e-3 / momentum 0.03 as inside Model
        sd = synth.synth_state_dict({k: v.clone() for k, v in m.state_dict().items()}, seed=3)
        m.load_state_dict(sd, strict=True)
        x, seed = block_inputs(name)
        out[f'{name}/x'] = x.numpy()                                # (weights are regenerated by the tests: synth_state_dict(seed=3))
        m.train()
        xin = x.clone().requires_grad_()
        y = m(xin)
        r = torch.from_numpy(np.random.RandomState(200 + seed).normal(0, 1, tuple(y.shape)).astype(np.float32))
        (y * r).sum().backward()
        out[f'{name}/r'] = r.numpy()
        out[f'{name}/train_out'] = y.detach().numpy()
        out[f'{name}/dx'] = xin.grad.numpy()
        for k, p_ in m.named_parameters():
            out[f'{name}/grad/{k}'] = p_.grad.numpy()
        for k, b in m.named_buffers():
            if 'running' in k:
                out[f'{name}/after/{k}'] = b.numpy().copy()
        m.load_state_dict(sd, strict=True)
        m.eval()
        with torch.no_grad():
            out[f'{name}/eval_out'] = m(x).numpy()
    np.savez_compressed(os.path.join(GOLD, 'blocks.npz'), **out)
    print('wrote blocks', len(out))


def loss_case(ref):
    out = {}
    rs = np.random.RandomState(7)
    B, nc = 2, 10
    p = [torch.from_numpy(rs.normal(0, 1.5, (B, 3, ny, nx, 5 + nc)).astype(np.float32)).requires_grad_()
         for ny, nx in ((16, 32), (8, 16), (4, 8))]
    targets = synth.synth_det_targets(B, 24, nc, seed=5)      # dense: duplicates in tobj scatter do occur
    targets[:4, 2:4] = torch.tensor([[0.003, 0.5], [0.997, 0.5], [0.5, 0.004], [0.5, 0.996]])  # border clamps

    class FakeDet:
        pass
    for ls in (0.0, 0.1):
        m, _ = build_ref(ref, 'yolov5s_city_seg.yaml') if ls == 0.0 else (m, None)
        m.nc, m.gr = nc, 1.0
        m.hyp = loss_ref.scaled_hyp(1024, nc, 3, label_smoothing=ls)
        cl = ref.loss.ComputeLoss(m)
        for q in p:
            q.grad = None
        loss, items = cl(p, targets)
        loss.backward()
        tag = f'ls{int(ls * 10)}'
        out[f'det_{tag}_loss'] = loss.detach().numpy()
        out[f'det_{tag}_items'] = items.numpy()
        for i, q in enumerate(p):
            out[f'det_{tag}_grad{i}'] = q.grad.numpy().copy()
    out['anchors'] = m.model[-1].anchors.numpy()
    for i, q in enumerate(p):
        out[f'det_p{i}'] = q.detach().numpy()
    out['det_targets'] = targets.numpy()
    # empty-target edge case
    loss, items = cl([q.detach() for q in p], torch.zeros(0, 6))
    out['det_empty_loss'] = loss.numpy()
    out['det_empty_items'] = items.numpy()

    # segmentation CE + OHEM on [2,19,32,64]
    logits = torch.from_numpy(rs.normal(0, 2.0, (2, 19, 32, 64)).astype(np.float32)).requires_grad_()
    mask = synth.synth_seg_targets(2, 32, 64, 19, seed=3, blocky=4)
    # make a blocky region "easy" so OHEM's threshold actually splits the pixels
    with torch.no_grad():
        easy = torch.zeros_like(mask, dtype=torch.bool)
        easy[:, :, :40] = True
        idx = mask.clamp(0)
        boost = torch.zeros_like(logits).scatter_(1, idx[:, None], 8.0)
        logits += boost * easy[:, None]
    ce = ref.loss.SegmentationLosses()(logits, mask)
    ce.backward()
    out['seg_logits'], out['seg_mask'] = logits.detach().numpy(), mask.numpy().astype(np.int16)
    out['ce_loss'], out['ce_grad'] = ce.detach().numpy().reshape(1), logits.grad.numpy().copy()
    for th in (0.7, 0.999999):     # 0.7: thresholded branch; ~1.0: thresh ~ 0 -> all kept; see below for topk branch
        logits.grad = None
        oh = ref.loss.OhemCELoss.__new__(ref.loss.OhemCELoss)       # ctor calls .cuda() (loss.py:306): bypass
        torch.nn.Module.__init__(oh)
        oh.thresh = -torch.log(torch.tensor(th, dtype=torch.float))
        oh.ignore_index, oh.aux = -1, False
        oh.criteria = torch.nn.CrossEntropyLoss(ignore_index=-1, reduction='none')
        v = oh(logits, mask)
        v.backward()
        out[f'ohem_{th}_loss'], out[f'ohem_{th}_grad'] = v.detach().numpy().reshape(1), logits.grad.numpy().copy()
    # top-k fallback branch: all pixels easy except few -> fewer than n_min above threshold
    logits2 = (logits.detach() + boost * 1.0).requires_grad_()
    oh.thresh = -torch.log(torch.tensor(0.7))
    v = oh(logits2, mask)
    v.backward()
    out['ohem_topk_logits'] = logits2.detach().numpy()
    out['ohem_topk_loss'], out['ohem_topk_grad'] = v.detach().numpy().reshape(1), logits2.grad.numpy().copy()
    np.savez_compressed(os.path.join(GOLD, 'losses.npz'), **out)
    print('wrote losses', len(out))


def nms_case(ref):
    out = {}
    pred = synth.synth_nms_pred(2, 3000, 10, seed=3)
    for name, kw in (('single', dict(conf_thres=0.25, iou_thres=0.45)),
                     ('multi', dict(conf_thres=0.001, iou_thres=0.6, multi_label=True)),
                     ('classes', dict(conf_thres=0.25, iou_thres=0.45, classes=[2, 5, 7])),
                     ('agnostic', dict(conf_thres=0.25, iou_thres=0.45, agnostic=True)),
                     ('multi_classes', dict(conf_thres=0.05, iou_thres=0.6, multi_label=True, classes=[0, 9]))):
        res = ref.general.non_max_suppression(pred.clone(), **kw)
        for i, r in enumerate(res):
            out[f'{name}_{i}'] = r.numpy()
    # fp16 predictions (detect.py --half, BASELINE config 5): the reference's own arithmetic in the input dtype, on the CPU
    for name, kw in (('single_f16', dict(conf_thres=0.25, iou_thres=0.45)), ('multi_f16', dict(conf_thres=0.001, iou_thres=0.6, multi_label=True))):
        res = ref.general.non_max_suppression(pred.clone().half(), **kw)
        for i, r in enumerate(res):
            assert r.dtype == torch.float32
            out[f'{name}_{i}'] = r.numpy()
    np.savez_compressed(os.path.join(GOLD, 'nms.npz'), **out)
    print('wrote nms', {k: v.shape for k, v in out.items()})


def metrics_case(ref):
    import utils.metrics as rmetrics          # the reference's own module (matplotlib is importable here)
    rs = np.random.RandomState(11)
    logits = torch.from_numpy(rs.normal(0, 2.0, (2, 19, 48, 80)).astype(np.float32))
    mask = synth.synth_seg_targets(2, 48, 80, 19, seed=6, blocky=4)
    with torch.no_grad():                      # make ~60 % of the pixels correct so that every counter is exercised
        good = torch.from_numpy(rs.uniform(size=(2, 48, 80)) < 0.6)
        logits += torch.zeros_like(logits).scatter_(1, mask.clamp(0)[:, None], 9.0) * good[:, None]
    correct, labeled = rmetrics.batch_pix_accuracy(logits, mask)
    inter, union = rmetrics.batch_intersection_union(logits, mask, 19)
    np.savez_compressed(os.path.join(GOLD, 'metrics.npz'), logits=logits.numpy(), mask=mask.numpy().astype(np.int16),
                        correct=np.array(correct), labeled=np.array(labeled), inter=inter, union=union)
    print('wrote metrics', int(correct), int(labeled))


def match_inputs(seed, n_lab=40, n_fp=60, nc=10, W=2048, H=1024):
    """labels [m,5] (cls, xyxy) and NMS-ordered predictions [n,6] (xyxy, conf, cls): jittered copies of the labels (2 per label, so
    duplicates compete for a target), class-confused copies and random false positives"""
    rs = np.random.RandomState(seed)
    cx, cy = rs.uniform(50, W - 50, n_lab), rs.uniform(50, H - 50, n_lab)
    w, h = rs.uniform(20, 300, n_lab), rs.uniform(20, 200, n_lab)
    lab = np.stack([rs.randint(0, nc, n_lab).astype(np.float32), cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)
    preds = []
    for rep, jit in ((0, 0.03), (1, 0.12), (2, 0.3)):
        j = lab[:, 1:5] + rs.normal(0, 1, (n_lab, 4)).astype(np.float32) * jit * np.stack([w, h, w, h], 1).astype(np.float32)
        cls = lab[:, 0].copy()
        if rep == 2:
            flip = rs.uniform(size=n_lab) < 0.3
            cls[flip] = (cls[flip] + 1) % nc
        preds.append(np.concatenate([j, rs.uniform(0.05, 1, (n_lab, 1)).astype(np.float32), cls[:, None]], 1))
    fx, fy = rs.uniform(0, W - 100, n_fp), rs.uniform(0, H - 100, n_fp)
    fp = np.stack([fx, fy, fx + rs.uniform(10, 200, n_fp), fy + rs.uniform(10, 200, n_fp), rs.uniform(0.05, 1, n_fp),
                   rs.randint(0, nc + 2, n_fp)], 1).astype(np.float32)
    p = np.concatenate(preds + [fp], 0).astype(np.float32)
    p = p[np.argsort(-p[:, 4], kind='stable')]
    return torch.from_numpy(p), torch.from_numpy(lab)


def match_case(ref):
    from . import metrics_ref
    out = {}
    iouv = torch.linspace(0.5, 0.95, 10)                        # test.py:98
    for i, seed in enumerate((1, 2, 3)):
        p, l = match_inputs(seed, n_lab=(40, 7, 120)[i])
        assert torch.equal(metrics_ref.box_iou(p[:, :4], l[:, 1:5]), ref.general.box_iou(p[:, :4], l[:, 1:5]))
        c = metrics_ref.match_predictions(p, l, iouv, iou_fn=ref.general.box_iou)      # test.py:230-262 around the reference's box_iou
        out[f'pred_{i}'], out[f'labels_{i}'], out[f'correct_{i}'] = p.numpy(), l.numpy(), c.numpy()
        print('match', i, p.shape[0], l.shape[0], int(c[:, 0].sum()), int(c[:, -1].sum()))
    np.savez_compressed(os.path.join(GOLD, 'match.npz'), iouv=iouv.numpy(), **out)


def letterbox_case(ref):
    """geometry (and border pixels) of the reference's own utils.datasets.letterbox for frames that need no resampling; cv2 is not
    installed, so its one call on that branch (copyMakeBorder BORDER_CONSTANT) is served by a numpy constant fill"""
    cv2 = sys.modules['cv2']

    def copy_make_border(img, top, bottom, left, right, border_type, value=(0, 0, 0)):
        out = np.empty((img.shape[0] + top + bottom, img.shape[1] + left + right, img.shape[2]), img.dtype)
        out[...] = np.array(value, img.dtype)
        out[top:top + img.shape[0], left:left + img.shape[1]] = img
        return out
    cv2.copyMakeBorder, cv2.BORDER_CONSTANT, cv2.INTER_LINEAR = copy_make_border, 0, 1
    import utils.datasets as rdatasets
    cases = [((1024, 2048), 2048, True), ((1000, 2048), 2048, True), ((37, 64), 64, True), ((64, 50), 64, False), ((480, 640), 640, True),
             ((640, 640), 640, False), ((333, 640), 640, True), ((720, 1280), 1280, True)]
    out = {'cases': np.array([[h, w, ns, int(auto)] for (h, w), ns, auto in cases])}
    rs = np.random.RandomState(0)
    for i, ((h, w), ns, auto) in enumerate(cases):
        im = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        img, ratio, pad = rdatasets.letterbox(im, ns, auto=auto, stride=32)
        out[f'geom_{i}'] = np.array([img.shape[0], img.shape[1], ratio[0], ratio[1], pad[0], pad[1]], np.float64)
        out[f'sum_{i}'] = np.array([int(img.astype(np.int64).sum()), int(img[0, 0, 0]), int(img[-1, -1, 2])])
    np.savez_compressed(os.path.join(GOLD, 'letterbox.npz'), **out)
    print('wrote letterbox', len(cases))


def boxes_case(ref):
    """the box helpers detect.py / test.py chain after NMS (utils/general.py xywh2xyxy 265-272, scale_coords 319-332 incl. clip_coords
    335-340, box_iou 388-410), run from the reference itself"""
    rs = np.random.RandomState(4)
    out = {}
    xywh = torch.from_numpy(rs.uniform(0, 600, (50, 4)).astype(np.float32))
    out['xywh'], out['xyxy'] = xywh.numpy(), ref.general.xywh2xyxy(xywh).numpy()
    cases = [((512, 1024), (1000, 2000), None), ((384, 640), (720, 1280), None), ((640, 640), (480, 640), None),
             ((1024, 2048), (1024, 2048), ((1.0, 1.0), (0.0, 12.0)))]
    out['ncases'] = np.array(len(cases))
    for i, (s1, s0, rp) in enumerate(cases):
        c = torch.from_numpy(rs.uniform(-50, max(s1) + 50, (40, 6)).astype(np.float32))
        out[f'coords_in_{i}'] = c.numpy().copy()
        out[f'coords_out_{i}'] = ref.general.scale_coords(s1, c[:, :4].clone(), s0, rp).numpy()
    np.savez_compressed(os.path.join(GOLD, 'boxes.npz'), **out)
    print('wrote boxes')


def augment_inputs(i):
    """synthetic (RGB uint8 image, Cityscapes label-id map) of case i: smooth gradients + noise so that resampling errors show"""
    shapes = [(96, 192), (150, 100), (64, 128), (200, 260)]
    h, w = shapes[i % len(shapes)]
    rs = np.random.RandomState(100 + i)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) * 7) % 256], 2).astype(np.int64)
    img = np.clip(img + rs.randint(-40, 41, img.shape), 0, 255).astype(np.uint8)
    ids = np.array([0, 7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33, 1, 5], np.uint8)
    mask = ids[rs.randint(0, len(ids), (h // 8 + 1, w // 8 + 1))].repeat(8, 0).repeat(8, 1)[:h, :w]
    return img, np.ascontiguousarray(mask)


AUG_CASES = [  # (input case, seed, base_size, crop_size (w, h))
    (0, 1, 128, (64, 48)), (0, 2, 128, (160, 96)), (1, 3, 128, (64, 64)), (2, 4, 96, (96, 64)), (3, 5, 256, (128, 96)), (3, 6, 64, (200, 120)),
]


def augment_case(ref):
    """SegmentationDataset.py:118-151 `_sync_transform` + :225-228 `_mask_transform`, run from the reference's own classes with PIL
    (installed here) under a seeded `random`: pins the random call order, Pillow's BILINEAR / NEAREST resize, padding, crop and the
    label-id table."""
    import random
    from PIL import Image
    import SegmentationDataset as rsd            # the reference's module (repo root)
    rsd.get_city_pairs = lambda *a, **k: (['x'], ['x'])
    out = {}
    for ci, (inp, seed, base, crop) in enumerate(AUG_CASES):
        ds = rsd.CitySegmentation(root='.', split='train', mode='train', base_size=base, crop_size=crop, low=0.65, high=3, sample_std=25)
        img, mask = augment_inputs(inp)
        random.seed(seed)
        a, b = ds._sync_transform(Image.fromarray(img), Image.fromarray(mask))
        lab = ds._mask_transform(b)
        out[f'c{ci}.img'] = np.array(a)
        out[f'c{ci}.lab'] = lab.numpy().astype(np.int8)
        out[f'c{ci}.after'] = np.float64(random.random())            # the generator state after the transform (same number of draws)
    # validation samples (mode='testval', train.py:228-229): image through _testval_img_transform, label map through _mask_transform only
    for inp, base in ((0, 128), (1, 96), (3, 160)):
        ds = rsd.CitySegmentation(root='.', split='val', mode='testval', base_size=base, crop_size=(64, 64))
        img, mask = augment_inputs(inp)
        out[f'tv{inp}.img'] = np.array(ds._testval_img_transform(Image.fromarray(img)))
        out[f'tv{inp}.lab'] = ds._mask_transform(Image.fromarray(mask)).numpy().astype(np.int8)
    np.savez_compressed(os.path.join(GOLD, 'augment_seg.npz'), **out)
    print('augment_seg', {k: v.shape for k, v in out.items() if k.endswith('img')})


JITTER_CASES = [  # (input case, order, brightness, contrast, saturation, hue)
    (0, (0, 1, 2, 3), 1.2, 0.8, 1.3, 0.05), (1, (3, 2, 1, 0), 0.6, 1.4, 0.7, -0.12), (2, (1, 3, 0, 2), 1.45, 0.55, 1.45, 0.15),
    (3, (2, 0, 3, 1), 0.9, 1.0, 0.0, -0.15), (0, (3, 1, 2, 0), 1.0, 1.2, 1.0, 0.0),
]


def jitter_case(ref):
    """ColorJitter's four adjustments through the REAL Pillow (ImageEnhance.Brightness/Contrast/Color, convert('HSV')), composed as
    torchvision's functional_pil.py composes them (torchvision itself is not installed)"""
    from PIL import Image, ImageEnhance
    out = {}
    for ci, (inp, order, b, c, sat, hue) in enumerate(JITTER_CASES):
        img = Image.fromarray(augment_inputs(inp)[0])
        for op in order:
            if op == 0:
                img = ImageEnhance.Brightness(img).enhance(b)
            elif op == 1:
                img = ImageEnhance.Contrast(img).enhance(c)
            elif op == 2:
                img = ImageEnhance.Color(img).enhance(sat)
            else:                                             # functional_pil.adjust_hue
                h, s_, v = img.convert('HSV').split()
                np_h = np.array(h, dtype=np.uint8)
                np_h = (np_h.astype(np.int64) + int(hue * 255)).astype(np.uint8)       # uint8 wrap-around add
                img = Image.merge('HSV', (Image.fromarray(np_h, 'L'), s_, v)).convert('RGB')
        out[f'c{ci}'] = np.array(img)
    np.savez_compressed(os.path.join(GOLD, 'augment_jitter.npz'), **out)
    print('augment_jitter', len(out))


DET_S = 96                                                        # img_size of the detection augmentation cases
DET_HYPS = {
    'scratch': dict(degrees=0.0, translate=0.1, scale=0.5, shear=0.0, perspective=0.0, flipud=0.0, fliplr=0.5, mosaic=1.0, mixup=0.0,
                    hsv_h=0.015, hsv_s=0.7, hsv_v=0.4),          # data/hyp.scratch.yaml
    'rot': dict(degrees=10.0, translate=0.1, scale=0.5, shear=5.0, perspective=0.0, flipud=0.5, fliplr=0.5, mosaic=1.0, mixup=0.0,
                hsv_h=0.015, hsv_s=0.7, hsv_v=0.4),
}
DET_HYPS['single'] = dict(DET_HYPS['scratch'], mosaic=0.0)       # the letterbox + random_perspective branch (datasets.py:536-556)
DET_HYPS['single_rot'] = dict(DET_HYPS['rot'], mosaic=0.0)
DET_CASES = [('scratch', 0, 1), ('scratch', 3, 2), ('rot', 1, 3), ('rot', 5, 4), ('scratch', 2, 7), ('rot', 4, 11),
             ('single', 0, 5), ('single', 3, 6), ('single_rot', 1, 8), ('single_rot', 4, 9)]   # (hyp, index, seed)


def det_dataset():
    """6 synthetic BGR images as load_image would cache them (long side = DET_S) + normalised xywh labels"""
    shapes = [(96, 64), (72, 96), (96, 96), (50, 96), (96, 80), (64, 96)]
    imgs, labels = [], []
    for i, (h, w) in enumerate(shapes):
        rs = np.random.RandomState(200 + i)
        yy, xx = np.mgrid[0:h, 0:w]
        im = np.stack([(xx * 255 // (w - 1)), (yy * 255 // (h - 1)), ((xx * 3 + yy * 5) % 256)], 2).astype(np.int64)
        imgs.append(np.clip(im + rs.randint(-30, 31, im.shape), 0, 255).astype(np.uint8))
        n = 3 + i % 4
        xy = rs.uniform(0.2, 0.8, (n, 2))
        wh = rs.uniform(0.1, 0.5, (n, 2))
        labels.append(np.concatenate([rs.randint(0, 10, (n, 1)).astype(np.float64), xy, wh], 1).astype(np.float32))
    return imgs, labels


def det_augment_case(ref):
    """utils/datasets.py LoadImagesAndLabels.__getitem__ (518-593, mosaic branch: load_mosaic 672-724 -> random_perspective 851-925 ->
    augment_hsv 646-658 -> flips -> transpose), the reference's OWN code on a stand-in dataset object; its cv2 calls are served by
    the restatements of oracle/aug_ref.py (cv2 is not installed): pins the random call order, mosaic geometry, label arithmetic"""
    import random
    from types import SimpleNamespace
    from . import aug_ref
    aug_ref.install_cv2_stub(sys.modules['cv2'])
    import utils.datasets as rds
    imgs, labels = det_dataset()
    out = {}
    for ci, (hyp, index, seed) in enumerate(DET_CASES):
        ds = SimpleNamespace(indices=range(len(imgs)), hyp=DET_HYPS[hyp], mosaic=True, augment=True, rect=False, n=len(imgs), img_size=DET_S,
                             mosaic_border=[-DET_S // 2, -DET_S // 2], labels=[l.copy() for l in labels], segments=[[] for _ in imgs],
                             imgs=imgs, img_hw0=[im.shape[:2] for im in imgs], img_hw=[im.shape[:2] for im in imgs],
                             img_files=[f'{i}.jpg' for i in range(len(imgs))])
        random.seed(seed)
        np.random.seed(seed)
        img, lab, _, _ = rds.LoadImagesAndLabels.__getitem__(ds, index)
        out[f'c{ci}.img'] = img.numpy()
        out[f'c{ci}.lab'] = lab.numpy()
        out[f'c{ci}.after'] = np.array([random.random(), np.random.rand()])
    np.savez_compressed(os.path.join(GOLD, 'augment_det.npz'), **out)
    print('augment_det', {k: v.shape for k, v in out.items() if not k.endswith('after')})


def main():
    os.makedirs(GOLD, exist_ok=True)
    ref = ref_shim.install()
    torch.set_num_threads(8)
    which = sys.argv[1:] or ['models', 'blocks', 'losses', 'nms', 'metrics', 'match', 'letterbox', 'boxes', 'augment']
    if 'models' in which:
        model_case(ref, 'yolov5s_city_seg.yaml', 's_psp', True)
        model_case(ref, 'yolov5s_city_seg_base.yaml', 's_base', True)
        model_case(ref, 'yolov5s_city_seg_lab.yaml', 's_lab', True)
        model_case(ref, 'yolov5s_city_seg_bise.yaml', 's_bise', True)
        model_case(ref, 'yolov5m_city_seg_lab.yaml', 'm_lab', True)
    if 'm_lab' in which:
        model_case(ref, 'yolov5m_city_seg_lab.yaml', 'm_lab', True)
    if 'blocks' in which:
        blocks_case(ref)
    if 'losses' in which:
        loss_case(ref)
    if 'nms' in which:
        nms_case(ref)
    if 'metrics' in which:
        metrics_case(ref)
    if 'match' in which:
        match_case(ref)
    if 'letterbox' in which:
        letterbox_case(ref)
    if 'boxes' in which:
        boxes_case(ref)
    if 'augment' in which:
        augment_case(ref)
        jitter_case(ref)
        det_augment_case(ref)


if __name__ == '__main__':
    main()
