"""-m gpu: training-time augmentation kernels (csrc/augment.hip) through the host mirror, bit-exact against the oracle and the golden
the reference's own `_sync_transform` wrote (SegmentationDataset.py:118-151 on the real Pillow)."""
import random

import numpy as np
import pytest
import torch

from oracle import aug_ref
from oracle.make_golden import AUG_CASES, augment_inputs
from tests.util import golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('ci', range(len(AUG_CASES)))
def test_seg_sync_transform_matches_reference_golden(ci):
    from multiyolov5_amd.utils import augment as A
    g = golden('augment_seg')
    inp, seed, base, crop = AUG_CASES[ci]
    img, mask = augment_inputs(inp)
    p = A.draw_sync_params(img.shape[1], img.shape[0], base, crop, rng=random.Random(seed))
    a, lab = A.seg_sync_transform(torch.from_numpy(img).to(DEV), torch.from_numpy(mask).to(DEV), p)
    assert a.dtype == torch.uint8 and lab.dtype == torch.int64
    assert np.array_equal(a.cpu().numpy(), g[f'c{ci}.img'])
    assert np.array_equal(lab.cpu().numpy(), g[f'c{ci}.lab'].astype(np.int64))


@pytest.mark.parametrize('case', [
    # H0, W0, flip, ow, oh, x1, y1, wc, hc
    (64, 128, 1, 128, 64, 0, 0, 128, 64),            # no resampling at all (identity tables), mirrored
    (64, 128, 0, 384, 192, 100, 50, 160, 96),        # x3 up-scale, interior crop
    (96, 200, 1, 67, 32, 0, 0, 96, 64),              # x3 down-scale (7-tap filter), padded on both sides
    (33, 47, 0, 47, 80, 3, 10, 40, 64),              # vertical pass only
    (80, 33, 1, 70, 80, 0, 0, 70, 80),               # horizontal pass only
    (256, 512, 0, 736, 368, 17, 5, 512, 256),        # a Cityscapes-like ratio
])
def test_seg_sync_transform_vs_oracle(case):
    from multiyolov5_amd.utils import augment as A
    H0, W0, flip, ow, oh, x1, y1, wc, hc = case
    rs = np.random.RandomState(H0 + W0)
    img = rs.randint(0, 256, (H0, W0, 3)).astype(np.uint8)
    ids = np.array(list(range(34)) + [255], np.uint8)
    mask = ids[rs.randint(0, len(ids), (H0, W0))]
    p = dict(flip=bool(flip), ow=ow, oh=oh, x1=x1, y1=y1, wc=wc, hc=hc)
    want_img, want_lab = aug_ref.sync_transform(img, mask, p)
    a, lab = A.seg_sync_transform(torch.from_numpy(img).to(DEV), torch.from_numpy(mask).to(DEV), p)
    assert np.array_equal(a.cpu().numpy(), want_img)
    assert np.array_equal(lab.cpu().numpy(), want_lab)
    # image only
    a2, lab2 = A.seg_sync_transform(torch.from_numpy(img).to(DEV), None, p)
    assert lab2 is None and torch.equal(a2, a)


def test_seg_sync_transform_rejects_cpu_and_bad_layouts():
    from multiyolov5_amd import _lib as L
    from multiyolov5_amd.utils import augment as A
    p = dict(flip=False, ow=8, oh=8, x1=0, y1=0, wc=8, hc=8)
    with pytest.raises(L.MyoloError):
        A.seg_sync_transform(torch.zeros(8, 8, 3, dtype=torch.uint8), None, p)
    with pytest.raises(L.MyoloError):
        A.seg_sync_transform(torch.zeros(8, 8, 3, device=DEV), None, p)
    with pytest.raises(L.MyoloError):
        A.seg_sync_transform(torch.zeros(8, 8, 3, dtype=torch.uint8, device=DEV), torch.zeros(4, 4, dtype=torch.uint8, device=DEV), p)


def test_color_jitter_matches_pillow_golden_and_oracle():
    from multiyolov5_amd.utils import augment as A
    from oracle.make_golden import JITTER_CASES
    g = golden('augment_jitter')
    for ci, (inp, order, b, c, s, h) in enumerate(JITTER_CASES):
        img, _ = augment_inputs(inp)
        p = dict(order=list(order), brightness=b, contrast=c, saturation=s, hue=h)
        d = torch.from_numpy(img).to(DEV)
        u8 = A.color_jitter(d, p, return_uint8=True)
        assert np.array_equal(u8.cpu().numpy(), g[f'c{ci}']), ci
        for dt in (torch.float32, torch.float16):                      # ToTensor: uint8 -> float32 / 255 (-> model dtype), CHW
            t = A.color_jitter(d, p, dtype=dt)
            want = (torch.from_numpy(g[f'c{ci}']).permute(2, 0, 1).float() / 255).to(dt)
            assert t.shape == want.shape and torch.equal(t.cpu(), want)


def test_color_jitter_partial_orders_and_every_hue_shift():
    from multiyolov5_amd.utils import augment as A
    rs = np.random.RandomState(3)
    img = rs.randint(0, 256, (64, 96, 3)).astype(np.uint8)
    img[0, :6] = [[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [9, 9, 9]]
    d = torch.from_numpy(img).to(DEV)
    for order, b, c, s, h in (((0, -1, -1, -1), 1.3, 1.0, 1.0, 0.0), ((-1, -1, 3, -1), 1.0, 1.0, 1.0, 0.37), ((2, -1, -1, 1), 1.0, 0.3, 1.8, 0.0),
                              ((-1, -1, -1, -1), 1.0, 1.0, 1.0, 0.0)):
        got = A.color_jitter(d, dict(order=list(order), brightness=b, contrast=c, saturation=s, hue=h), return_uint8=True).cpu().numpy()
        assert np.array_equal(got, aug_ref.color_jitter(img, [o for o in order if o >= 0], b, c, s, h)), order
    for shift in range(0, 256, 7):
        h = shift / 255.0 + 1e-9 if shift < 128 else (shift - 256) / 255.0 - 1e-9
        got = A.color_jitter(d, dict(order=[3, -1, -1, -1], brightness=1, contrast=1, saturation=1, hue=h), return_uint8=True).cpu().numpy()
        assert np.array_equal(got, aug_ref.color_jitter(img, [3], 1, 1, 1, h)), shift


def test_hue_round_trip_over_all_colours():
    """2^24 colours through rgb -> hsv -> (+0) -> rgb on the device against the numpy restatement (itself pinned to Pillow)"""
    from multiyolov5_amd.utils import augment as A
    v = np.arange(256, dtype=np.uint8)
    for r0 in range(0, 256, 64):
        rr, gg, bb = np.meshgrid(v[r0:r0 + 64], v, v, indexing='ij')
        a = np.ascontiguousarray(np.stack([rr, gg, bb], -1).reshape(64 * 256, 256, 3))
        got = A.color_jitter(torch.from_numpy(a).to(DEV), dict(order=[3, -1, -1, -1], brightness=1, contrast=1, saturation=1, hue=0.1),
                             return_uint8=True).cpu().numpy()
        assert np.array_equal(got, aug_ref.color_jitter(a, [3], 1, 1, 1, 0.1)), r0


def test_mosaic_train_sample_matches_reference_golden():
    """the whole detection sample (utils/datasets.py:518-593) on the device against the golden the reference's own __getitem__ wrote
    with oracle/aug_ref.py's restated cv2 underneath (cv2 is not installed: pixels are pinned to the restatement, everything else to
    the reference)"""
    from multiyolov5_amd.utils import augment as A
    from oracle.make_golden import DET_CASES, DET_HYPS, DET_S, det_dataset
    imgs, labels = det_dataset()
    g = golden('augment_det')
    t = [torch.from_numpy(im).to(DEV) for im in imgs]
    for ci, (hyp, index, seed) in enumerate(DET_CASES):
        out, lab = A.mosaic_train_sample(index, lambda i: t[i], lambda i: labels[i], range(len(imgs)), DET_S, DET_HYPS[hyp],
                                         random.Random(seed), np.random.RandomState(seed))
        assert out.dtype == torch.uint8 and np.array_equal(out.cpu().numpy(), g[f'c{ci}.img']), ci
        np.testing.assert_array_equal(lab.numpy(), g[f'c{ci}.lab'])


@pytest.mark.parametrize('shape', [((480, 640), (96, 72)), ((100, 37), (96, 35)), ((64, 128), (32, 64)), ((50, 50), (96, 96))])
def test_resize_u8_matches_cv_restatement(shape):
    from multiyolov5_amd.utils import augment as A
    from oracle import frame_ref
    (h0, w0), (rw, rh) = shape
    img = np.random.RandomState(h0).randint(0, 256, (h0, w0, 3)).astype(np.uint8)
    got = A.resize_u8(torch.from_numpy(img).to(DEV), rw, rh).cpu().numpy()
    assert np.array_equal(got, frame_ref.cv_resize_linear_u8(img, (rw, rh)))


def test_mosaic_warp_kernel_vs_oracle_pieces():
    """C ABI directly: single-source canvas + warp (rotation / shear / scale), HSV tables, flips, against the numpy restatements"""
    import ctypes as C
    from multiyolov5_amd import _lib as L
    from multiyolov5_amd.utils import augment as A
    rs = np.random.RandomState(5)
    img = rs.randint(0, 256, (70, 90, 3)).astype(np.uint8)
    d_img = torch.from_numpy(img).to(DEV)
    for case, (M, lut_on, fl, fu) in enumerate([
            (np.array([[1.2, 0.15, -8.0], [-0.1, 0.85, 6.0], [0, 0, 1.0]]), True, 0, 0),
            (np.array([[0.7, -0.3, 20.5], [0.25, 0.9, -3.25], [0, 0, 1.0]]), False, 1, 1),
            (np.eye(3), True, 1, 0)]):
        ow, oh = (64, 48) if case < 2 else (90, 70)
        warp = int(case < 2)
        lut = A.hsv_luts(dict(hsv_h=0.015, hsv_s=0.7, hsv_v=0.4), np.random.RandomState(case)) if lut_on else None
        want = aug_ref.cv_warp_affine_u8(img, M[:2], (ow, oh)) if warp else img.copy()
        if lut is not None:
            hsv = aug_ref.cv_bgr2hsv_u8(want)
            want = aug_ref.cv_hsv2bgr_u8(np.stack([lut[0][hsv[..., 0]], lut[1][hsv[..., 1]], lut[2][hsv[..., 2]]], -1))
        if fu:
            want = want[::-1]
        if fl:
            want = want[:, ::-1]
        d = L.MosaicDesc()
        d.nsrc, d.cw, d.ch, d.warp, d.ow, d.oh, d.fill, d.fliplr, d.flipud = 1, 90, 70, warp, ow, oh, 114, fl, fu
        s0 = d.src[0]
        s0.img, s0.h, s0.w, s0.x1a, s0.y1a, s0.x2a, s0.y2a, s0.padw, s0.padh = d_img.data_ptr(), 70, 90, 0, 0, 90, 70, 0, 0
        Mi = A._invert_for_warp(M)
        for k in range(6):
            d.M[k] = float(Mi.reshape(-1)[k])
        lut_d = torch.from_numpy(np.ascontiguousarray(lut)).to(DEV) if lut is not None else None
        hwc = torch.empty(oh, ow, 3, dtype=torch.uint8, device=DEV)
        chw = torch.empty(3, oh, ow, dtype=torch.uint8, device=DEV)
        d.hsv_lut, d.out_chw, d.out_hwc = (lut_d.data_ptr() if lut_d is not None else None), chw.data_ptr(), hwc.data_ptr()
        L.check(L.lib().myolo_mosaic_warp(C.byref(d), L.stream_ptr()), 'myolo_mosaic_warp')
        assert np.array_equal(hwc.cpu().numpy(), np.ascontiguousarray(want)), case
        assert np.array_equal(chw.cpu().numpy(), np.ascontiguousarray(want[:, :, ::-1].transpose(2, 0, 1))), case
    # rejected descriptors
    d.src[0].x2a = 200
    assert L.lib().myolo_mosaic_warp(C.byref(d), L.stream_ptr()) == L.EINVAL


@pytest.mark.parametrize('case', [(0, 128), (1, 96), (3, 160)])
def test_seg_testval_transform_matches_reference_golden(case):
    """validation samples (train.py:228-229, mode='testval'): golden from the reference's `_testval_img_transform` / `_mask_transform`"""
    from multiyolov5_amd.utils import augment as A
    inp, base = case
    g = golden('augment_seg')
    img, mask = augment_inputs(inp)
    a, lab = A.seg_testval_transform(torch.from_numpy(img).to(DEV), torch.from_numpy(mask).to(DEV), base)
    assert np.array_equal(a.cpu().numpy(), g[f'tv{inp}.img'])
    assert np.array_equal(lab.cpu().numpy(), g[f'tv{inp}.lab'].astype(np.int64))
