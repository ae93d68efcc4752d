"""CPU: the segmentation-augmentation oracle (oracle/aug_ref.py) against the golden written by the reference's own `_sync_transform`
running on the real Pillow (tests/golden/augment_seg.npz, oracle/make_golden.py augment_case), and the host half of the product
(multiyolov5_amd/utils/augment.py: random call order, Pillow coefficient / index tables) against the oracle."""
import random

import numpy as np
import pytest

from oracle import aug_ref
from oracle.make_golden import AUG_CASES, augment_inputs
from tests.util import golden


@pytest.mark.parametrize('ci', range(len(AUG_CASES)))
def test_oracle_sync_transform_matches_reference_golden(ci):
    g = golden('augment_seg')
    inp, seed, base, crop = AUG_CASES[ci]
    img, mask = augment_inputs(inp)
    rng = random.Random(seed)
    p = aug_ref.draw_params(rng, img.shape[1], img.shape[0], base, crop)
    a, lab = aug_ref.sync_transform(img, mask, p)
    assert np.array_equal(a, g[f'c{ci}.img'])
    assert np.array_equal(lab, g[f'c{ci}.lab'].astype(np.int64))
    assert rng.random() == float(g[f'c{ci}.after'])               # same number of `random` draws as the reference made


@pytest.mark.parametrize('ci', range(len(AUG_CASES)))
def test_product_random_sequence_matches_oracle(ci):
    from multiyolov5_amd.utils import augment as A
    inp, seed, base, crop = AUG_CASES[ci]
    img, _ = augment_inputs(inp)
    want = aug_ref.draw_params(random.Random(seed), img.shape[1], img.shape[0], base, crop)
    rng = random.Random(seed)
    got = A.draw_sync_params(img.shape[1], img.shape[0], base, crop, rng=rng)
    assert got == {k: (bool(v) if k == 'flip' else int(v)) for k, v in want.items()}
    assert rng.random() == float(golden('augment_seg')[f'c{ci}.after'])


@pytest.mark.parametrize('sizes', [(64, 100), (100, 64), (200, 67), (67, 200), (37, 37), (2048, 1504), (1024, 2336), (5, 3), (3, 5)])
def test_product_pillow_tables_match_oracle(sizes):
    from multiyolov5_amd.utils import augment as A
    n_in, n_out = sizes
    b, k = A.pil_bilinear_tables(n_in, n_out)
    if n_in != n_out:
        rb, rk = aug_ref._coeffs(n_in, n_out)
        assert np.array_equal(b, rb) and np.array_equal(k, rk)
    else:
        assert np.array_equal(b[:, 0], np.arange(n_out)) and np.all(b[:, 1] == 1) and np.all(k == 1 << 22)
    assert np.array_equal(A.pil_nearest_table(n_in, n_out), aug_ref._nearest_tab(n_in, n_out))


def test_label_table_matches_class_to_index():
    from multiyolov5_amd.utils import augment as A
    lut = A.city_label_lut()
    ids = np.array(list(range(34)) + [255], np.uint8)
    assert np.array_equal(lut[ids], aug_ref.class_to_index(ids))


def test_oracle_resize_matches_pillow_when_installed():
    """extra pin where Pillow is importable (it is in this image): random sizes, up- and down-scaling"""
    Image = pytest.importorskip('PIL.Image')
    rs = np.random.RandomState(0)
    for (h, w, ow, oh) in [(37, 64, 100, 58), (64, 128, 40, 20), (50, 70, 70, 33), (33, 47, 47, 80), (120, 200, 67, 40), (90, 90, 31, 200)]:
        a = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        assert np.array_equal(aug_ref.pil_resize_bilinear(a, ow, oh), np.array(Image.fromarray(a).resize((ow, oh), Image.BILINEAR)))
        m = rs.randint(0, 34, (h, w)).astype(np.uint8)
        assert np.array_equal(aug_ref.pil_resize_nearest(m, ow, oh), np.array(Image.fromarray(m).resize((ow, oh), Image.NEAREST)))


def test_oracle_color_jitter_matches_pillow_golden():
    from oracle.make_golden import JITTER_CASES
    g = golden('augment_jitter')
    for ci, (inp, order, b, c, s, h) in enumerate(JITTER_CASES):
        img, _ = augment_inputs(inp)
        assert np.array_equal(aug_ref.color_jitter(img, order, b, c, s, h), g[f'c{ci}']), ci


def test_oracle_hsv_round_trip_matches_pillow_on_a_colour_lattice():
    """every 5th level of each channel + the extremes (the full 2^24 sweep was run once while writing the restatement)"""
    Image = pytest.importorskip('PIL.Image')
    lv = np.unique(np.concatenate([np.arange(0, 256, 5), [1, 2, 127, 128, 253, 254, 255]])).astype(np.uint8)
    rr, gg, bb = np.meshgrid(lv, lv, lv, indexing='ij')
    a = np.stack([rr, gg, bb], -1).reshape(len(lv), -1, 3)
    assert np.array_equal(aug_ref.rgb2hsv(a), np.array(Image.fromarray(a).convert('HSV')))
    assert np.array_equal(aug_ref.hsv2rgb(a), np.array(Image.fromarray(a, 'HSV').convert('RGB')))


def test_mosaic_sample_host_half_matches_reference_golden(monkeypatch):
    """labels, geometry and random call order of utils/augment.mosaic_train_sample against the golden the reference's own
    LoadImagesAndLabels.__getitem__ produced (the pixel launch is replaced by a recorder: no GPU here)"""
    import random
    import torch
    from multiyolov5_amd import _lib as L
    from multiyolov5_amd.utils import augment as A
    from oracle.make_golden import DET_CASES, DET_HYPS, DET_S, det_dataset
    imgs, labels = det_dataset()
    g = golden('augment_det')
    calls = []

    class FakeLib:
        def myolo_mosaic_warp(self, d, st):
            calls.append(d)
            return 0
    monkeypatch.setattr(L, 'lib', lambda: FakeLib())
    monkeypatch.setattr(L, 'require_gpu', lambda t: None)
    monkeypatch.setattr(L, 'stream_ptr', lambda: None)
    for ci, (hyp, index, seed) in enumerate(DET_CASES):
        rng = random.Random(seed)
        nprng = np.random.RandomState(seed)
        t = [torch.from_numpy(im) for im in imgs]
        out, lab = A.mosaic_train_sample(index, lambda i: t[i], lambda i: labels[i], range(len(imgs)), DET_S, DET_HYPS[hyp], rng, nprng)
        assert tuple(out.shape) == (3, DET_S, DET_S)
        np.testing.assert_array_equal(lab.numpy(), g[f'c{ci}.lab'])
        assert [rng.random(), nprng.rand()] == list(g[f'c{ci}.after'])
    assert len(calls) == len(DET_CASES)


def test_testval_size_matches_reference_golden_shapes():
    from multiyolov5_amd.utils import augment as A
    g = golden('augment_seg')
    for inp, base in ((0, 128), (1, 96), (3, 160)):
        img, _ = augment_inputs(inp)
        ow, oh = A.testval_size(img.shape[1], img.shape[0], base)
        assert g[f'tv{inp}.img'].shape[:2] == (oh, ow)
        assert np.array_equal(aug_ref.pil_resize_bilinear(img, ow, oh), g[f'tv{inp}.img'])
