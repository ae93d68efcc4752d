"""-m gpu: K15 -- the segmentation head's final bilinear upsample (align_corners=True, reference models/yolo.py:163) + mean cross entropy
(utils/loss.py:236-237) + both backward passes in ONE pass over the low-resolution class logits (csrc/loss.hip seg_upce_kernel,
myolo_seg_upce_fwd_grad / myolo_seg_lowgrad_apply).

Reference of the test: plain torch fp32 on the CPU -- F.interpolate(bilinear, align_corners=True) of the low-res logits (rounded to
the storage type when it is fp16, as the reference's loss reads a materialised fp16 tensor under autocast), F.cross_entropy(sum) and
autograd.  Tolerances: loss 2e-6 relative (fp32) / 2e-4 (fp16: the rounding point of one logit may differ by an fma contraction);
low-res gradient 1e-4 of its largest entry (fp32 atomics in arbitrary order over <= ~300 addends)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _reference(low, tgt, H, W, ignore, dtype):
    x = low.detach().float().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_()       # [N,C,h,w]
    up = F.interpolate(x, size=(H, W), mode='bilinear', align_corners=True)
    if dtype == torch.float16:
        up = up + (up.detach().half().float() - up.detach())        # value rounded to fp16, gradient of the un-rounded map
    t = tgt.cpu()
    loss_sum = F.cross_entropy(up, t, ignore_index=ignore, reduction='sum')
    nvalid = int((t != ignore).sum())
    loss_sum.backward()
    return float(loss_sum), nvalid, x.grad.permute(0, 2, 3, 1).contiguous()


CASES = [
    # n, h, w, scale, ignore fraction
    (2, 8, 16, 8, 0.1),          # one strip, several row blocks
    (1, 9, 37, 8, 0.3),          # ragged strip (W = 296), odd low size
    (2, 4, 40, 16, 0.0),         # x16 (BiSe aux32), two strips
    (1, 5, 7, 8, 1.0),           # every pixel ignored
    (3, 16, 32, 8, 0.05),
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16], ids=['f32', 'f16'])
@pytest.mark.parametrize('case', CASES, ids=[f'{c[0]}x{c[1]}x{c[2]}x{c[3]}' for c in CASES])
def test_fused_upsample_ce_vs_torch_reference(case, dtype):
    from multiyolov5_amd import _lib as L
    n, h, w, scale, pign = case
    H, W = h * scale, w * scale
    g = torch.Generator().manual_seed(h * 100 + w)
    buf = torch.zeros(n, h, w, 24, dtype=dtype, device=DEV)                  # the plan's layout: 19 classes inside a 24-channel buffer
    low = buf[..., :19]
    low.copy_((torch.randn(n, h, w, 19, generator=g) * 3).to(DEV, dtype))
    tgt = torch.randint(0, 19, (n, H, W), generator=g)
    tgt[torch.rand(n, H, W, generator=g) < pign] = -1
    tgt = tgt.to(DEV)
    acc = torch.empty(2, dtype=torch.float64, device=DEV)
    loss = torch.empty(1, dtype=torch.float32, device=DEV)
    g32 = torch.full((n, h, w, 19), 7.0, dtype=torch.float32, device=DEV)    # (the entry point zeroes it)
    ld = L.Tensor(low.data_ptr(), n, h, w, 19, low.stride(0), low.stride(1), low.stride(2), L.DT[dtype], 0)
    L.check(L.lib().myolo_seg_upce_fwd_grad(C.byref(ld), H, W, L.ptr(tgt), -1, L.ptr(acc), L.ptr(loss), L.ptr(g32), L.stream_ptr()),
            'myolo_seg_upce_fwd_grad')
    torch.cuda.synchronize()
    ref_sum, nvalid, ref_g = _reference(low, tgt, H, W, -1, dtype)
    a = acc.cpu().numpy()
    assert a[1] == nvalid
    rtol = 2e-6 if dtype == torch.float32 else 2e-4
    np.testing.assert_allclose(a[0], ref_sum, rtol=rtol, atol=1e-6)
    if nvalid:
        np.testing.assert_allclose(float(loss), ref_sum / nvalid, rtol=rtol)
    got = g32.cpu()
    tol = 1e-4 * max(float(ref_g.abs().max()), 1e-6)
    assert float((got - ref_g).abs().max()) <= tol, (float((got - ref_g).abs().max()), tol)
    assert torch.all(buf[..., 19:] == 0)
    # gradient publication: glow (+)= g32 * scale
    gbuf = torch.zeros(n, h, w, 24, dtype=dtype, device=DEV)
    gbuf[..., :19] = 0.5
    gd = L.Tensor(gbuf.data_ptr(), n, h, w, 19, gbuf.stride(0), gbuf.stride(1), gbuf.stride(2), L.DT[dtype], 0)
    sc = torch.tensor([1.0 / max(nvalid, 1)], dtype=torch.float32, device=DEV)
    for accum in (0, 1):
        L.check(L.lib().myolo_seg_lowgrad_apply(L.ptr(g32), C.byref(gd), accum, L.ptr(sc), L.stream_ptr()), 'myolo_seg_lowgrad_apply')
    v = got * sc.cpu()                                                       # fp32 product, as the kernel forms it
    want = (v + v.to(dtype).float()).to(dtype).float()                       # call 1 stores T(v); call 2 stores T(v + T(v))
    torch.testing.assert_close(gbuf[..., :19].float().cpu(), want, rtol=2e-3, atol=1e-6)
    assert torch.all(gbuf[..., 19:] == 0)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16], ids=['f32', 'f16'])
def test_psp_head_training_step_fused_equals_unfused(dtype, monkeypatch):
    """the whole path through Model-level plumbing: the same head + SegmentationLosses step with MYOLO_FUSED_UPCE on and off gives the
    same loss and the same parameter / input gradients (up to the fp16 rounding of the full-resolution gradient the unfused path stores)."""
    from multiyolov5_amd.models import yolo as Y
    from multiyolov5_amd.utils import loss as loss_mod
    from multiyolov5_amd.utils.torch_utils import initialize_weights
    from tests.test_gpu_ops import _randomize
    torch.manual_seed(2)
    mod = Y.SegMaskPSP(19, 1, 64, False, [64, 128, 256])
    initialize_weights(mod)
    _randomize(mod)
    mod = mod.to(DEV).train()
    g = torch.Generator().manual_seed(5)
    shapes = [(2, 64, 16, 32), (2, 128, 8, 16), (2, 256, 4, 8)]
    xs_cpu = [torch.randn(s, generator=g) for s in shapes]
    tgt = torch.randint(-1, 19, (2, 128, 256), generator=g).to(DEV)
    crit = loss_mod.SegmentationLosses(ignore_index=-1)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(loss_mod, 'FUSED_UPCE', fused)
        mod.zero_grad(set_to_none=True)
        xs = [x.to(DEV, dtype).requires_grad_() for x in xs_cpu]
        out = mod(xs)
        assert out._myolo_grad_scale[1]['low'] is False
        loss = crit(out, tgt)
        assert out._myolo_grad_scale[1]['low'] is fused
        (loss * 3.0).backward()
        assert out._myolo_grad_scale[1]['low'] is False
        res[fused] = (float(loss), [x.grad.float().cpu() for x in xs], [p.grad.float().cpu().clone() for p in mod.parameters()])
    lt = 1e-5 if dtype == torch.float32 else 5e-4
    assert abs(res[True][0] - res[False][0]) <= lt * abs(res[False][0])
    tol = 2e-4 if dtype == torch.float32 else 2e-2
    for a, b in zip(res[True][1] + res[True][2], res[False][1] + res[False][2]):
        den = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) / den <= tol, float((a - b).abs().max()) / den


def test_unused_segmentation_loss_contributes_no_gradient():
    """loss computed through the fused path but left out of the backward: the plan must not apply the stale low-resolution gradient"""
    from multiyolov5_amd.models import yolo as Y
    from multiyolov5_amd.utils import loss as loss_mod
    from multiyolov5_amd.utils.torch_utils import initialize_weights
    torch.manual_seed(3)
    mod = Y.SegMaskBiSe(19, 1, 64, False, [64, 128, 256])
    initialize_weights(mod)
    mod = mod.to(DEV).train()
    g = torch.Generator().manual_seed(6)
    xs = [torch.randn(s, generator=g).to(DEV).requires_grad_() for s in [(2, 64, 16, 32), (2, 128, 8, 16), (2, 256, 4, 8)]]
    tgt = torch.randint(-1, 19, (2, 128, 256), generator=g).to(DEV)
    outs = mod(xs)
    assert isinstance(outs, list) and len(outs) == 3
    l0 = loss_mod.seg_cross_entropy(outs[0], tgt)
    l1 = loss_mod.seg_cross_entropy(outs[1], tgt)            # computed, not part of the objective
    assert float(l1) > 0
    l0.backward()
    aux = [p for n, p in mod.named_parameters() if n.startswith('aux16')]
    assert aux and all(float(p.grad.abs().max()) == 0 for p in aux)
    main = [p for n, p in mod.named_parameters() if n.startswith('out')]
    assert main and any(float(p.grad.abs().max()) > 0 for p in main)


def _psp_head():
    from multiyolov5_amd.models import yolo as Y
    from multiyolov5_amd.utils.torch_utils import initialize_weights
    from tests.test_gpu_ops import _randomize
    torch.manual_seed(4)
    mod = Y.SegMaskPSP(19, 1, 64, False, [64, 128, 256])
    initialize_weights(mod)
    _randomize(mod)
    g = torch.Generator().manual_seed(9)
    xs = [torch.randn(s, generator=g) for s in [(2, 64, 16, 32), (2, 128, 8, 16), (2, 256, 4, 8)]]
    tgt = torch.randint(-1, 19, (2, 128, 256), generator=g).to(DEV)
    return mod.to(DEV).train(), xs, tgt


def test_training_logits_are_materialised_only_on_demand(monkeypatch):
    """models/yolo.py:163's x8-upsampled logits in training mode: the fused loss never needs them (no upsample launch), any other use
    runs the deferred launch first and sees exactly the values of the eager path; a tensor of an older forward raises"""
    from multiyolov5_amd import _lib as L, engine as E, runtime as R
    from multiyolov5_amd.utils import loss as loss_mod
    mod, xs_cpu, tgt = _psp_head()
    xs = [x.to(DEV).requires_grad_() for x in xs_cpu]
    out = mod(xs)
    assert isinstance(out, R.LazySegLogits) and tuple(out.shape) == (2, 19, 128, 256) and out.dtype == torch.float32
    st = out._myolo_lazy_state
    assert st['done'] is False
    loss = loss_mod.seg_cross_entropy(out, tgt)
    loss.backward()
    assert st['done'] is False                                    # the whole step ran without the full-resolution tensor
    g_fused = [x.grad.clone() for x in xs]
    # second forward: now read the logits like any tensor
    xs2 = [x.to(DEV).requires_grad_() for x in xs_cpu]
    out2 = mod(xs2)
    with pytest.raises(L.MyoloError):
        out.sum()                                                 # first forward's logits: the plan has moved on
    vals = out2.detach().float().cpu()
    assert out2._myolo_lazy_state['done'] is True and type(vals) is torch.Tensor
    loss2 = torch.nn.functional.cross_entropy(out2, tgt, ignore_index=-1)     # a consumer that is NOT the fused loss
    loss2.backward()
    assert abs(float(loss2) - float(loss)) <= 1e-5 * abs(float(loss))
    for a, b in zip(g_fused, [x.grad for x in xs2]):
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max())
    # eager reference: a fresh module instance with the deferral switched off
    monkeypatch.setattr(E, 'LAZY_SEG', False)
    mod3, _, _ = _psp_head()
    out3 = mod3([x.to(DEV) for x in xs_cpu])
    assert not isinstance(out3, R.LazySegLogits)
    ref = out3.detach().float().cpu()                             # (two forwards differ in the last bits: the BatchNorm sums are fp32 atomics)
    assert float((ref - vals).abs().max()) <= 2e-5 * float(ref.abs().max())


def test_eval_logits_are_materialised_only_on_demand(monkeypatch):
    """eval (detect.py:144-193): the x8 upsample of the logits is deferred -- seg_argmax labels come from the low-resolution logits and
    equal the argmax of the materialised tensor, which equals the eager path's values bit for bit; an older frame's logits raise"""
    from multiyolov5_amd import _lib as L, engine as E, runtime as R
    from multiyolov5_amd.utils.general import seg_argmax
    mod, xs_cpu, _ = _psp_head()
    mod.eval()
    for _ in range(4):                                            # eager warm-up runs, capture, replay: every path returns a lazy tensor
        with torch.no_grad():
            out = mod([x.to(DEV) for x in xs_cpu])
        assert isinstance(out, R.LazySegLogits) and out._myolo_lazy_state['done'] is False
    lab = seg_argmax(out)
    assert out._myolo_lazy_state['done'] is False                 # labels without the full-resolution tensor
    vals = out.float().cpu()                                      # any other use materialises
    assert out._myolo_lazy_state['done'] is True
    from tests.test_gpu_model import assert_argmax_exact_or_near_tie
    assert_argmax_exact_or_near_tie('eval lazy logits', lab.cpu(), vals.argmax(1), vals, 1e-6)
    with torch.no_grad():
        out_b = mod([(x * 0.5).to(DEV) for x in xs_cpu])          # never read ...
        out_c = mod([(x * 0.25).to(DEV) for x in xs_cpu])
    with pytest.raises(L.MyoloError):
        out_b.sum()                                               # ... until the plan has moved on to the next frame: no stale data is served
    assert torch.isfinite(out_c.float()).all()
    monkeypatch.setattr(E, 'LAZY_SEG_EVAL', False)
    mod2, _, _ = _psp_head()
    mod2.eval()
    with torch.no_grad():
        ref = mod2([x.to(DEV) for x in xs_cpu])
    assert not isinstance(ref, R.LazySegLogits)
    refv = ref.float().cpu()                                      # (the pyramid's average pools sum through fp32 atomics: last-bit differences)
    assert float((refv - vals).abs().max()) <= 2e-5 * float(refv.abs().max())


def test_train_mode_without_autograd_returns_real_logits(monkeypatch):
    """`model.train(); with torch.no_grad(): model(x)` (BatchNorm recalibration, train-mode validation): no LazySegLogits wrapper is
    there to trigger the deferred x8 upsample, so the forward itself must run it -- the logits equal the eager path's"""
    from multiyolov5_amd import engine as E, runtime as R
    mod, xs_cpu, _ = _psp_head()
    with torch.no_grad():
        out = mod([x.to(DEV) for x in xs_cpu])
    assert not isinstance(out, R.LazySegLogits) and tuple(out.shape) == (2, 19, 128, 256)
    vals = out.detach().float().cpu().clone()
    assert torch.isfinite(vals).all()
    monkeypatch.setattr(E, 'LAZY_SEG', False)
    mod2, _, _ = _psp_head()
    with torch.no_grad():
        ref = mod2([x.to(DEV) for x in xs_cpu]).detach().float().cpu()
    assert float((ref - vals).abs().max()) <= 2e-5 * float(ref.abs().max())
    # and a second no_grad forward on new data overwrites the same storage with the new values (nothing stale is served)
    with torch.no_grad():
        out_b = mod([(x * 0.5).to(DEV) for x in xs_cpu]).detach().float().cpu()
    assert float((out_b - vals).abs().max()) > 1e-3


def test_graph_train_segmentation_gradients_match_eager(monkeypatch):
    """MYOLO_GRAPH_TRAIN=1: the captured backward bakes in the full-resolution branch of the segmentation gradient switch, so the
    fused low-resolution loss must not be offered -- head gradients of graph replays (step 3 onwards) equal the eager ones"""
    from multiyolov5_amd import engine as E
    from multiyolov5_amd.utils import loss as loss_mod

    def grads(graph):
        monkeypatch.setattr(E, 'GRAPH_TRAIN', graph)
        mod, xs_cpu, tgt = _psp_head()
        out_g = None
        for _ in range(4):                          # two eager warm-up runs, capture, replay
            for p in mod.parameters():
                p.grad = None
            xs = [x.to(DEV).requires_grad_() for x in xs_cpu]
            loss = loss_mod.seg_cross_entropy(mod(xs), tgt)
            loss.backward()
            out_g = [x.grad.clone() for x in xs] + [p.grad.clone() for p in mod.parameters()]
        torch.cuda.synchronize()
        return float(loss), out_g

    l0, g0 = grads(False)
    l1, g1 = grads(True)
    assert abs(l0 - l1) <= 1e-5 * abs(l0)
    assert any(float(g.abs().max()) > 0 for g in g1[3:])
    for a, b in zip(g1, g0):
        den = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) / den <= 5e-4, float((a - b).abs().max()) / den


@pytest.mark.parametrize('thresh', [0.7, 0.05], ids=['above_thresh', 'topk'])
def test_fused_upsample_ohem_matches_the_unfused_path(monkeypatch, thresh):
    """OhemCELoss (loss.py:303-328) over the low-resolution logits (per-pixel losses -> device selection -> gradient of the hard pixels
    folded into the low map, no full-resolution logits) against the unfused path on the materialised logits: same loss, same gradients.
    thresh 0.7 (train.py:287): enough pixels above -log(0.7) -> mean over those; thresh 0.05: fewer than n_min above -log(0.05) ~ 3.0 ->
    the top-k branch (loss.py:325-326) with its tie handling"""
    from multiyolov5_amd.utils import loss as loss_mod

    def run(fused):
        monkeypatch.setattr(loss_mod, 'FUSED_UPCE', fused)
        mod, xs_cpu, tgt = _psp_head()
        xs = [x.to(DEV).requires_grad_() for x in xs_cpu]
        crit = loss_mod.OhemCELoss(thresh=thresh)
        out = mod(xs)
        loss = crit(out, tgt)
        loss.backward()
        torch.cuda.synchronize()
        lazy_done = getattr(out, '_myolo_lazy_state', {'done': None})['done']
        return float(loss), [x.grad.clone() for x in xs] + [p.grad.clone() for p in mod.parameters()], lazy_done

    l1, g1, done1 = run(True)
    l0, g0, done0 = run(False)
    assert done1 is False and done0 is True            # the fused path never materialised the x8 logits, the unfused one had to
    assert abs(l1 - l0) <= 2e-6 * abs(l0), (l1, l0)
    for a, b in zip(g1, g0):
        den = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) / den <= 5e-4, float((a - b).abs().max()) / den
