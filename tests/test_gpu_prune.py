"""-m gpu: a backward that starts from ONE of the model's outputs (train.py:371 = the detection pass, train.py:392 = the segmentation
pass) runs the pruned launch list (engine.Plan.bwd_schedule).  Its gradients must equal the full list's (which runs the other head
over zero gradients) and the oracle's autograd on the same single loss -- also when the variants alternate on one plan, where a stale
gradient buffer of the previous backward would show."""
import os

import pytest
import torch

from oracle import loss_ref, model_ref, synth
from tests.gpu_util import check
from tests.util import CFG, TAGS, load_cfg, synth_sd, tie_free_images

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
H, W = 64, 128


def build(tag):
    from multiyolov5_amd.models.yolo import Model
    m = Model(os.path.join(CFG, TAGS[tag]))
    sd = synth_sd(tag)
    m.load_state_dict(sd, strict=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    hyp = loss_ref.scaled_hyp(imgsz=128, nc=10, nl=3)
    m.hyp, m.gr, m.nc = hyp, 1.0, 10
    return m.to(DEV).train(), sd, hyp


def losses(m, x, targets, mask, which):
    from multiyolov5_amd.utils.loss import ComputeLoss, SegmentationLosses
    det, seg = m(x)
    tot = 0.0
    if 'det' in which:
        tot = tot + ComputeLoss(m)(det, targets)[0] * 0.6
    if 'seg' in which:
        segs = seg if isinstance(seg, (list, tuple)) else [seg]
        for s_ in segs:
            tot = tot + SegmentationLosses()(s_, mask) * 0.7
    return tot


def grads_of(m):
    return {k: (p.grad.detach().float().cpu().clone() if p.grad is not None else None) for k, p in m.named_parameters()}


def run_sequence(tag, seq, prune, monkeypatch, dtype=torch.float32, staged=None):
    from multiyolov5_amd import engine as E, runtime as R
    monkeypatch.setattr(E, 'PRUNE_BWD', prune)
    monkeypatch.setattr(E, 'TINY_CONV', False)           # (pruned == full is an exactness statement; the tiny launches' LDS-atomics noise stays out)
    if staged is not None:
        monkeypatch.setattr(R, 'STAGED_BWD', staged)
    m, sd, hyp = build(tag)
    x, _ = tie_free_images(tag, 2, H, W)
    x = x.to(DEV, dtype)
    targets = synth.synth_det_targets(2, 8, 10, seed=1).to(DEV)
    mask = synth.synth_seg_targets(2, H, W, 19, seed=1).to(DEV)
    out = []
    for which in seq:
        m.zero_grad(set_to_none=True)
        m.load_state_dict({k: v.to(DEV) for k, v in sd.items()}, strict=True)         # same BatchNorm running stats every time
        with torch.autocast('cuda', enabled=dtype == torch.float16):
            loss = losses(m, x, targets, mask, which)
        (loss * (64.0 if dtype == torch.float16 else 1.0)).backward()        # (a fixed loss scale keeps fp16 gradients off the underflow edge)
        torch.cuda.synchronize()
        out.append((float(loss), grads_of(m)))
    plan = next(iter(m.__dict__['_plans'].values())).plan
    return out, plan, m


@pytest.mark.parametrize('tag', ['s_psp', 's_base', 's_lab', 's_bise', 'm_lab'])
def test_pruned_backward_equals_the_full_list_in_any_order(tag, monkeypatch):
    seq = [('det', 'seg'), ('det',), ('seg',), ('det',), ('det', 'seg'), ('seg',), ('seg',), ('det',)]
    a, plan, _ = run_sequence(tag, seq, True, monkeypatch)
    progs = plan.__dict__.get('_nprog_bwd', {})
    assert len({id(v) for v in progs.values() if v}) == 3, 'full, detection-only and segmentation-only programs were built'
    b, plan_b, _ = run_sequence(tag, seq, False, monkeypatch)
    assert len({id(v) for v in plan_b.__dict__.get('_nprog_bwd', {}).values() if v}) == 1
    bad = []
    for i, ((la, ga), (lb, gb)) in enumerate(zip(a, b)):
        assert abs(la - lb) <= 2e-6 * max(1.0, abs(lb))       # (two runs: statistics atomics in arrival order; tiny launches off)
        for k in ga:
            if gb[k].abs().max() == 0:
                assert ga[k].abs().max() == 0, f'step {i} {seq[i]}: {k} must stay zero'
            else:
                # (same kernels on the same values; BatchNorm / pooling sums go through atomics in arbitrary order)
                check(f'prune/{tag}/step{i}/{k}', ga[k], gb[k], 1e-3, collect=bad)
    assert not bad, '\n'.join(bad[:20])
    # the parameters only the other head reaches really are zero in a one-loss backward (and some are: the test would be vacuous otherwise)
    det_only, seg_only = a[1][1], a[2][1]
    assert any(v.abs().max() == 0 for v in det_only.values()) and any(v.abs().max() == 0 for v in seg_only.values())
    assert sum(float(v.abs().max()) > 0 for v in a[0][1].values()) > 0.95 * len(a[0][1])


@pytest.mark.parametrize('which', ['det', 'seg'])
def test_one_loss_backward_vs_oracle(which, monkeypatch):
    """the detection pass / the segmentation pass of train.py against the oracle's autograd on that loss alone (parameters the loss
    does not reach: no gradient in the oracle, zeros here)"""
    tag = 's_psp'
    out, plan, m = run_sequence(tag, [(which,)], True, monkeypatch)
    g = out[0][1]
    assert any(k[1] is not None for k in plan.__dict__['_nprog_bwd']), 'the pruned program ran'
    sd = synth_sd(tag)
    x, _ = tie_free_images(tag, 2, H, W)
    targets = synth.synth_det_targets(2, 8, 10, seed=1)
    mask = synth.synth_seg_targets(2, H, W, 19, seed=1)
    params = {k: v.clone().requires_grad_() for k, v in sd.items()
              if v.dtype.is_floating_point and 'running' not in k and 'anchor' not in k}
    sdt = {k: (params[k] if k in params else v.clone()) for k, v in sd.items()}
    rdet, rseg = model_ref.forward(load_cfg(tag), sdt, x, training=True, dropout_p=0.0)
    if which == 'det':
        (loss_ref.compute_loss(rdet, targets, sd['model.25.anchors'], m.hyp)[0] * 0.6).backward()
    else:
        (loss_ref.seg_ce(rseg, mask) * 0.7).backward()
    bad = []
    for k, p in params.items():
        if p.grad is None or p.grad.abs().max() == 0:
            assert g[k].abs().max() == 0, f'{k}: not reached by the {which} loss'
        else:
            check(f'prune_oracle/{which}/{k}', g[k], p.grad, 5e-3, collect=bad)
    assert not bad, '\n'.join(bad[:20])


def test_pruned_backward_in_the_staged_chain(monkeypatch):
    """stock-DDP form (the backward as a chain of autograd nodes, runtime.PlanStageFn: the program is cut at the stage marks, which
    sit at other op indices in a pruned program): pruned == full"""
    seq = [('det',), ('seg',), ('det', 'seg'), ('det',)]
    a, plan, m = run_sequence('s_psp', seq, True, monkeypatch, staged='force')
    h = next(iter(m.__dict__['_plans'].values()))
    assert h.__dict__.get('_stages', (None, None))[1], 'the staged chain ran'
    assert any(k[1] is not None for k in plan.__dict__['_nprog_bwd']), 'pruned programs were built for the staged chain'
    b, _, _ = run_sequence('s_psp', seq, False, monkeypatch, staged='force')
    bad = []
    for i, ((la, ga), (lb, gb)) in enumerate(zip(a, b)):
        for k in ga:
            if gb[k].abs().max() == 0:
                assert ga[k].abs().max() == 0, f'step {i}: {k}'
            else:
                check(f'prune_staged/step{i}/{k}', ga[k], gb[k], 1e-3, collect=bad)
    assert not bad, '\n'.join(bad[:20])


def test_pruned_backward_fp16_amp(monkeypatch):
    """the AMP dtype train.py runs (other kernels, same schedule): finite gradients, the same parameters untouched; values are pinned
    by the fp32 tests above (two fp16 runs of one list differ by their atomics-ordered BatchNorm sums)"""
    seq = [('det', 'seg'), ('det',), ('seg',)]
    a, plan, _ = run_sequence('s_psp', seq, True, monkeypatch, dtype=torch.float16)
    b, _, _ = run_sequence('s_psp', seq, False, monkeypatch, dtype=torch.float16)
    for i, ((la, ga), (lb, gb)) in enumerate(zip(a, b)):
        assert abs(la - lb) <= 2e-2 * max(1.0, abs(lb))
        rels = []
        for k in ga:
            assert torch.isfinite(ga[k]).all(), f'step {i}: {k}'
            assert (ga[k].abs().max() == 0) == (gb[k].abs().max() == 0), f'step {i}: {k} zero pattern'
            if gb[k].abs().max() > 0:
                rels.append(float((ga[k] - gb[k]).norm() / gb[k].norm()))
        rels.sort()
        assert rels[len(rels) // 2] < 0.2, f'step {i}: median relative difference {rels[len(rels) // 2]:.2f}'


def test_staged_chain_with_a_frozen_backbone(monkeypatch):
    """ADVICE r3 (runtime._stage_plan): train.py's `freeze` list sets requires_grad = False on the early backbone layers.  With every
    parameter of the deepest stage frozen (and images that need no gradient) autograd would never visit that stage -- the final join and
    the flat accumulation would be skipped and ALL gradients of the pass dropped.  The chain must fall back to one node: gradients of the
    live parameters equal the unfrozen run's, frozen ones stay None, a second backward still works."""
    from multiyolov5_amd import engine as E, runtime as R
    monkeypatch.setattr(R, 'STAGED_BWD', 'force')
    m, sd, hyp = build('s_psp')
    x, _ = tie_free_images('s_psp', 2, H, W)
    x = x.to(DEV)
    targets = synth.synth_det_targets(2, 8, 10, seed=1).to(DEV)
    mask = synth.synth_seg_targets(2, H, W, 19, seed=1).to(DEV)

    def one():
        m.zero_grad(set_to_none=True)
        m.load_state_dict({k: v.to(DEV) for k, v in sd.items()}, strict=True)
        losses(m, x, targets, mask, ('det', 'seg')).backward()
        torch.cuda.synchronize()
        return grads_of(m)
    ref = one()
    h = next(iter(m.__dict__['_plans'].values()))
    stages = h.__dict__['_stages'][1]
    assert stages and len(stages) >= 2
    last = stages[-1]['params']
    names = [k for k, _ in m.named_parameters()]
    plist = list(h.plan.params)
    frozen = {id(plist[i]) for i in last}
    for p in m.parameters():
        if id(p) in frozen:
            p.requires_grad_(False)
    for rep in range(2):
        g = one()
        assert h.__dict__['_stages'][1] is None, 'a stage of frozen parameters: the chain falls back to one autograd node'
        assert not h.pending_bwd
        bad = []
        for (k, p) in m.named_parameters():
            if id(p) in frozen:
                assert g[k] is None
            else:
                check(f'frozen_stage/{rep}/{k}', g[k], ref[k], 1e-3, collect=bad)
        assert not bad, '\n'.join(bad[:20])
    assert len(names) > len(frozen) > 0
