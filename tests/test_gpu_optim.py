"""-m gpu: multi-tensor optimizer kernels (myolo_mt_sgd / _check_finite / _ema, myolo_scaler_update) against the
torch CPU implementations the reference uses (torch.optim.SGD train.py:133, ModelEMA torch_utils.py:290-300,
torch.cuda.amp.GradScaler semantics train.py:265,397-398)."""
import math

import pytest
import torch

from tests.gpu_util import check

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
SHAPES = [(64, 32, 3, 3), (64,), (64,), (128, 64, 1, 1), (128,), (45, 128, 1, 1), (45,), (70000,), (1,)]


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g) for s in SHAPES]


def test_fused_sgd_matches_torch_sgd_over_steps():
    from multiyolov5_amd.utils.optim import FusedSGD
    ref = [torch.nn.Parameter(p.clone()) for p in _params(0)]
    got = [torch.nn.Parameter(p.clone().to(DEV)) for p in _params(0)]
    groups = lambda ps: [{'params': ps[1:3]}, {'params': [ps[0], ps[3], ps[5], ps[7]], 'weight_decay': 5e-4},
                         {'params': [ps[4], ps[6], ps[8]]}]
    o_ref = torch.optim.SGD(groups(ref), lr=0.0015, momentum=0.937, nesterov=True)
    o_got = FusedSGD(groups(got), lr=0.0015, momentum=0.937, nesterov=True)
    for step in range(4):
        grads = _params(10 + step)
        for j, (a, b) in enumerate(zip(ref, got)):
            a.grad = grads[j].clone()
            b.grad = grads[j].clone().to(DEV)
        if step == 2:                                    # warm-up style per-group rewrite (train.py:344-352)
            for o in (o_ref, o_got):
                o.param_groups[2]['lr'] = 0.05
                o.param_groups[1]['momentum'] = 0.85
        o_ref.step()
        o_got.step()
    for j, (a, b) in enumerate(zip(ref, got)):
        check(f'sgd/param{j}', b, a, 1e-6)
        check(f'sgd/buf{j}', o_got.state[b]['momentum_buffer'], o_ref.state[a]['momentum_buffer'], 1e-6)


def test_grad_scaler_unscale_skip_and_growth():
    from multiyolov5_amd.utils.optim import FusedSGD, GradScaler
    p = torch.nn.Parameter(torch.ones(5000, device=DEV))
    opt = FusedSGD([p], lr=0.1, momentum=0.0)
    sc = GradScaler(init_scale=1024.0, growth_interval=2)
    loss = (p * 2.0).sum()
    sc.scale(loss).backward()
    assert abs(float(p.grad[0]) - 2048.0) < 1e-3
    sc.step(opt); sc.update(); opt.zero_grad()
    check('scaler/step1', p, torch.full((5000,), 0.8), 1e-6)                  # unscaled gradient 2.0 * lr 0.1
    assert sc.get_scale() == 1024.0
    # inf gradient: update skipped, scale halves
    sc.scale((p * 2.0).sum()).backward()
    p.grad[123] = float('inf')
    sc.step(opt); sc.update(); opt.zero_grad()
    check('scaler/skip', p, torch.full((5000,), 0.8), 1e-6)
    assert sc.get_scale() == 512.0
    # two clean steps -> growth
    for _ in range(2):
        sc.scale((p * 2.0).sum()).backward()
        sc.step(opt); sc.update(); opt.zero_grad()
    assert sc.get_scale() == 1024.0
    check('scaler/after', p, torch.full((5000,), 0.4), 1e-5)
    # nan also detected
    sc.scale((p * 2.0).sum()).backward()
    p.grad[4999] = float('nan')
    sc.step(opt); sc.update()
    assert sc.get_scale() == 512.0 and bool(torch.isfinite(p).all())


def test_model_ema_matches_reference_formula():
    from multiyolov5_amd.models.common import Conv
    from multiyolov5_amd.utils.torch_utils import ModelEMA, initialize_weights
    torch.manual_seed(0)
    net = torch.nn.Sequential(Conv(16, 32, 3), Conv(32, 32, 1)).to(DEV)
    initialize_weights(net)
    ema = ModelEMA(net)
    ref = {k: v.detach().cpu().clone() for k, v in ema.ema.state_dict().items()}
    for u in range(1, 4):
        with torch.no_grad():
            for p in net.parameters():
                p.add_(torch.randn_like(p) * 0.1)
            for b in net.buffers():
                if b.dtype.is_floating_point:
                    b.add_(0.05)
                else:
                    b.add_(1)
        ema.update(net)
        d = 0.9999 * (1 - math.exp(-u / 2000))
        for k, v in net.state_dict().items():
            if v.dtype.is_floating_point:
                ref[k] = ref[k] * d + (1 - d) * v.detach().cpu()
    assert ema.updates == 3
    for k, v in ema.ema.state_dict().items():
        if v.dtype.is_floating_point:
            check(f'ema/{k}', v, ref[k], 1e-6)
        else:
            assert int(v) == int(ref[k])                      # integer buffers are not averaged (torch_utils.py:297)


def test_eval_plan_follows_fused_ema_and_sgd_updates():
    """train.py evaluates ema.ema with test.py after every epoch: the multi-tensor EMA / SGD kernels write the parameters through raw
    pointers, so they must move autograd's version counters themselves -- an eval plan built in epoch 1 re-derives its packed weights and
    folded BatchNorm constants in epoch 2 (round 4: it did not, the second evaluation ran on the first epoch's weights)"""
    import os
    from multiyolov5_amd.models.yolo import Model
    from multiyolov5_amd.utils.optim import FusedSGD
    from multiyolov5_amd.utils.torch_utils import ModelEMA
    from tests.util import CFG, TAGS, synth_sd
    m = Model(os.path.join(CFG, TAGS['s_psp']))
    m.load_state_dict(synth_sd('s_psp'), strict=True)
    m = m.to(DEV)
    ema = ModelEMA(m)
    x = torch.rand(1, 3, 64, 128, device=DEV)
    with torch.no_grad():
        (p0, _), s0 = ema.ema(x)
        p0, s0 = p0.clone(), s0.clone()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.5)                                      # "training" moved the weights
    ema.updates = 100000                                     # decay ~ 0.9999: force a visible step instead
    ema.decay = lambda n: 0.5
    ema.update(m)
    with torch.no_grad():
        (p1, _), s1 = ema.ema(x)
    ref = ModelEMA(m)                                        # the same EMA weights in a fresh module (fresh plan)
    ref.ema.load_state_dict(ema.ema.state_dict())
    with torch.no_grad():
        (pr, _), sr = ref.ema(x)
    assert float((s1 - s0).abs().max()) > 1e-3, 'the EMA step changed the outputs of the evaluated model'
    torch.testing.assert_close(s1, sr, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(p1, pr, rtol=1e-5, atol=1e-5)
    # FusedSGD: an eval forward of the trained model itself after a step
    m.eval()
    with torch.no_grad():
        _, a0 = m(x)
        a0 = a0.clone()
    opt = FusedSGD(m.parameters(), lr=0.5, momentum=0.0, nesterov=False)
    for p in m.parameters():
        p.grad = torch.ones_like(p) * 1e-2
    opt.step()
    with torch.no_grad():
        _, a1 = m(x)
    assert float((a1 - a0).abs().max()) > 1e-4, 'the optimizer step is visible to the eval plan'
