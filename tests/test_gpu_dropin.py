"""-m gpu: INTEGRATION.md section 1 exercised end to end -- the body of the reference's training loop (train.py:243-245, 265,
363-401) with STOCK torch.optim.SGD, torch.cuda.amp.autocast / GradScaler and a world-size-1 DistributedDataParallel wrapper over
the mirror, then the checkpoint block (train.py:481-499) and detect.py's attempt_load (experimental.py:114-134)."""
import os
import socket
from copy import deepcopy

import pytest
import torch
import torch.distributed as dist

from oracle import loss_ref
from tests.gpu_util import check
from tests.util import CFG, TAGS

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _world1():
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', WORLD_SIZE='1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(DEV))       # backend 'nccl' == RCCL (train.py:619)


def _groups(model):                                            # train.py:121-137
    pg0, pg1, pg2 = [], [], []
    for k, v in model.named_modules():
        if hasattr(v, 'bias') and isinstance(v.bias, torch.nn.Parameter):
            pg2.append(v.bias)
        if isinstance(v, torch.nn.BatchNorm2d):
            pg0.append(v.weight)
        elif hasattr(v, 'weight') and isinstance(v.weight, torch.nn.Parameter):
            pg1.append(v.weight)
    return pg0, pg1, pg2


def test_train_py_loop_body_with_stock_sgd_amp_ddp_then_checkpoint_roundtrip(tmp_path):
    """runs in its own process (like the gloo tests): a process group / RCCL communicator and the sys.modules rebinding are process
    state, and tests that follow an in-process RCCL init in the same interpreter were observed to fail at random"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (f'import sys; sys.path.insert(0, {root!r}); import tests.test_gpu_dropin as t; import pathlib; '
            f't._ddp_loop_body(pathlib.Path({str(tmp_path)!r})); print("__DDP_BODY_OK__")')
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and '__DDP_BODY_OK__' in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]


def _ddp_loop_body(tmp_path):
    import multiyolov5_amd.dropin as _d
    _d.install()
    _world1()
    from torch.cuda import amp
    from torch.nn.parallel import DistributedDataParallel as DDP
    from models.yolo import Model                                                     # train.py:20 under the rebinding
    from models.experimental import attempt_load                                      # detect.py:13
    from multiyolov5_amd import synth
    from multiyolov5_amd.utils.loss import ComputeLoss, SegmentationLosses           # what utils.loss.* are rebound to
    from multiyolov5_amd.utils.torch_utils import ModelEMA, is_parallel
    assert Model.__module__ == 'models.yolo'
    H, W, B = 256, 512, 4
    torch.manual_seed(0)
    model = Model(os.path.join(CFG, TAGS['s_psp'])).to(DEV)
    synth.randomize_(model, seed=0)
    pg0, pg1, pg2 = _groups(model)
    optimizer = torch.optim.SGD(pg0, lr=0.01, momentum=0.937, nesterov=True)          # train.py:133-137 (stock optimizer)
    optimizer.add_param_group({'params': pg1, 'weight_decay': 5e-4})
    optimizer.add_param_group({'params': pg2})
    ema = ModelEMA(model)
    model = DDP(model, device_ids=[0], output_device=0)                               # train.py:243-245
    nl = model.module.model[-1].nl
    model.module.nc, model.module.gr = 10, 1.0
    model.module.hyp = loss_ref.scaled_hyp(W, 10, nl)
    scaler = amp.GradScaler(enabled=True, init_scale=1024.0)                          # train.py:265
    compute_loss, compute_seg_loss = ComputeLoss(model.module), SegmentationLosses()
    imgs = synth.images(B, H, W, seed=1).to(DEV)                                      # fp32 in [0,1] as the loaders deliver (train.py:342)
    segimgs = synth.images(B, H, W, seed=2).to(DEV)
    targets = synth.det_targets(B, 8, 10, seed=1).to(DEV)
    segtargets = synth.seg_targets(B, H, W, 19, seed=2).to(DEV)
    w0 = model.module.model[1].conv.weight.detach().clone()
    model.train()
    # stock DDP learns that a gradient is final from the parameter's AccumulateGrad hook: with the backward cut into a chain of autograd
    # nodes (runtime.PlanStageFn) the head's hooks must fire BEFORE the launches of the last stage (the backbone) are enqueued
    from multiyolov5_amd import runtime as _R
    order = []
    _orig_stage = _R.PlanStageFn.backward

    def _logged(ctx, *g):
        order.append(('stage', ctx.k))
        return _orig_stage(ctx, *g)
    _R.PlanStageFn.backward = staticmethod(_logged)
    head_p, stem_p = model.module.model[25].m[0].weight, model.module.model[0].conv.conv.weight
    head_p.register_post_accumulate_grad_hook(lambda p: order.append(('hook', 'head')))
    stem_p.register_post_accumulate_grad_hook(lambda p: order.append(('hook', 'stem')))
    accumulate, seen = 2, []
    for ni in range(1, 5):
        with amp.autocast(enabled=True):                                              # train.py:363-370
            pred = model(imgs)
            loss, loss_items = compute_loss(pred[0], targets)
            loss *= 1                                                                 # opt.world_size
            loss *= 0.6
        scaler.scale(loss).backward()                                                 # train.py:371
        with amp.autocast(enabled=True):                                              # train.py:380-391
            pred = model(segimgs)
            segloss = compute_seg_loss(pred[1], segtargets) * B
            segloss *= 0.35
        scaler.scale(segloss).backward()                                              # train.py:392
        assert all(p.grad is not None and p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all() for p in model.parameters())
        if ni % accumulate == 0:                                                      # train.py:396-401
            scaler.step(optimizer)
            scaler.update()
            optimizer.zero_grad()
            ema.update(model)
        seen.append((float(loss.detach()), float(segloss.detach())))
    assert scaler.get_scale() == 1024.0                                               # no step was skipped for inf/nan
    first = order[:order.index(('hook', 'stem')) + 1]                                 # the first backward of the run
    stages = [k for kind, k in first if kind == 'stage']
    assert stages == [0, 1, 2], order[:12]
    assert first.index(('hook', 'head')) < first.index(('stage', 1)) < first.index(('stage', 2)) < first.index(('hook', 'stem')), first
    assert float((model.module.model[1].conv.weight - w0).abs().max()) > 0
    assert seen[-1][1] < seen[0][1] and all(torch.isfinite(torch.tensor(s)).all() for s in seen)
    # ---- checkpoint block, train.py:481-499 ----
    ckpt = {'epoch': 0, 'best_fitness': 0.0, 'training_results': '',
            'model': deepcopy(model.module if is_parallel(model) else model).half(),
            'ema': deepcopy(ema.ema).half(), 'updates': ema.updates, 'optimizer': optimizer.state_dict(), 'wandb_id': None}
    last = str(tmp_path / 'last.pt')
    torch.save(ckpt, last)
    del ckpt
    from tests.test_dropin_cpu import _globals_of
    assert not any(m.startswith('multiyolov5_amd') for m, _ in _globals_of(last))
    # ---- detect.py:34,103: attempt_load(...).half(); eval outputs == the live EMA model's ----
    m2 = attempt_load(last, map_location=DEV)
    assert type(m2).__module__ == 'models.yolo' and not m2.training
    m2 = m2.half()
    live = deepcopy(ema.ema).half().float().fuse().eval().half()
    x = synth.images(1, H, W, seed=5).to(DEV).half()
    with torch.no_grad():
        (p2, _), s2 = m2(x)
        (p1, _), s1 = live(x)
    check('dropin/ckpt/pred', p2, p1, 1e-5)
    check('dropin/ckpt/seg', s2, s1, 1e-5)
    dist.destroy_process_group()


def test_stock_sgd_step_equals_fused_sgd_step():
    """two loss-scaled joint steps from the same weights: stock torch.optim.SGD + torch.cuda.amp.GradScaler vs the library's
    FusedSGD + GradScaler (one multi-tensor launch, unscale folded in): same parameter update.  Run in fp32 arithmetic (autocast
    off) so that the two runs are comparable at 1e-3 -- two fp16 runs of this random-weight network differ by 5-20 % per gradient
    tensor from the storage noise alone (atomics reorder the BatchNorm sums), which says nothing about the optimizers."""
    from torch.cuda import amp
    from multiyolov5_amd import synth
    from multiyolov5_amd.models.yolo import Model
    from multiyolov5_amd.utils.loss import ComputeLoss, SegmentationLosses
    from multiyolov5_amd.utils.optim import FusedSGD, GradScaler
    H, W, B = 128, 256, 2
    res = []
    for kind in ('stock', 'fused'):
        torch.manual_seed(0)
        m = Model(os.path.join(CFG, TAGS['s_psp'])).to(DEV)
        synth.randomize_(m, seed=0)
        m.train()
        m.nc, m.gr, m.hyp = 10, 1.0, loss_ref.scaled_hyp(W, 10, 3)
        pg0, pg1, pg2 = _groups(m)
        groups = [{'params': pg0}, {'params': pg1, 'weight_decay': 5e-4}, {'params': pg2}]
        if kind == 'stock':
            opt, scaler = torch.optim.SGD(groups, lr=0.01, momentum=0.937, nesterov=True), amp.GradScaler(init_scale=256.0)
        else:
            opt, scaler = FusedSGD(groups, lr=0.01, momentum=0.937, nesterov=True), GradScaler(init_scale=256.0)
        x = synth.images(B, H, W, seed=1).to(DEV)
        t = synth.det_targets(B, 8, 10, seed=1).to(DEV)
        mk = synth.seg_targets(B, H, W, 19, seed=1).to(DEV)
        w0 = {k: p.detach().clone() for k, p in m.named_parameters()}
        for _ in range(2):
            with amp.autocast(enabled=False):
                det, seg = m(x)
                loss, _ = ComputeLoss(m)(det, t)
                sl = SegmentationLosses()(seg, mk) * B
            scaler.scale(loss * 0.6 + sl * 0.35).backward()
            scaler.step(opt); scaler.update(); opt.zero_grad()
        assert scaler.get_scale() == 256.0
        res.append({k: (p.detach() - w0[k]).float() for k, p in m.named_parameters()})
    bad = []
    for k in res[0]:
        check(f'dropin/sgd/{k}', res[1][k], res[0][k], 6e-3, collect=bad)      # two fp32 runs: measured <= 2.2e-3 (atomics reorder the sums)
    assert not bad, '\n'.join(bad[:10])


def test_gradient_accumulation_over_two_backward_passes_equals_the_sum(monkeypatch):
    """train.py:364-401: a detection backward and a segmentation backward accumulate into `.grad` before one optimizer step.  The
    second backward adds its flat gradient buffer into the first one's in ONE kernel (runtime._accumulate_in_place) -- same values
    as autograd's per-parameter accumulation, and any `.grad` the caller replaced switches back to that path"""
    from multiyolov5_amd import engine as E, runtime as R
    from multiyolov5_amd.models.yolo import Model
    from multiyolov5_amd.utils.loss import ComputeLoss, SegmentationLosses
    from oracle import loss_ref, synth
    from tests.util import CFG, TAGS, synth_sd
    monkeypatch.setattr(E, 'TINY_CONV', False)           # (accumulation == sum is an exactness statement; the tiny launches' arrival-order noise stays out)
    torch.manual_seed(0)
    m = Model(os.path.join(CFG, TAGS['s_psp']))
    m.load_state_dict(synth_sd('s_psp'), strict=True)
    m = m.to(DEV).train()
    m.hyp, m.gr, m.nc = loss_ref.scaled_hyp(imgsz=128, nc=10, nl=3), 1.0, 10
    x = synth.synth_images(2, 64, 128, seed=1).to(DEV)
    targets = synth.synth_det_targets(2, 8, 10, seed=1).to(DEV)
    mask = synth.synth_seg_targets(2, 64, 128, 19, seed=1).to(DEV)
    cl, sl = ComputeLoss(m), SegmentationLosses()

    def two_passes():
        for p in m.parameters():
            p.grad = None
        det, seg = m(x)
        cl(det, targets)[0].backward()
        det, seg = m(x)
        (sl(seg, mask) * 2).backward()
        return [p.grad.clone() for p in m.parameters()]
    calls = []
    orig = R._accumulate_in_place
    monkeypatch.setattr(R, '_accumulate_in_place', lambda h, p: calls.append(orig(h, p)) or calls[-1])
    fast = two_passes()
    assert calls == [False, True]                         # the second backward took the flat add
    monkeypatch.setattr(R, 'FLAT_ACCUMULATE', False)
    calls.clear()
    slow = two_passes()
    assert calls == [False, False]
    for a, b in zip(fast, slow):
        # two runs of the same step: bit-equal kernels except for the arrival order of the statistics atomics (tiny launches off: 1e-5)
        assert float((a - b).abs().max()) <= 2e-5 * (float(b.abs().max()) + 1e-12)
    # a replaced .grad (not a view of the flat buffer) falls back to autograd's accumulation
    monkeypatch.setattr(R, 'FLAT_ACCUMULATE', True)
    for p in m.parameters():
        p.grad = None
    det, seg = m(x)
    cl(det, targets)[0].backward()
    first = next(m.parameters())
    first.grad = first.grad.clone()
    calls.clear()
    det, seg = m(x)
    (sl(seg, mask) * 2).backward()
    assert calls == [False]
    for a, p in zip(slow, m.parameters()):
        assert float((a - p.grad).abs().max()) <= 2e-5 * (float(a.abs().max()) + 1e-12)      # (two runs: see above)
