"""CPU baseline leg of bench.py (TEST INFRASTRUCTURE: the oracle is timed here, never shipped or used by the product).

`joint_step` times the CPU restatement of the reference path (oracle.model_ref + oracle.loss_ref, torch CPU fp32 --
the same ATen CPU kernels the reference itself would run) on the host cores of the box: forward of a
[B,3,H,W] batch, ComputeLoss + segmentation CE, backward.  kind = "port" (the reference itself is not on the GPU box).
"""
import os
import time

import torch
import yaml

from . import loss_ref, model_ref, shapes, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def joint_step(cfg='yolov5s_city_seg.yaml', batch=2, H=512, W=1024, budget_s=20.0, max_threads=16):
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, max_threads))          # torch CPU convs stop scaling (and can collapse) far below 256 threads
    torch.set_num_threads(cores)
    with open(os.path.join(ROOT, 'multiyolov5_amd', 'cfg', cfg)) as f:
        cfgd = yaml.safe_load(f)
    sd = synth.synth_state_dict(shapes.template_state_dict(cfgd), 0)
    x = synth.synth_images(batch, H, W, seed=1)
    targets = synth.synth_det_targets(batch, 8, 10, seed=1)
    mask = synth.synth_seg_targets(batch, H, W, 19, seed=1)
    hyp = loss_ref.scaled_hyp(max(H, W), 10, 3)
    anchors = sd['model.25.anchors']

    def step():
        params = {k: v.detach().clone().requires_grad_() for k, v in sd.items()
                  if v.dtype.is_floating_point and 'running' not in k and 'anchor' not in k}
        sdt = {k: (params[k] if k in params else v.clone()) for k, v in sd.items()}
        det, seg = model_ref.forward(cfgd, sdt, x, training=True, dropout_p=0.0)
        loss, _ = loss_ref.compute_loss(det, targets, anchors, hyp)
        segs = seg if isinstance(seg, list) else [seg]
        sl = loss_ref.seg_ce(segs[0], mask)
        (loss * 0.6 + sl * batch * 0.35).backward()

    t0 = time.perf_counter()
    step()                                   # warm-up (allocator, thread pool)
    warm = time.perf_counter() - t0
    if warm > budget_s:                      # bounded sample: a box this slow gets the single warm-up step as its sample
        return {'value': batch / warm, 'unit': 'images/s', 'cores': cores, 'kind': 'port',
                'sample': f'1 joint step (the warm-up, {warm:.1f} s) of {cfg} at {batch}x3x{H}x{W} fp32, torch CPU, {cores} threads'}
    t0 = time.perf_counter()
    n = 0
    best = float('inf')
    while True:
        t1 = time.perf_counter()
        step()
        dt = time.perf_counter() - t1
        best = min(best, dt)
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 10:
            break
    return {'value': batch / best, 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} joint steps (fwd + ComputeLoss + seg CE + bwd) of {cfg} at {batch}x3x{H}x{W} fp32, torch CPU, '
                      f'{cores} threads, best of {n} after 1 warm-up ({best * 1e3:.0f} ms/step)'}
