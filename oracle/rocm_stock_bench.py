"""Stock PyTorch-ROCm baseline (TEST INFRASTRUCTURE -- a measurement of the reference's own execution model, never used by
the product): the oracle's restatement of the reference graph (oracle.model_ref / oracle.loss_ref = the same ATen ops the
reference modules call) is run on the GPU through ATen + MIOpen under torch.autocast(fp16), with torch.optim.SGD,
torch.amp.GradScaler and a foreach EMA -- i.e. what `train.py` / `detect.py` of the reference would execute on an MI355X
without this library (SURVEY.md 8(d) "unmodified reference on MI355X" row).

usage: python -m oracle.rocm_stock_bench [--batch 16] [--steps 10] [--channels-last]   (prints one JSON line per leg)
"""
import argparse
import json
import os
import sys
import time

import torch
import yaml

from . import loss_ref, model_ref, shapes, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', default='yolov5s_city_seg.yaml')
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--img', type=int, nargs=2, default=(512, 1024))
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--channels-last', action='store_true')
    ap.add_argument('--no-train', action='store_true')
    ap.add_argument('--no-infer', action='store_true')
    ap.add_argument('--device', default='cuda:0', help="'cpu' = dry run of the script logic (fp32, no autocast)")
    a = ap.parse_args()
    dev = torch.device(a.device)
    gpu = dev.type == 'cuda'
    sync = torch.cuda.synchronize if gpu else (lambda: None)
    torch.backends.cudnn.benchmark = False            # MIOpen immediate mode: no exhaustive find on a cold box
    with open(os.path.join(ROOT, 'multiyolov5_amd', 'cfg', a.cfg)) as f:
        cfg = yaml.safe_load(f)
    H, W = a.img
    B = a.batch
    sd0 = synth.synth_state_dict(shapes.template_state_dict(cfg), 0)
    mf = torch.channels_last if a.channels_last else torch.contiguous_format

    def to_dev(v):
        v = v.to(dev)
        return v.contiguous(memory_format=mf) if v.dim() == 4 else v

    if not a.no_train:
        sd = {k: to_dev(v) for k, v in sd0.items()}
        names = [k for k, v in sd.items() if v.dtype.is_floating_point and 'running' not in k and 'anchor' not in k]
        for k in names:
            sd[k].requires_grad_()
        params = [sd[k] for k in names]
        decay = [sd[k] for k in names if k.endswith('.weight') and sd[k].dim() == 4]
        other = [sd[k] for k in names if not (k.endswith('.weight') and sd[k].dim() == 4)]
        opt = torch.optim.SGD([{'params': other}, {'params': decay, 'weight_decay': 0.0005 * B / 64 * max(round(64 / B), 1)}],
                              lr=0.0015, momentum=0.937, nesterov=True)
        scaler = torch.amp.GradScaler('cuda', enabled=gpu)
        ema = [p.detach().clone() for p in params]
        x = to_dev(synth.synth_images(B, H, W, seed=1))
        targets = synth.synth_det_targets(B, 8, 10, seed=1).to(dev)
        mask = synth.synth_seg_targets(B, H, W, 19, seed=1).to(dev)
        hyp = loss_ref.scaled_hyp(max(H, W), 10, 3)
        anchors = sd['model.25.anchors']

        def step():
            with torch.autocast('cuda', dtype=torch.float16, enabled=gpu):
                det, seg = model_ref.forward(cfg, sd, x, training=True, dropout_p=0.0)
                loss, _ = loss_ref.compute_loss(det, targets, anchors, hyp)
                segs = seg if isinstance(seg, list) else [seg]
                sl = loss_ref.seg_ce(segs[0], mask) * B
            scaler.scale(loss * 0.6 + sl * 0.35).backward()
            scaler.step(opt)
            scaler.update()
            opt.zero_grad(set_to_none=True)
            with torch.no_grad():
                torch._foreach_mul_(ema, 0.999)
                torch._foreach_add_(ema, [p.detach() for p in params], alpha=0.001)

        t0 = time.perf_counter()
        for _ in range(a.warmup):
            step()
        sync()
        warm = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        sync()
        dt = (time.perf_counter() - t0) / a.steps
        print(json.dumps({'leg': 'train', 'images_per_s': B / dt, 'ms_per_step': dt * 1e3, 'batch': B, 'img': [H, W],
                          'channels_last': a.channels_last, 'warmup_s': warm,
                          'what': 'oracle graph on ATen/MIOpen, autocast fp16, torch SGD + GradScaler + foreach EMA'}), flush=True)
        del sd, params, opt, ema

    if not a.no_infer:
        sdf = {k: ((to_dev(v).half() if gpu else to_dev(v)) if v.dtype.is_floating_point else v.to(dev)) for k, v in model_ref.fuse_state_dict(sd0).items()}
        x = to_dev(synth.synth_images(1, 2 * H, 2 * W, seed=2))
        x = x.half() if gpu else x

        def infer():
            with torch.no_grad():
                det, seg = model_ref.forward(cfg, sdf, x, training=False)
                segs = seg if isinstance(seg, list) else [seg]
                return det, segs[0].argmax(1)

        t0 = time.perf_counter()
        for _ in range(a.warmup):
            infer()
        sync()
        warm = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(a.steps * 3):
            infer()
        sync()
        dt = (time.perf_counter() - t0) / (a.steps * 3)
        print(json.dumps({'leg': 'infer', 'fps': 1 / dt, 'ms': dt * 1e3, 'img': [2 * H, 2 * W], 'channels_last': a.channels_last,
                          'warmup_s': warm, 'what': 'fused fp16 forward + seg argmax (no NMS) on ATen/MIOpen'}), flush=True)


if __name__ == '__main__':
    sys.exit(main())
