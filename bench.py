#!/usr/bin/env python
"""bench.py -- joint detection+segmentation training step of yolov5s_city_seg (PSP head) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" (BASELINE.json configs[1], SURVEY.md 8(d)(i) joint step): forward of a [16,3,512,1024] fp16 batch through
the libmyolo plan (Model.forward), ComputeLoss + segmentation cross-entropy, backward, loss-scaled fused SGD(nesterov)
+ EMA update.  Inputs are synthetic Cityscapes-shaped tensors already resident in HBM.  N>1: one process per GPU,
the same per-GPU batch (weak scaling), gradients all-reduced over RCCL by multiyolov5_amd.parallel.GradReducer.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field definitions).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F16_PEAK_TF = 2500.0      # dense fp16/bf16 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=16, help='per-GPU batch (configs[1]: 16)')
    ap.add_argument('--img', type=int, nargs=2, default=(512, 1024), help='H W')
    ap.add_argument('--cfg', default='yolov5s_city_seg.yaml')
    ap.add_argument('--dtype', default='f16', choices=['f16', 'f32'])
    ap.add_argument('--stage', default='train', choices=['train', 'fwdbwd', 'infer'],
                    help="dev only: 'fwdbwd' = model fwd+bwd with fixed output gradients (no loss/optimizer; NOT a valid "
                         "bench line), 'infer' = detect.py path FPS only")
    ap.add_argument('--infer-size', type=int, nargs=2, default=(1024, 2048), help="--stage infer: frame H W")
    ap.add_argument('--eval-fork', default=None, choices=['sem', 'event'],
                    help="--stage infer: how the segmentation head's stream is ordered behind the neck (runtime.EVAL_FORK; A/B of round 6's semaphore)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-infer', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-stock-baseline', action='store_true',
                    help='skip the stock PyTorch-ROCm (ATen + MIOpen) run of the same graph on this GPU (SURVEY 8(d) last line; ~25 s)')
    ap.add_argument('--sync-bn', action='store_true', help='train.py --sync-bn: nn.SyncBatchNorm statistics over all ranks (N > 1)')
    ap.add_argument('--ddp', default='reducer', choices=['reducer', 'stock'],
                    help="N > 1 gradient exchange: 'reducer' = parallel.GradReducer (3 flat slices on a side stream), 'stock' = the model "
                         "wrapped in torch.nn.parallel.DistributedDataParallel exactly as train.py:243-245 does")
    ap.add_argument('--dry-dist', action='store_true',
                    help='dev/test only: set the process group up, check the world size against --gpus, print the header line and exit '
                         '(no GPU needed with MYOLO_DIST_BACKEND=gloo; NOT a bench line)')
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: re-execute this command line under torch.distributed.run with
    one rank per GPU (the form the header documents) and hand its exit code back.  Under a launcher (WORLD_SIZE set) this is a no-op."""
    if args.gpus <= 1 or 'WORLD_SIZE' in os.environ:
        return None
    import socket
    import subprocess
    port = os.environ.get('MASTER_PORT')
    if not port:
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def setup_dist(args):
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')         # before the first HIP call: RCCL needs dmabuf IPC on this driver
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to print a line whose '
                         f'n_gpus would not be the number of GPUs asked for')
    if torch.cuda.is_available():
        torch.cuda.set_device(local if local < torch.cuda.device_count() else 0)
    elif not args.dry_dist:
        raise SystemExit('bench.py: no GPU visible (the product path has no CPU fallback)')
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = os.environ.get('MYOLO_DIST_BACKEND', 'nccl')       # 'gloo' only for single-GPU / CPU smoke tests of the N>1 path
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', init_method='env://', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend=backend, init_method='env://')
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f'bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}')
    return world, rank, local


def barrier(world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


class Trainer:
    """the reference's hot loop body (train.py:364-401) restricted to one joint batch."""

    def __init__(self, args, world, rank, dev):
        from multiyolov5_amd.models.yolo import Model
        from multiyolov5_amd import synth
        self.args, self.world, self.dev = args, world, dev
        H, W = args.img
        B = args.batch
        torch.manual_seed(0)
        m = Model(os.path.join(ROOT, 'multiyolov5_amd', 'cfg', args.cfg))
        synth.randomize_(m, seed=0)                     # random-init weights + non-trivial BN stats (no checkpoints offline)
        self.model = m.to(dev).train()
        nc, nl = 10, 3
        hyp = dict(box=0.05 * 3. / nl, cls=0.5 * nc / 80. * 3. / nl, obj=1.0 * (max(H, W) / 640) ** 2 * 3. / nl,
                   cls_pw=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)   # hyp.scratch + train.py:248-250
        m.nc, m.hyp, m.gr = nc, hyp, 1.0
        self.dtype = torch.float16 if args.dtype == 'f16' else torch.float32
        self.imgs = synth.images(B, H, W, seed=1 + rank).to(dev, self.dtype)
        self.targets = synth.det_targets(B, 8, nc, seed=1 + rank).to(dev)
        self.mask = synth.seg_targets(B, H, W, 19, seed=1 + rank).to(dev)
        self.segimgs = None                             # second, independent batch of the train.py-faithful step (lazily)
        self.ni = 0
        if args.stage == 'train':
            from multiyolov5_amd.utils.loss import ComputeLoss, SegmentationLosses
            from multiyolov5_amd.utils.optim import FusedSGD, GradScaler
            from multiyolov5_amd.utils.torch_utils import ModelEMA
            self.compute_loss = ComputeLoss(m)
            self.compute_seg_loss = SegmentationLosses()
            pg0, pg1, pg2 = [], [], []                 # train.py:121-137
            for k, v in m.named_modules():
                if hasattr(v, 'bias') and isinstance(v.bias, torch.nn.Parameter):
                    pg2.append(v.bias)
                if isinstance(v, torch.nn.BatchNorm2d):
                    pg0.append(v.weight)
                elif hasattr(v, 'weight') and isinstance(v.weight, torch.nn.Parameter):
                    pg1.append(v.weight)
            total_bs = B * world
            wd = 0.0005 * total_bs * max(round(64 / total_bs), 1) / 64
            self.opt = FusedSGD([{'params': pg0}, {'params': pg1, 'weight_decay': wd}, {'params': pg2}],
                                lr=0.0015, momentum=0.937, nesterov=True)
            self.scaler = GradScaler(enabled=self.dtype == torch.float16)
            self.ema = ModelEMA(m) if rank == 0 else None
            self.reducer = None
            if world > 1 and args.ddp == 'reducer':
                from multiyolov5_amd.parallel import GradReducer
                self.reducer = GradReducer(m, world)
            if args.sync_bn:                           # train.py:190-193 (after the optimizer's parameter groups, as there)
                self.model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(m)
                self.model.invalidate_plans()
            self.raw_model = self.model
            if world > 1 and args.ddp == 'stock':      # train.py:243-245
                from torch.nn.parallel import DistributedDataParallel as DDP
                if dev.type == 'cuda':
                    self.model = DDP(self.model, device_ids=[dev.index], output_device=dev.index)
                else:
                    self.model = DDP(self.model)
        else:
            self.fixed = None

    def step(self):
        a = self.args
        m = self.model
        if a.stage == 'fwdbwd':
            det, seg = m(self.imgs)
            if self.fixed is None:
                g = torch.Generator(device=self.dev).manual_seed(3)
                self.fixed = [torch.randn(d.shape, device=self.dev, generator=g).to(d.dtype) * 1e-3 for d in det] + \
                    [torch.randn(seg.shape, device=self.dev, generator=g).to(seg.dtype) * 1e-4]
            torch.autograd.backward(list(det) + [seg], self.fixed)
            for p in m.parameters():
                p.grad = None
            return
        B = a.batch
        pred = m(self.imgs)                                              # train.py:364
        loss, items = self.compute_loss(pred[0], self.targets)          # train.py:365
        if self.world > 1:
            loss = loss * self.world                                     # train.py:366-367
        segloss = self.compute_seg_loss(pred[1], self.mask) * B         # train.py:385
        total = loss * 0.6 + segloss * 0.35                              # train.py:290,370,391 (detgain, seggain)
        self.scaler.scale(total).backward()                              # train.py:371/392
        if self.reducer is not None:
            self.reducer.wait()
        self.scaler.step(self.opt)                                       # train.py:397
        self.scaler.update()
        self.opt.zero_grad()
        if self.ema is not None:
            self.ema.update(self.raw_model)                              # train.py:400-401 (model.module under DDP)
        self.last = (loss, segloss)


def step_checks(tr):
    """after a timed region: the last step's losses are finite and the loss scaler never skipped an optimizer step (a skipped step
    halves the scale; the growth interval is 2000 steps, so an untouched scale = no skip).  Raises instead of reporting an invalid rate."""
    if tr.args.stage != 'train':
        return None
    loss, segloss = tr.last
    vals = [float(loss), float(segloss)]
    ok = all(v == v and abs(v) != float('inf') for v in vals)
    scale = tr.scaler.get_scale() if tr.scaler.is_enabled() else None
    found = float(tr.scaler._found) if (tr.scaler.is_enabled() and tr.scaler._found is not None) else 0.0
    if not ok or found != 0.0 or (scale is not None and scale < 65536.0):
        raise RuntimeError(f'invalid bench step: losses {vals}, loss scale {scale}, found_inf {found}')
    # round 6: the launches with a device-wide barrier inside (myolo_conv_bn_act, myolo_bn_act_bwd_fused) bound every spin; a spin that gave
    # up sets a sticky word in the plan's barrier block -- a step measured with it set would be garbage computed quickly
    timeouts = 0
    for h in getattr(tr, 'raw_model', tr.model).__dict__.get('_plans', {}).values():
        bar = h.plan.__dict__.get('_grid_bar')
        if bar is not None:
            timeouts += int(bar[18 * 32])
    if timeouts:
        raise RuntimeError('invalid bench step: a grid barrier inside a fused launch timed out')
    return {'det_loss': vals[0], 'seg_loss_x_batch': vals[1], 'loss_scale': scale, 'optimizer_steps_skipped': 0, 'grid_barrier_timeouts': 0}


def train_py_step(tr):
    """one iteration of the reference's loop body as written (train.py:364-401): detection pass (forward, ComputeLoss, backward)
    on one batch, segmentation pass (forward, CE * batch_size, backward) on a second independent batch, optimizer / EMA every
    `accumulate = max(round(64 / total_batch), 1)` iterations (train.py:140,396-401).  SURVEY 8(d) "training image" definition (ii)."""
    a, m, B = tr.args, tr.model, tr.args.batch
    if tr.segimgs is None:
        from multiyolov5_amd import synth
        tr.segimgs = synth.images(B, a.img[0], a.img[1], seed=101).to(tr.dev, tr.dtype)
        tr.accumulate = max(round(64 / (B * tr.world)), 1)
    pred = m(tr.imgs)
    loss, _ = tr.compute_loss(pred[0], tr.targets)
    if tr.world > 1:
        loss = loss * tr.world
    tr.scaler.scale(loss * 0.6).backward()
    pred = m(tr.segimgs)
    segloss = tr.compute_seg_loss(pred[1], tr.mask) * B * 0.35
    tr.scaler.scale(segloss).backward()
    if tr.reducer is not None:
        tr.reducer.wait()
    tr.ni += 1
    if tr.ni % tr.accumulate == 0:
        tr.scaler.step(tr.opt)
        tr.scaler.update()
        tr.opt.zero_grad()
        if tr.ema is not None:
            tr.ema.update(m)


def train_py_rate(tr, iters=8, warm=4):
    for _ in range(warm):
        train_py_step(tr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        train_py_step(tr)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    B = tr.args.batch * tr.world
    return {'pairs_per_s': B / dt, 'images_per_s': 2 * B / dt, 'ms_per_iteration': dt * 1e3, 'accumulate': tr.accumulate,
            'what': 'train.py:364-401 as written: det pass + seg pass (two forward+backward on two batches) per iteration, optimizer '
                    'and EMA every `accumulate` iterations; a pair = one detection image + one segmentation image'}


def conv_kernel_timing(trainer, nsteps=3):
    """HIP-event timing of every myolo_conv launch (the dominant kernel) over `nsteps` steps, on the launch stream.
    Returns (total algorithmic bytes, total flops, total seconds, launches) per step."""
    from multiyolov5_amd import engine as E
    rec = []

    orig = E.Call.__call__

    def timed(self, st):
        if self.name not in ('myolo_conv', 'myolo_conv_dgrad_s2', 'myolo_conv_dgrad_bn', 'myolo_conv_bn_act'):
            return orig(self, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(self, st)
        e1.record()
        rec.append((self, e0, e1))

    E.Call.__call__ = timed
    graph_mode, E.GRAPH_TRAIN = E.GRAPH_TRAIN, False     # per-launch events need the launch list issued call by call, not a graph replay
    native_mode, E.NATIVE_EXEC = E.NATIVE_EXEC, False    # ... nor through the native executor
    try:
        for _ in range(nsteps):
            trainer.step()
        torch.cuda.synchronize()
    finally:
        E.Call.__call__ = orig
        E.GRAPH_TRAIN = graph_mode
        E.NATIVE_EXEC = native_mode
    tot_t = sum(e0.elapsed_time(e1) for _, e0, e1 in rec) * 1e-3 / nsteps
    # SURVEY 8(d)'s convention: each conv launch's input + weights + output, once (the BatchNorm-backward passes a dgrad
    # launch carries are real traffic of that launch but not part of the convention -- they go into a second field)
    tot_b = sum(E.conv_call_bytes(c) for c, _, _ in rec) / nsteps
    tot_bn = sum(E.bnb_call_bytes(c) + E.apply_fold_bytes(c) for c, _, _ in rec) / nsteps
    tot_f = sum(E.conv_call_flops(c) for c, _, _ in rec) / nsteps
    return tot_b, tot_f, tot_t, len(rec) // nsteps, tot_bn


def infer_report(args, dev, frames=200, warm=16, H=1024, W=2048, cpu=False):
    """detect.py path (detect.py:144-193): fused eval forward at 1x3xHxW fp16 (hipGraph replay) + NMS + seg upsample/argmax.
    Returns the FPS of the whole frame loop (wall clock, the per-frame host sync of NMS included), the per-stage split measured with
    HIP events on the launch stream, whether the forward really was a graph replay, and the frame's HBM roofline (SURVEY 8(d):
    200.8 MB of conv input + weight + output bytes per 512x1024 image, fp16, + the NMS input rows + the label map)."""
    from multiyolov5_amd.models.yolo import Model
    from multiyolov5_amd.utils.general import non_max_suppression, seg_argmax
    from multiyolov5_amd import synth
    if getattr(args, 'eval_fork', None):
        from multiyolov5_amd import runtime as _R
        _R.EVAL_FORK = args.eval_fork
    m = Model(os.path.join(ROOT, 'multiyolov5_amd', 'cfg', 'yolov5s_city_seg.yaml'))
    synth.randomize_(m, seed=0)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):               # (fuse() prints 'Fusing layers...' like the reference: keep stdout to the JSON line)
        m = m.to(dev).half().fuse().eval()
    img = synth.images(1, H, W, seed=7).to(dev, torch.float16)
    na = 3 * ((H // 8) * (W // 8) + (H // 16) * (W // 16) + (H // 32) * (W // 32))          # 129 024 at 1024x2048, 32 256 at 512x1024
    pred_syn = synth.nms_pred(1, na, 10, seed=3, img_w=W, img_h=H).to(dev, torch.float16)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def frame_unchanged():
        """the statements of detect.py:144-149,191-193 as the reference writes them, under the drop-in: `F.interpolate(seg, ...)[0]` +
        `.max(axis=0)[1]` on the model's lazy logits reach the same fused kernel (runtime.LazyResized)"""
        import torch.nn.functional as F
        with torch.no_grad():
            out = m(img)
            det = non_max_suppression(pred_syn, 0.25, 0.45)
            seg = F.interpolate(out[1], (H, W), mode='bilinear', align_corners=True)[0]
            lab = seg.max(axis=0)[1]
        return det, lab

    def frame(timed=False):
        with torch.no_grad():
            if timed:
                ev[0].record()
            out = m(img)
            if timed:
                ev[1].record()
            # random-init heads give no candidates above conf 0.25: NMS is fed the synthetic prediction
            # tensor of the same shape/dtype so that suppression actually happens
            det = non_max_suppression(pred_syn, 0.25, 0.45)
            if timed:
                ev[2].record()
            lab = seg_argmax(out[1], H, W)
            if timed:
                ev[3].record()
        return det, lab
    for _ in range(warm):
        det, _ = frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        frame()
    torch.cuda.synchronize()
    fps = frames / (time.perf_counter() - t0)
    for _ in range(4):
        _, lab_u = frame_unchanged()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        frame_unchanged()
    torch.cuda.synchronize()
    fps_unchanged = frames / (time.perf_counter() - t0)
    same_labels = bool(torch.equal(lab_u, frame()[1][0]))
    # after all those replays the graphs must still compute what the plain launch list computes (round 6: a memset node of the captured graph had
    # stopped clearing the head's accumulators after a few hundred frames, DESIGN section 3 zero_fill_kernel): same frame through both, label maps
    # equal up to the near-tie pixels the head's atomics order can flip
    from multiyolov5_amd import runtime as _R
    lab_g = frame()[1]
    _R.GRAPH_EVAL = False
    try:
        lab_e = frame()[1]
    finally:
        _R.GRAPH_EVAL = True
    torch.cuda.synchronize()
    label_px_vs_launch_list = int((lab_g != lab_e).sum())
    st = [0.0, 0.0, 0.0]
    k = 10
    for _ in range(k):
        frame(timed=True)
        torch.cuda.synchronize()
        for i in range(3):
            st[i] += ev[i].elapsed_time(ev[i + 1]) / k
    holders = [h for h in m.__dict__.get('_plans', {}).values() if not h.plan.training]
    unjoined = any(h.__dict__.get('_graph_c') is not None for h in holders)
    fwd_main = st[0]
    if unjoined:
        # the segmentation head runs on the plan's side stream and is not joined inside the forward (NMS runs beside it): the events
        # above saw only the main chain.  The forward's own duration = the same call followed by the join its consumers perform
        st[0] = 0.0
        for _ in range(k):
            with torch.no_grad():
                ev[0].record()
                m(img)
                for h in holders:
                    h.wait_branch()
                ev[1].record()
            torch.cuda.synchronize()
            st[0] += ev[0].elapsed_time(ev[1]) / k
    replayed = bool(holders) and all(h.__dict__.get('_graph') is not None and not h.__dict__.get('_graph_failed') for h in holders)
    from multiyolov5_amd import engine as E
    conv_b = sum(E.conv_call_bytes(c) for h in holders[:1] for op in h.plan.ops for c in op.fwd_calls if c.name in ('myolo_conv', 'myolo_conv_pair'))
    nlaunch = sum(len(op.fwd_calls) for h in holders[:1] for op in h.plan.ops)
    survey_b = 200.8e6 * (H * W) / (512 * 1024)
    alg = survey_b + na * 15 * 2 + H * W * 8
    r = {'value': fps, 'unit': 'frames/s',
         'workload': f'pspv5s fused fp16 1x3x{H}x{W} fwd + NMS({na} rows, {int(det[0].shape[0])} kept) + x8 upsample+argmax (int64 labels)',
         'ms_per_frame': 1e3 / fps,
         'unchanged_caller': {'value': fps_unchanged, 'unit': 'frames/s', 'same_labels_as_seg_argmax': same_labels,
                              'what': "the same frame loop with detect.py:191-193's own statements -- F.interpolate(seg, (h0, w0), "
                                      "mode='bilinear', align_corners=True)[0]; seg.max(axis=0)[1] -- instead of utils.general.seg_argmax: "
                                      'the lazy logits turn them into the fused resize + arg-max launch (runtime.LazyResized)'},
         'stage_ms': {'forward': st[0], 'nms': st[1], 'argmax': st[2], 'forward_main_chain': fwd_main, 'head_unjoined': unjoined,
                      'what': 'HIP events on the launch stream, 10 frames; nms includes its device->host read of the keep counts; with '
                              'head_unjoined the segmentation head (side stream) overlaps NMS, so the stages add up to more than the frame: '
                              'forward = whole forward incl. the join, forward_main_chain = what the main stream waits for before NMS'},
         'graph_replayed': replayed, 'forward_launches': nlaunch,
         'label_pixels_differing_from_the_eager_launch_list_after_the_timed_loops': label_px_vs_launch_list,
         'roofline': {'bound': 'hbm', 'achieved': alg * fps / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                      'frac': alg * fps / 1e9 / HBM_PEAK_GBS, 'algorithmic_bytes_per_frame': alg,
                      'what': 'SURVEY 8(d): conv input + weights + output once each, fp16 (200.8 MB per 512x1024 image) + NMS input rows + '
                              'int64 label map, over the whole frame time',
                      'conv_bytes_of_this_plan': conv_b,
                      'forward_only_frac': survey_b / (st[0] * 1e-3) / 1e9 / HBM_PEAK_GBS if st[0] > 0 else None}}
    if cpu:
        from oracle import cpu_bench
        r['cpu_baseline'] = cpu_bench.detect_frame(H=H, W=W, budget_s=8.0)
    return r


def cpu_baseline(args):
    """the oracle (CPU restatement of the reference path, torch CPU fp32; kind "port": the reference itself is not on the GPU box)
    timed on this box's host cores on a bounded sample: BASELINE configs[0] -- yolov5s_city_seg with the BASE head, joint step (fwd +
    both losses + bwd) at 2x3x512x1024 -- plus the same sample on the benchmarked config's own head."""
    from oracle import cpu_bench
    r = cpu_bench.joint_step(cfg='yolov5s_city_seg_base.yaml', batch=2, H=args.img[0], W=args.img[1], budget_s=12.0)
    r['workload'] = 'BASELINE configs[0]: yolov5s_city_seg base head, 2x3x512x1024 fp32 joint step (oracle port of the reference CPU path)'
    if args.cfg != 'yolov5s_city_seg_base.yaml':
        o = cpu_bench.joint_step(cfg=args.cfg, batch=2, H=args.img[0], W=args.img[1], budget_s=10.0)
        r['same_head_as_bench'] = {'value': o['value'], 'unit': o['unit'], 'sample': o['sample']}
    return r


SURVEY_FWD_BYTES_PER_IMAGE = {          # conv input + weights + output once each, fp16, one 512x1024 image, forward
    'yolov5s_city_seg.yaml': 200.8e6, 'yolov5s_city_seg_base.yaml': 233.6e6, 'yolov5m_city_seg_lab.yaml': 333.6e6}


def stock_rocm_baseline(args):
    """SURVEY 8(d) 'unmodified reference on MI355X': the reference graph as the ATen ops its modules call (oracle.rocm_stock_bench: ATen +
    MIOpen under torch.autocast(fp16), torch.optim.SGD, GradScaler, foreach EMA), on THIS GPU, in its own process, after every timed
    region of this file.  A baseline beside `cpu_baseline`, never the product path."""
    import subprocess
    env = dict(os.environ, MIOPEN_FIND_MODE='FAST', MIOPEN_USER_DB_PATH='/tmp/miopen', MIOPEN_LOG_LEVEL='2')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    out = {'what': 'oracle.rocm_stock_bench: PyTorch %s ATen + MIOpen (immediate mode, no gfx950 find-db), NCHW, autocast fp16, same synthetic '
                   'batch; train = fwd + both losses + bwd + SGD + EMA, infer = fused fp16 forward + seg argmax WITHOUT NMS' % torch.__version__}
    try:
        r = subprocess.run([sys.executable, '-m', 'oracle.rocm_stock_bench', '--cfg', args.cfg, '--batch', str(args.batch), '--img',
                            str(args.img[0]), str(args.img[1]), '--steps', '6', '--warmup', '2'], cwd=ROOT, env=env, capture_output=True,
                           text=True, timeout=240)
        for ln in r.stdout.splitlines():
            if ln.startswith('{'):
                j = json.loads(ln)
                out[j.pop('leg')] = j
        if r.returncode != 0 and len(out) == 1:
            out['error'] = r.stderr[-400:]
    except Exception as e:                                # a baseline must not take the primary line down
        out['error'] = repr(e)
    return out


def whole_step_roofline(tr, ms):
    """SURVEY 8(d) convention: a training step moves 3x the forward's conv bytes (forward + dgrad + wgrad; every conv operand once,
    no fusion credit, nothing else counted) -- 3 x 200.8 MB x 16 = 9.64 GB for the benchmarked step -- over the WHOLE step time."""
    a = tr.args
    per = SURVEY_FWD_BYTES_PER_IMAGE.get(a.cfg)
    if per is None or a.dtype != 'f16':
        return None
    tot = 3.0 * per * a.batch * (a.img[0] * a.img[1]) / (512 * 1024)
    return {'bound': 'hbm', 'achieved': tot / (ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': tot / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'algorithmic_bytes_per_step': tot,
            'what': 'SURVEY 8(d): 3 x (conv input + weights + output of the forward, fp16) x batch, over the whole step time'}


def whole_step_launch_bytes(tr, ms):
    """a DIFFERENT count, kept for continuity with rounds 1-2: the operand bytes of the launches this implementation actually issues
    (conv + dgrad + weight-gradient + BatchNorm passes, every operand once) over the whole step time.  It counts the BatchNorm passes
    as work, which SURVEY 8(d) does not -- it is NOT the roofline fraction."""
    from multiyolov5_amd import engine as E
    plans = [h.plan for h in tr.model.__dict__.get('_plans', {}).values() if h.plan.training]
    if not plans:
        return None
    by = E.plan_algorithmic_bytes(plans[0])
    tot = sum(by.values())
    return {'GBps': tot / (ms * 1e-3) / 1e9, 'bytes_per_step': tot, 'by_family': by,
            'what': 'operand bytes of the issued conv + dgrad + weight-gradient + BatchNorm launches (each once) / whole step time'}


def second_config_rate(args, world, rank, dev, cfg='yolov5m_city_seg_lab.yaml', batch=8, steps=10, warm=4):
    """BASELINE configs[3] per-GPU share: yolov5m + Lab head, bs 8 per GPU, the same joint step"""
    import copy
    a = copy.copy(args)
    a.cfg, a.batch = cfg, batch
    tr = Trainer(a, world, rank, dev)
    for _ in range(warm):
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    r = {'value': batch / dt, 'unit': 'images/s', 'ms_per_step': dt * 1e3,
         'workload': f'{cfg} bs={batch}/GPU {args.img[1]}x{args.img[0]} {args.dtype} joint train step (BASELINE configs[3] per-GPU share)'}
    r['whole_step_roofline'] = whole_step_roofline(tr, dt * 1e3)
    r['launch_operand_bytes'] = whole_step_launch_bytes(tr, dt * 1e3)
    r['checks'] = step_checks(tr)
    return r


def augment_rates(dev, n=24):
    """device augmentation throughput at the training scripts' real sizes, next to the reference's own CPU path
    where its library is installed (Pillow: SegmentationDataset.py:118-151 + ColorJitter on one core; cv2 is absent, so the
    detection side has no CPU number here).  Host-side parameter draws and table construction are inside the timed region."""
    import random
    import numpy as np
    from multiyolov5_amd.utils import augment as A
    rs = np.random.RandomState(0)
    out = {}
    # segmentation: a 2048x1024 Cityscapes frame -> random scale [0.65, 3] x 1024 -> 1024x512 crop -> ColorJitter -> fp16 CHW
    img = torch.from_numpy(rs.randint(0, 256, (1024, 2048, 3)).astype(np.uint8)).to(dev)
    mask = torch.from_numpy(rs.randint(0, 34, (1024, 2048)).astype(np.uint8)).to(dev)
    rng = random.Random(0)
    g = torch.Generator().manual_seed(0)

    def seg_one():
        p = A.draw_sync_params(2048, 1024, 1024, (1024, 512), rng=rng)
        a, lab = A.seg_sync_transform(img, mask, p)
        return A.color_jitter(a, A.draw_color_jitter_params(generator=g), dtype=torch.float16), lab
    for _ in range(3):
        seg_one()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n):
        seg_one()
    torch.cuda.synchronize(dev)
    out['seg_samples_per_s'] = n / (time.perf_counter() - t0)
    # detection: 4-image mosaic at img_size 1024 from 1024x512 frames, hyp.scratch.yaml
    hyp = dict(degrees=0.0, translate=0.1, scale=0.5, shear=0.0, perspective=0.0, flipud=0.0, fliplr=0.5, mosaic=1.0, mixup=0.0,
               hsv_h=0.015, hsv_s=0.7, hsv_v=0.4)
    frames = [torch.from_numpy(rs.randint(0, 256, (512, 1024, 3)).astype(np.uint8)).to(dev) for _ in range(8)]
    labels = [np.concatenate([rs.randint(0, 10, (8, 1)), rs.uniform(0.2, 0.8, (8, 2)), rs.uniform(0.05, 0.3, (8, 2))], 1).astype(np.float32)
              for _ in range(8)]
    nprng = np.random.RandomState(0)

    def det_one(i):
        return A.mosaic_train_sample(i % 8, lambda k: frames[k], lambda k: labels[k], range(8), 1024, hyp, rng, nprng)
    for i in range(3):
        det_one(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(n):
        det_one(i)
    torch.cuda.synchronize(dev)
    out['det_mosaic_samples_per_s'] = n / (time.perf_counter() - t0)
    out['what'] = ('seg: _sync_transform (2048x1024 -> random scale -> 1024x512 crop) + ColorJitter + ToTensor fp16; det: 4-image mosaic at '
                   'img_size 1024 + warpAffine + augment_hsv + flip; one sample at a time, host parameter draws included')
    try:                                                  # the reference's CPU path for the segmentation sample, through the real Pillow
        from PIL import Image, ImageEnhance, ImageOps
        im0, m0 = Image.fromarray(img.cpu().numpy()), Image.fromarray(mask.cpu().numpy())
        rng2 = random.Random(0)
        t0 = time.perf_counter()
        k = 4
        for _ in range(k):
            p = A.draw_sync_params(2048, 1024, 1024, (1024, 512), rng=rng2)
            a, b = (im0.transpose(Image.FLIP_LEFT_RIGHT), m0.transpose(Image.FLIP_LEFT_RIGHT)) if p['flip'] else (im0, m0)
            a, b = a.resize((p['ow'], p['oh']), Image.BILINEAR), b.resize((p['ow'], p['oh']), Image.NEAREST)
            if p['ow'] < 1024 or p['oh'] < 512:
                a = ImageOps.expand(a, border=(0, 0, max(1024 - p['ow'], 0), max(512 - p['oh'], 0)), fill=0)
                b = ImageOps.expand(b, border=(0, 0, max(1024 - p['ow'], 0), max(512 - p['oh'], 0)), fill=255)
            a = a.crop((p['x1'], p['y1'], p['x1'] + 1024, p['y1'] + 512))
            b = b.crop((p['x1'], p['y1'], p['x1'] + 1024, p['y1'] + 512))
            a = ImageEnhance.Color(ImageEnhance.Contrast(ImageEnhance.Brightness(a).enhance(1.2)).enhance(0.9)).enhance(1.1)
            hh, ss, vv = a.convert('HSV').split()
            a = Image.merge('HSV', (hh.point(lambda v: (v + 12) % 256), ss, vv)).convert('RGB')
            torch.from_numpy(np.array(a)).permute(2, 0, 1).float().div(255)
            torch.from_numpy(np.array(b)).long()
        out['seg_cpu_pillow_samples_per_s'] = k / (time.perf_counter() - t0)
        out['seg_cpu_cores'] = 1
    except Exception as e:
        out['seg_cpu_pillow_samples_per_s'] = None
        out['seg_cpu_error'] = repr(e)
    return out


def dry_exchange(world, rank):
    """`--dry-dist` with parallel.GradReducer (no GPU, gloo): a training plan of the benchmarked model is dry-built on the CPU, the backward's
    segment list is walked exactly like engine.Plan._bwd_eager walks it (no kernels: every rank fills the flat gradient buffer with its own
    random numbers), the reducer gets each slice the moment the walk passes the op that completes it, and the result must be the mean over
    the ranks.  Returns what a real N-rank line carries: the reducer's description incl. the slice issue order, the buckets' completion order,
    and `allreduce_exposed_ms` (here: host wall time of finish(), gloo on CPU tensors -- plumbing, not a measurement)."""
    from multiyolov5_amd import runtime as R
    from multiyolov5_amd.models.yolo import Model
    from multiyolov5_amd.parallel import GradReducer
    torch.manual_seed(0)
    m = Model(os.path.join(ROOT, 'multiyolov5_amd', 'cfg', 'yolov5s_city_seg.yaml')).train()
    red = GradReducer(m, world, nbuckets=3)
    plan = R.PlanHolder(m, [torch.zeros(2, 3, 64, 128)], ('t', 0), torch.float32, True).plan
    buckets = plan.grad_buckets(red)                                     # [(lo, hi, ready_after_op)] in completion order
    total = plan.flat_grad.numel()
    plan.flat_grad.copy_(torch.randn(total, generator=torch.Generator().manual_seed(1000 + rank)))
    walked = []
    for hi, lo, ready in plan._bwd_segments(red):                        # descending op index = the order the backward launches run in
        walked.append((hi, lo))
        for a, b in ready:
            red.reduce_slice(plan.flat_grad, a, b)
    t0 = time.perf_counter()
    red.finish(plan.flat_grad)
    exposed = (time.perf_counter() - t0) * 1e3
    ref = sum(torch.randn(total, generator=torch.Generator().manual_seed(1000 + r)) for r in range(world)) / world
    err = float((plan.flat_grad - ref).abs().max())
    worst = torch.tensor([err])
    dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    return {'grad_exchange_detail': red.describe(), 'bucket_completion_order': [[int(a), int(b), int(r)] for a, b, r in buckets],
            'backward_segments': [[int(h), int(l)] for h, l in walked], 'flat_grad_elements': int(total),
            'exchange_max_abs_err_over_ranks': float(worst.item()), 'allreduce_exposed_ms': exposed}


def main():
    args = parse()
    rc = spawn_ranks(args)
    if rc is not None:
        sys.exit(rc)
    world, rank, local = setup_dist(args)
    if args.dry_dist:
        n = world
        if world > 1:
            t = torch.ones(1)
            dist.all_reduce(t)                           # every rank is really there
            n = int(t.item())
            dist.barrier()
        rec = {'metric': 'DEV ONLY launch check -- not a bench line', 'n_gpus': n, 'value': None,
               'config': {'parallelism': f'dp{world}', 'ddp': args.ddp}}
        if world > 1 and args.ddp == 'reducer':
            rec.update(dry_exchange(world, rank))        # the gradient exchange itself over a dry-built plan
        if rank == 0:
            print(json.dumps(rec))
        if world > 1:
            dist.destroy_process_group()
        return
    dev = torch.device('cuda', local if local < torch.cuda.device_count() else 0)
    from multiyolov5_amd import _lib
    _lib.lib()                                           # fail loudly if the HIP library is missing
    out = {}
    if args.stage == 'infer':
        H, W = args.infer_size
        r = infer_report(args, dev, H=H, W=W, frames=args.steps, cpu=not args.no_cpu_baseline)
        if rank == 0:
            r['metric'] = f'detect.py FPS (fwd+NMS+argmax) {W}x{H}'
            print(json.dumps(r))
        return
    tr = Trainer(args, world, rank, dev)
    for _ in range(args.warmup):
        tr.step()
    barrier(world)
    if getattr(tr, 'reducer', None) is not None:
        tr.reducer.exposed = []                          # (two event records per step; the wait itself is unchanged)
    from multiyolov5_amd.engine import SyncPoint
    SyncPoint.calls, SyncPoint.host_s = 0, 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.step()
    barrier(world)
    dt = time.perf_counter() - t0
    exposed_ms = None
    if getattr(tr, 'reducer', None) is not None and tr.reducer.exposed:
        exposed_ms = sum(a.elapsed_time(b) for a, b in tr.reducer.exposed) / len(tr.reducer.exposed)
        tr.reducer.exposed = None
    checks = step_checks(tr)                             # finite losses, no skipped optimizer step in the timed region
    per_rank_ms = None
    if world > 1:
        # every rank's own wall time of the timed region (diagnosis of a scaling run from ONE line: a straggler rank, or all ranks slow),
        # then the MAX over ranks is the job's time
        mine = torch.tensor([dt], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [float(e.item()) / args.steps * 1e3 for e in every]
        t = mine.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    ips = args.batch * world * args.steps / dt
    H, W = args.img
    out = {
        'metric': 'train images/sec (joint det+seg step: fwd + ComputeLoss + seg CE + bwd + SGD + EMA)',
        'value': ips, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': (f'{args.cfg} bs={args.batch}/GPU {W}x{H} {args.dtype} joint train step'
                                + (' (psp head; BASELINE configs[1])' if args.cfg == 'yolov5s_city_seg.yaml' and args.batch == 16
                                   and args.dtype == 'f16' else '')),
                   'global_batch': args.batch * world, 'parallelism': f'dp{world}' + ('+syncbn' if args.sync_bn else ''), 'stage': args.stage,
                   'grad_exchange': None if world == 1 else ('torch DistributedDataParallel (train.py:243-245)' if args.ddp == 'stock'
                                                              else 'parallel.GradReducer: 3 flat slices, RCCL on a side stream')},
    }
    assert out['n_gpus'] == args.gpus, (out['n_gpus'], args.gpus)
    out['checks'] = checks
    if args.sync_bn:
        # nn.SyncBatchNorm: host-issued all-reduces of the per-layer statistics (2 per Conv+BatchNorm layer and step), each cutting the
        # native launch program (engine.SyncPoint); their enqueue cost on this rank
        out['syncbn'] = {'collectives_per_step': SyncPoint.calls / max(args.steps, 1), 'host_ms_per_step': SyncPoint.host_s / max(args.steps, 1) * 1e3}
    if world > 1:
        # the main stream's wait for the RCCL slices at the end of the backward (HIP events, mean over the timed steps, this rank):
        # what the overlap with the backward did NOT hide
        out['allreduce_exposed_ms'] = exposed_ms
        out['per_rank_ms_per_step'] = per_rank_ms
        try:
            out['collective_library'] = {'backend': dist.get_backend(), 'rccl_version': '.'.join(str(v) for v in torch.cuda.nccl.version())}
        except Exception as e:                           # (gloo smoke runs: no RCCL in the process)
            out['collective_library'] = {'backend': dist.get_backend(), 'rccl_version': None, 'note': repr(e)}
        if getattr(tr, 'reducer', None) is not None:
            out['grad_exchange_detail'] = tr.reducer.describe()
    if args.stage != 'train':
        out['metric'] = 'DEV ONLY fwd+bwd images/sec (no loss / optimizer) -- not a bench line'
    if rank == 0 and world == 1:
        if not args.no_kernel_timing:
            b, f, t, n, b_bn = conv_kernel_timing(tr)
            traffic, tsrc = None, None
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc.json')))
            pmc = cands[-1] if cands else ''        # newest separate rocprofv3 --pmc passes (scripts/gpu_pmc.sh + scripts/pmc_summary.py)
            if os.path.exists(pmc) and args.batch == 16 and tuple(args.img) == (512, 1024) and args.dtype == 'f16':
                rec = json.load(open(pmc))
                traffic, tsrc = rec['conv_hbm_bytes_per_launch'], 'profiles/' + os.path.basename(pmc) + ': ' + rec['source']
            out['roofline'] = {'bound': 'hbm', 'achieved': b / t / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                               'frac': b / t / 1e9 / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': tsrc,
                               'kernel': 'myolo_conv launches of one step: conv_mid_kernel + conv_midx_kernel + conv_halo_kernel + conv_stream_kernel + conv_igemm_kernel (forward convs + dgrads); achieved = SURVEY 8(d) bytes (input + weights + output of each launch, once) / their summed durations',
                               'with_batchnorm_riders': {'achieved': (b + b_bn) / t / 1e9, 'frac': (b + b_bn) / t / 1e9 / HBM_PEAK_GBS,
                                                         'what': 'the same launches with the BatchNorm-backward reduce / apply passes some dgrads carry counted as work (rounds 4-5 reported this as `achieved`)'},
                               'launches_per_step': n,
                               'avg_launch_us': t / n * 1e6, 'algorithmic_bytes_per_launch': b / n,
                               'mfma_tflops': f / t / 1e12, 'mfma_frac': f / t / 1e12 / MFMA_F16_PEAK_TF,
                               'conv_time_frac_of_step': t / (ms * 1e-3)}
        if args.stage == 'train':
            out['whole_step_roofline'] = whole_step_roofline(tr, ms)
            out['launch_operand_bytes'] = whole_step_launch_bytes(tr, ms)
        if args.stage == 'train' and not args.no_infer:
            try:
                cpu = not args.no_cpu_baseline
                out['detect_fps'] = infer_report(args, dev, cpu=cpu)                       # BASELINE configs[4]: 2048x1024
                out['detect_fps_1024x512'] = infer_report(args, dev, H=512, W=1024, cpu=cpu)
                out['detect_fps_1024x512']['workload'] += " (the resolution of BASELINE.md's ~141 FPS point, unstated NVIDIA GPU)"
            except Exception as e:                       # the secondary metric must not take the primary line down
                out['detect_fps'] = {'value': None, 'error': repr(e)}
            try:
                if args.cfg == 'yolov5s_city_seg.yaml':
                    out['train_m_lab'] = second_config_rate(args, world, rank, dev)
            except Exception as e:
                out['train_m_lab'] = {'value': None, 'error': repr(e)}
        if args.stage == 'train' and not args.no_infer:
            try:                                         # SURVEY 8(d) definition (ii); secondary, must not take the primary line down
                out['train_py_step'] = train_py_rate(tr)
            except Exception as e:
                out['train_py_step'] = {'pairs_per_s': None, 'error': repr(e)}
        if args.stage == 'train' and not args.no_infer:
            try:
                out['augment'] = augment_rates(dev)
            except Exception as e:
                out['augment'] = {'error': repr(e)}
        if args.stage == 'train' and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args)
        if args.stage == 'train' and not args.no_stock_baseline and not args.no_infer:
            del tr                                       # (the stock run needs a few GB of the same GPU)
            torch.cuda.empty_cache()
            out['stock_rocm_baseline'] = stock_rocm_baseline(args)
            sb = out['stock_rocm_baseline'].get('train', {}).get('images_per_s')
            if sb:
                out['stock_rocm_baseline']['speedup_train'] = out['value'] / sb
    if rank == 0:
        # the secondary figures as short scalars right behind the headline fields (the nested objects at the end of the line did
        # not survive the driver's stdout truncation): FPS of the detect.py path at both sizes, the yolov5m + Lab share of config 4, the
        # reference loop body as written, the CPU oracle port
        def num(path):
            v = out
            for k in path:
                v = v.get(k) if isinstance(v, dict) else None
            return round(v, 3) if isinstance(v, (int, float)) else None
        short = {'fps_2048x1024': num(('detect_fps', 'value')), 'fps_1024x512': num(('detect_fps_1024x512', 'value')),
                 'm_lab_bs8_img_s': num(('train_m_lab', 'value')), 'train_py_pairs_s': num(('train_py_step', 'pairs_per_s')),
                 'conv_roofline_frac': num(('roofline', 'frac')), 'conv_roofline_frac_r5_convention': num(('roofline', 'with_batchnorm_riders', 'frac')),
                 'step_roofline_frac': num(('whole_step_roofline', 'frac')),
                 'cpu_port_img_s': num(('cpu_baseline', 'value')), 'stock_rocm_img_s': num(('stock_rocm_baseline', 'train', 'images_per_s'))}
        head = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step')
        line = {k: out[k] for k in head if k in out}
        line['secondary'] = {k: v for k, v in short.items() if v is not None}
        line.update({k: v for k, v in out.items() if k not in head})
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
