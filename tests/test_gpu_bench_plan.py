"""-m gpu: the BENCHMARKED plans, launch by launch (VERDICT r4 "the benchmarked configuration in the benchmarked dtype at the benchmarked
batch is never compared with the oracle"; the planner's tile-count thresholds can pick other kernel variants at batch 16 than the
batch-2 C-ABI / layer tests force).

* `bench.Trainer` itself builds BASELINE configs[1] (yolov5s + PSP, 16x3x512x1024, fp16, loss scale 65536) and runs one joint training
  step with the launch lists issued call by call; `tests.desc_ref.LaunchChecker` evaluates every `myolo_conv` / `myolo_conv_dgrad_s2` /
  `myolo_conv_dgrad_bn` / `myolo_conv_wgrad` descriptor -- and every `myolo_bn_act_*` launch -- of that step in torch fp32 on the CPU over the operands the launch actually read
  and compares everything the launch stored -- outputs (accumulated ones against previous + result), BatchNorm statistics, folded
  BatchNorm-backward sums, the dy / dgamma / dbeta of the apply fold, weight and bias gradients.  Tolerances: 3e-3 relative L2 on
  fp16-rounded outputs (half-ulp rounding alone is ~3e-4), 2e-3 on fp32 sums and weight gradients.
* the same for config 5's frame (fused eval plan, 1x3x1024x2048 fp16) and for the per-GPU share of config 4 (yolov5m + Lab, batch 8).
* the library's launch trace (myolo_trace_start / _read) lists the kernel variants (family + template arguments) those plans ran; the
  list goes to gpurun_out/bench_plan_variants.txt and every convolution family the docs name must appear in it."""
import os
from types import SimpleNamespace

import pytest
import torch

from tests.gpu_util import check

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(**kw):
    a = dict(gpus=1, steps=1, warmup=0, batch=16, img=(512, 1024), cfg='yolov5s_city_seg.yaml', dtype='f16', stage='train',
             infer_size=(1024, 2048), no_cpu_baseline=True, no_infer=True, no_kernel_timing=True, no_stock_baseline=True,
             sync_bn=False, ddp='reducer', dry_dist=False)
    a.update(kw)
    return SimpleNamespace(**a)


def _dump_trace(tag):
    from multiyolov5_amd import _lib as L
    tr = L.launch_trace()
    L.lib().myolo_trace_start(0)
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'bench_plan_variants.txt'), 'a') as f:
            f.write(f'== {tag}\n')
            for site, n in sorted(tr.items(), key=lambda kv: -kv[1]):
                f.write(f'{n:5d}  {site}\n')
    except OSError:
        pass
    return tr


def _train_step_checked(args, tag, expect, bn=False, every=1, only=None):
    import bench
    from multiyolov5_amd import _lib as L
    from tests.desc_ref import LaunchChecker
    tr = bench.Trainer(args, 1, 0, torch.device(DEV))
    tr.step()                                   # builds the plan (native executor, as the bench runs it), moves the BatchNorm statistics once
    torch.cuda.synchronize()
    L.lib().myolo_trace_start(1)
    with LaunchChecker(check, tag, bn=bn, every=every, only=only) as lc:
        tr.step()
        torch.cuda.synchronize()
    sites = _dump_trace(tag)
    assert not lc.bad, f'{len(lc.bad)} of {lc.k} launches differ from torch fp32:\n' + '\n'.join(lc.bad[:40])
    bench.step_checks(tr)                       # finite losses, no skipped optimizer step
    for name, lo in expect.items():
        assert lc.n.get(name, 0) >= lo, (name, lc.n)
    return lc, sites


def test_every_conv_launch_of_the_benchmarked_training_step_matches_torch_fp32():
    """BASELINE configs[1]: 79 convolutions -> 70 forward launches (9 merged pairs), their dgrads (stride-2 ones as one fused launch, 1x1
    Conv+BatchNorm+SiLU ones with the apply fold) and 79 weight gradients: 78 myolo_conv_wgrad launches + the stem layer's one-pass BatchNorm-backward + weight-gradient launch"""
    # ... and EVERY BatchNorm launch of that step, unsampled (rounds 5's suite evaluated every fourth): forward with its
    # saved / running statistics, backward reduce, backward apply with dgamma / dbeta and the shortcut's gradient pass-through, plain and
    # split, and round 6's one-launch form (reduce + barrier + apply): the other 7.6 GB of the step's traffic
    lc, sites = _train_step_checked(_args(), 'bench16', {'myolo_conv': 90, 'myolo_conv_wgrad': 78, 'myolo_bn_wgrad_stem': 1, 'myolo_conv_dgrad_s2': 6, 'myolo_conv_dgrad_bn': 15,
                                                        'myolo_bn_act_fwd': 40, 'myolo_bn_act_bwd_reduce': 20, 'myolo_bn_act_bwd_apply': 18,
                                                        'myolo_bn_act_bwd_fused': 4}, bn=1)
    fam = ' '.join(sites)
    for k in ('mid::launch', 'midx::launch', 'halo::launch', 'stream::launch', 'launch_conv4', 'wgt::launch'):     # every conv family DESIGN section 3 names
        assert k in fam, (k, sorted(sites))


def test_the_opt_in_fused_conv_bn_silu_launches_of_the_benchmarked_step_match_torch_fp32(monkeypatch):
    """engine.CONV_BN_ACT (MYOLO_CONV_BN_ACT=1, north_star's "fused Conv+BN+SiLU" in training mode): the 27 one-launch layers of BASELINE
    configs[1] at their real tile counts -- raw output, statistics, saved / running statistics and the activation of every one of them against
    torch fp32 over the operands the launch read, and the barrier's timeout word; the step's losses stay finite"""
    from multiyolov5_amd import engine as E
    monkeypatch.setattr(E, 'CONV_BN_ACT', True)
    _train_step_checked(_args(), 'bench16_fused_fwd', {'myolo_conv_bn_act': 27}, only=('myolo_conv_bn_act',))


def test_every_conv_launch_of_the_yolov5m_lab_share_matches_torch_fp32():
    """BASELINE configs[3]'s per-GPU share: yolov5m + Lab head (ASPP encoder, FFM), batch 8, fp16: the 48 / 96 / 192 / 384 / 768-channel
    layers and the dilated 3x3 of ASPP at their real tile counts"""
    _train_step_checked(_args(cfg='yolov5m_city_seg_lab.yaml', batch=8), 'mlab8', {'myolo_conv': 110, 'myolo_conv_wgrad': 75})       # (every launch: VERDICT r5 item 7)


@pytest.mark.parametrize('size', [(1024, 2048), (512, 1024)], ids=['2048x1024', '1024x512'])
def test_every_conv_launch_of_the_detect_frame_matches_torch_fp32(size):
    """BASELINE configs[4]: pspv5s fused, .half(), one frame (detect.py:144-149): folded-BatchNorm epilogue (scale, shift, SiLU, residual),
    Detect's permuted store, the split-K small-map kernel -- first (eager) forward of the plan that later replays as hipGraphs"""
    from multiyolov5_amd import _lib as L, synth
    from multiyolov5_amd.models.yolo import Model
    from tests.desc_ref import LaunchChecker
    H, W = size
    m = Model(os.path.join(ROOT, 'multiyolov5_amd', 'cfg', 'yolov5s_city_seg.yaml'))
    synth.randomize_(m, seed=0)
    m = m.to(DEV).half().fuse().eval()
    img = synth.images(1, H, W, seed=7).to(DEV, torch.float16)
    L.lib().myolo_trace_start(1)
    with LaunchChecker(check, f'frame{W}x{H}') as lc, torch.no_grad():
        out = m(img)
        torch.cuda.synchronize()
    sites = _dump_trace(f'frame{W}x{H}')
    assert not lc.bad, f'{len(lc.bad)} of {lc.k} launches differ from torch fp32:\n' + '\n'.join(lc.bad[:40])
    assert lc.n.get('myolo_conv', 0) + 2 * lc.n.get('myolo_conv_pair', 0) >= 60 and torch.isfinite(out[0][0].float()).all()
    assert any('small::launch' in s for s in sites) and any('mid::launch' in s for s in sites)
    # the graph replays that follow run the same descriptors: same outputs
    ref = [out[0][0].clone(), out[1].float().clone()]
    for _ in range(4):
        with torch.no_grad():
            o = m(img)
    torch.cuda.synchronize()
    assert torch.equal(o[0][0], ref[0]) and torch.equal(o[1].float(), ref[1])
