"""CPU (dry plan): why each BatchNorm-backward reduce pass of the benchmarked training plan is, or is not, folded into the dgrad that
completes its output gradient (engine.Plan._plan_bn_stats).  python scripts/bn_reduce_table.py > profiles/r5_bn_reduce_layers.md"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiyolov5_amd import engine as E, runtime as R  # noqa: E402
from multiyolov5_amd.models.yolo import Model  # noqa: E402

B, H, W = 16, 512, 1024
m = Model(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multiyolov5_amd', 'cfg', 'yolov5s_city_seg.yaml')).train()
names = {id(mod): n for n, mod in m.named_modules()}
plan = R.PlanHolder(m, [torch.zeros(B, 3, H, W)], ('t', 0), torch.float16, True).plan
from multiyolov5_amd import engine as _E
max_elems = _E.BN_STATS_MAX_ELEMS
rows, tally = [], {}
for op in plan.ops:
    if not isinstance(op, E.ConvOp) or op.bn is None:
        continue
    o = op.out
    elems = o.n * o.h * o.w * o.c
    name = names.get(id(op.bn), '?')
    if op.group:
        why = 'tiny: one-workgroup launch does the whole BatchNorm backward (myolo_tiny_conv_bwd)'
    elif op.reduce_by is not None:
        w = op.reduce_by
        why = f'FOLDED into the dgrad of {names.get(id(w.bn), "?") if w.bn is not None else "conv"} ({w.k}x{w.k} s{w.s})'
    elif op.bn2 is not None:
        why = 'merged cv1|cv2 pair: one split reduce launch for both parameter sets (two gout slices with different last writers)'
    elif elems > max_elems:
        why = f'map of {elems / 2**20:.0f} M elements > {max_elems >> 20} M: the standalone pass streams at 3-4 TB/s, the epilogue form re-reads y with 8-byte loads (measured slower, round 2)'
    else:
        lo, hi = o.coff, o.coff + o.c
        ws = [(a, z, w) for a, z, w in o.buf.gwriters if a < hi and z > lo]
        if not ws:
            why = 'no gradient writer (output only)'
        else:
            a, z, w = ws[-1]
            if not isinstance(w, E.ConvOp):
                why = f'last writer of gout is {type(w).__name__} (not a convolution)'
            elif a > lo or z < hi:
                why = 'last conv writer covers only part of the channel range'
            elif (w.x.n, w.x.h, w.x.w) != (o.n, o.h, o.w) or w.x.buf is not o.buf:
                why = 'last writer reads another view'
            else:
                why = 'other (stride / segment limit)'
    key = why.split(':')[0].split('(')[0].strip()[:40]
    tally[key] = tally.get(key, 0) + 1
    rows.append((name, f'{o.h}x{o.w}x{o.c}', f'{elems / 2**20:.1f}', why))
print(f'# BatchNorm-backward reduce passes of yolov5s+PSP, {B}x3x{H}x{W} fp16 (dry plan, `scripts/bn_reduce_table.py`)\n')
print('| BatchNorm layer | map | M elements | reduce pass |\n|---|---|---|---|')
for r in rows:
    print('| ' + ' | '.join(r) + ' |')
print('\n## tally\n')
for k, v in sorted(tally.items(), key=lambda kv: -kv[1]):
    print(f'* {v:3d}  {k}')
