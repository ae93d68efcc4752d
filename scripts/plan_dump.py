#!/usr/bin/env python
"""dry (CPU) build of a launch plan: one line per forward launch with its geometry.  usage: plan_dump.py [tag] [B H W] [train|eval] [fuse]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiyolov5_amd import runtime as R, engine as E
from multiyolov5_amd.models.yolo import Model
cfg = sys.argv[1] if len(sys.argv) > 1 else 'yolov5s_city_seg.yaml'
B, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (1, 512, 1024)
training = len(sys.argv) > 5 and sys.argv[5] == 'train'
m = Model(os.path.join(os.path.dirname(E.__file__), 'cfg', cfg))
if not training:
    m = m.fuse() if 'fuse' in sys.argv else m
m.train(training)
plan = R.PlanHolder(m, [torch.zeros(B, 3, H, W)], ('t', 0), torch.float16, training).plan
n = 0
for op in plan.ops:
    for c in op.fwd_calls:
        n += 1
        d = ''
        if c.name == 'myolo_conv':
            k = c.args[0]._obj
            d = f'x {k.x.n}x{k.x.h}x{k.x.w}x{k.x.c} -> y {k.y.h}x{k.y.w}x{k.y.c} taps {k.ntaps} s{k.stride} M={k.y.n*k.y.h*k.y.w} K={k.ntaps*k.x.c} bytes={E.conv_call_bytes(c)/1e6:.2f}MB'
        else:
            for a in c.args:
                o = getattr(a, '_obj', None)
                if isinstance(o, E.CT):
                    d += f' [{o.n}x{o.h}x{o.w}x{o.c}]'
        print(f'{n:3d} {type(op).__name__:12s} {c.name:32s} {d}')
if training:
    print('bwd launches', sum(len(op.bwd_calls) for op in plan.ops))
