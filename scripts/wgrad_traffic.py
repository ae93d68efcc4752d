#!/usr/bin/env python
"""dry (CPU) estimate of the bytes the weight-gradient launches of one training step ISSUE against their algorithmic operand bytes:
x is read once per co block (with the halo of the tile height the LDS-DMA ring picks), dy once per ci block (conv_wgrad_tile.hip).
The round-4 PMC pass counted 5.27 GB fetched per step for 3.25 GB of operands -- this script reproduces that figure (5.1 GB) from the
launch list alone, i.e. none of the re-reads hits an L2 today.  usage: wgrad_traffic.py [cfg] [B H W]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiyolov5_amd import runtime as R, engine as E, _lib as L
from multiyolov5_amd.models.yolo import Model

cfg = sys.argv[1] if len(sys.argv) > 1 else 'yolov5s_city_seg.yaml'
B, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (16, 512, 1024)
m = Model(os.path.join(os.path.dirname(E.__file__), 'cfg', cfg))
m.train(True)
plan = R.PlanHolder(m, [torch.zeros(B, 3, H, W)], ('t', 0), torch.float16, True).plan
rows, tot_alg, tot_iss = [], 0, 0
for op in plan.ops:
    for c in op.bwd_calls:
        if getattr(c, 'name', '') != 'myolo_conv_wgrad':
            continue
        d = next(a._obj for a in c.args if isinstance(getattr(a, '_obj', None), L.WgradDesc))
        cin, cout = (d.cin if d.cin > 0 else d.x.c), (d.cout if d.cout > 0 else d.dy.c)
        nt, s = d.ntaps, d.stride
        xb, db = d.x.n * d.x.h * d.x.w * cin * 2, d.dy.n * d.dy.h * d.dy.w * cout * 2
        if cin <= 16 and nt == 9:                          # Focus-sized: compact kernel, one block
            rows.append((xb + db, xb + db, f'{cin}->{cout} t{nt} s{s} {d.dy.h}x{d.dy.w} (compact kernel)'))
            tot_alg += xb + db
            tot_iss += xb + db
            continue
        big = 4 if nt == 1 else 2
        cof = big if cout > 64 else (2 if cout > 32 else 1)
        cif = big if cin > 64 else (2 if cin > 32 else 1)
        tco, tci = -(-cout // (32 * cof)), -(-cin // (32 * cif))
        ext_y = max(d.tap_dy[t] for t in range(nt)) - min(d.tap_dy[t] for t in range(nt))
        ext_x = max(d.tap_dx[t] for t in range(nt)) - min(d.tap_dx[t] for t in range(nt))
        th = 3 if nt > 1 else 2                            # typical tile heights of the 3- / 4-stage ring
        halo = ((th - 1) * s + 1 + ext_y) * (31 * s + 1 + ext_x) / (th * s * 32 * s)
        iss = xb * halo * tco + db * tci
        rows.append((xb + db, iss, f'{cin}->{cout} t{nt} s{s} {d.dy.h}x{d.dy.w} blocks {tco}x{tci} halo x{halo:.2f}'))
        tot_alg += xb + db
        tot_iss += iss
rows.sort(key=lambda r: -(r[1] - r[0]))
for alg, iss, what in rows[:30]:
    print(f'{alg / 1e6:8.1f} MB operands {iss / 1e6:8.1f} MB issued  x{iss / alg:.2f}  {what}')
print(f'{len(rows)} launches: operands {tot_alg / 1e9:.2f} GB, issued {tot_iss / 1e9:.2f} GB per step')
