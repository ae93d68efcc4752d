"""In-situ A/B of the convolution kernel families, launch by launch: the real launch plan (eval frame or one training step) is issued call
by call with HIP events around every myolo_conv / myolo_conv_dgrad_s2, once per variant of the dispatch options, and the per-launch times
are printed side by side (the data the dispatch thresholds of conv_igemm.hip / conv_halo.hip / conv_stream.hip / conv_small.hip come from).
usage: python scripts/conv_algo_ab.py eval|train B H W [cfg] [reps]"""
import os
import sys

import torch

sys.path.insert(0, '.')
os.environ['MYOLO_GRAPH'] = '0'
os.environ['MYOLO_NATIVE_EXEC'] = '0'
from multiyolov5_amd import _lib as L, engine as E, synth  # noqa: E402
from multiyolov5_amd.models.yolo import Model  # noqa: E402

mode, B, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = sys.argv[5] if len(sys.argv) > 5 else 'yolov5s_city_seg.yaml'
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 12
dev = torch.device('cuda', 0)
lib = L.lib()
VARIANTS = [('default', {}), ('no_halo', {'halo_off': 1}), ('no_stream', {'stream_off': 1}), ('no_small', {'small_off': 1}),
            ('igemm', {'halo_off': 1, 'stream_off': 1, 'small_off': 1})]
RESET = {'halo_off': 0, 'stream_off': 0, 'small_off': 0}

m = Model(os.path.join('multiyolov5_amd', 'cfg', cfg))
synth.randomize_(m, seed=0)
img = synth.images(B, H, W, seed=7).to(dev, torch.float16)
if mode == 'eval':
    m = m.to(dev).half().fuse().eval()

    def step():
        with torch.no_grad():
            m(img)
else:
    import argparse
    import bench
    tr = bench.Trainer(argparse.Namespace(img=(H, W), batch=B, cfg=cfg, dtype='f16', stage='fwdbwd'), 1, 0, dev)
    step = tr.step

rec = []
orig = E.Call.__call__


def timed(self, st):
    if self.name not in ('myolo_conv', 'myolo_conv_dgrad_s2', 'myolo_conv_dgrad_bn'):
        return orig(self, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(self, st)
    e1.record()
    rec.append((self, e0, e1))


def describe(c):
    if c.name == 'myolo_conv':
        k = c.args[0]._obj
        return (f'{k.x.h}x{k.x.w}x{k.x.c}->{k.y.c} t{k.ntaps} s{k.stride}' + (' up' if k.up_shift else '') + (' acc' if k.accumulate else '')
                + (' st' if k.stats else '') + (f' bnb{k.nbnb}' if k.nbnb else '') + (' res' if k.res.ptr else ''), k.y.n * k.y.h * k.y.w)
    ds = E._s2_descs(c)
    k = ds[0]
    return f'dgrad_s2 {k.x.h}x{k.x.w}x{k.x.c}->{k.y.c} x{len(ds)}', k.y.n * k.y.h * k.y.w * len(ds)


for _ in range(3):
    step()
torch.cuda.synchronize()
E.Call.__call__ = timed
table, calls = {}, None
for name, opts in VARIANTS:
    for k, v in {**RESET, **opts}.items():
        lib.myolo_set_option(k.encode(), v)
    step()
    torch.cuda.synchronize()
    rec.clear()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    n = len(rec) // reps
    ts = [0.0] * n
    for i, (c, e0, e1) in enumerate(rec):
        ts[i % n] += e0.elapsed_time(e1) * 1e3 / reps
    table[name] = ts
    calls = [r[0] for r in rec[:n]]
E.Call.__call__ = orig
names = [v[0] for v in VARIANTS]
print(f'{mode} {B}x{H}x{W} {cfg}: {len(calls)} conv launches, us per launch (events, eager issue)')
print(f'{"#":>3s} {"launch":44s} {"M":>8s} | ' + ' '.join(f'{n:>9s}' for n in names) + ' | best')
tot = {n: 0.0 for n in names}
best_tot = 0.0
for i, c in enumerate(calls):
    d, M = describe(c)
    row = [table[n][i] for n in names]
    for n, t in zip(names, row):
        tot[n] += t
    b = min(range(len(row)), key=lambda j: row[j])
    best_tot += row[b]
    flag = '' if row[0] <= row[b] * 1.03 else f'  <-- {names[b]} -{row[0] - row[b]:.1f}us'
    print(f'{i:3d} {d:44s} {M:8d} | ' + ' '.join(f'{t:9.1f}' for t in row) + f' | {names[b]}{flag}')
print('total us: ' + ' '.join(f'{n}={tot[n]:.0f}' for n in names) + f' per-launch-best={best_tot:.0f}')
