"""Forward convolutions of the training step (batch 16, 512x1024, raw output + BatchNorm statistics epilogue) one layer shape at a time
through the raw C ABI, hipGraph-timed over rotating buffers, against the two rooflines of each shape: algorithmic bytes (input +
weights + output, fp16) at 5 TB/s (what a copy kernel sustains) and the MFMA time at 1.25 PFLOP/s (half the dense fp16 peak).
usage: python scripts/conv_train_ubench.py [B] [cfg]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, '.')
from multiyolov5_amd import _lib as L, engine as E, runtime as R
from multiyolov5_amd.models.yolo import Model

lib = L.lib()
dev = 'cuda'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = sys.argv[2] if len(sys.argv) > 2 else 'yolov5s_city_seg.yaml'
m = Model(os.path.join(os.path.dirname(E.__file__), 'cfg', cfg)).train()
plan = R.PlanHolder(m, [torch.zeros(1, 3, 512, 1024)], ('t', 0), torch.float16, True).plan
shapes = {}
for op in plan.ops:
    if isinstance(op, E.ConvOp) and not op.det and op.x.c % 8 == 0 and op.cout % 4 == 0:
        key = (op.x.c, op.cout, op.k, op.s, op.d, op.x.h, op.x.w)
        shapes[key] = shapes.get(key, 0) + 1


def tdesc(t):
    n, h, w, c = t.shape
    return L.Tensor(t.data_ptr(), n, h, w, c, h * w * c, w * c, c, L.F16, 0)


def run(cin, cout, k, s, d, H, W, iters=16, stats_on=True):
    torch.manual_seed(0)
    Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
    byt = B * (H * W * cin * 2 + Ho * Wo * cout * 2)
    nbuf = max(2, min(12, int(600e6 // byt) + 1))
    xs = [(torch.randn(B, H, W, cin, device=dev) * 0.5).half() for _ in range(nbuf)]
    ys = [torch.zeros(B, Ho, Wo, cout, device=dev, dtype=torch.float16) for _ in range(nbuf)]
    w = (torch.randn(cout, cin, k, k, device=dev) * (1.0 / (cin * k * k) ** 0.5))
    cin_pad, cout_pad = E.rup(cin, 32), E.rup(cout, 32)
    wp = torch.zeros(cout_pad, k * k, cin_pad, device=dev, dtype=torch.float16)
    L.check(lib.myolo_pack_weight(L.ptr(w), L.F32, cout, cin, k, k, L.ptr(wp), L.F16, cout_pad, cin_pad, 0, None, L.stream_ptr()))
    stats = torch.zeros(L.STAT_COPIES * 2 * cout, device=dev)
    descs = []
    for x, y in zip(xs, ys):
        dd = L.ConvDesc()
        dd.x, dd.y, dd.w = tdesc(x), tdesc(y), wp.data_ptr()
        dd.cin_pad, dd.cout_pad, dd.wtaps, dd.ntaps, dd.stride, dd.up_shift = cin_pad, cout_pad, k * k, k * k, s, 0
        E.fill_taps(dd, *E.taps_fwd(k, d, d * (k // 2)))
        dd.res = E.null_tensor()
        dd.act, dd.stats = L.ACT_NONE, (stats.data_ptr() if stats_on else None)
        descs.append(dd)
    sp = L.stream_ptr()
    for dd in descs[:2]:
        L.check(lib.myolo_conv(C.byref(dd), sp))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        spc = L.stream_ptr()
        for i in range(iters):
            L.check(lib.myolo_conv(C.byref(descs[i % nbuf]), spc))
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * iters)


if os.environ.get('PROBE') == 'one':
    # one shape given as PROBE_SHAPE="cin,cout,k,s,d,H,W"
    shp = tuple(int(v) for v in os.environ['PROBE_SHAPE'].split(','))
    print(shp, f'{run(*shp):.1f} us with stats, {run(*shp, stats_on=False):.1f} without', flush=True)
    sys.exit(0)
if os.environ.get('PROBE') == 'mid':
    # conv_mid (LDS-DMA ring, 8 waves) against the kernels it replaces, forward (statistics) and dgrad-shaped (no statistics) calls
    P = [(128, 128, 3, 1, 1, 32, 64), (128, 128, 1, 1, 1, 32, 64), (256, 256, 1, 1, 1, 32, 64), (256, 128, 1, 1, 1, 32, 64), (512, 256, 1, 1, 1, 32, 64),
         (512, 512, 1, 1, 1, 16, 32), (256, 256, 1, 1, 1, 16, 32), (256, 256, 3, 1, 1, 16, 32), (1024, 512, 1, 1, 1, 16, 32), (512, 256, 1, 1, 1, 16, 32),
         (256, 128, 3, 1, 1, 64, 128), (128, 256, 3, 1, 1, 64, 128), (128, 256, 3, 2, 1, 64, 128), (256, 512, 3, 2, 1, 32, 64),
         (64, 64, 3, 1, 1, 64, 128), (128, 128, 1, 1, 1, 64, 128), (64, 64, 1, 1, 1, 64, 128), (256, 128, 1, 1, 1, 64, 128), (64, 128, 3, 2, 1, 128, 256),
         (64, 64, 1, 1, 1, 128, 256)]
    V = [('old', {'mid_mode': 0}), ('mid', {'mid_mode': 2}), ('v5', {'mid_mode': 2, 'mid_var': 5}), ('v2', {'mid_mode': 2, 'mid_var': 2}), ('v3', {'mid_mode': 2, 'mid_var': 3}),
         ('v4', {'mid_mode': 2, 'mid_var': 4})]
    for shp in P:
        cin, cout, k, s, d, H, W = shp
        M = B * ((H + s - 1) // s) * ((W + s - 1) // s)
        byt = B * H * W * cin * 2 + M * cout * 2 + cout * cin * k * k * 2
        fl = 2.0 * M * cout * cin * k * k
        ideal = max(byt / 8e12, fl / 2.5e15) * 1e6
        row = []
        for name, opts in V:
            for kk, vv in {'mid_mode': 1, 'mid_var': 0, 'midx_mode': 0, **opts}.items():
                lib.myolo_set_option(kk.encode(), vv)
            row.append(f'{name} {run(*shp):.1f}/{run(*shp, stats_on=False):.1f}')
        for kk, vv in {'mid_mode': 2, 'mid_var': 0, 'midx_mode': 1}.items():
            lib.myolo_set_option(kk.encode(), vv)
        print(f'{str(shp):34s} roofline {ideal:5.1f} us | ' + ' | '.join(row), '(us with stats / without)', flush=True)
    sys.exit(0)
if os.environ.get('PROBE') == 'midx':
    # conv_midx (input halo resident in LDS, weights streamed) against conv_mid on the k x k stride-1 layers
    P = [(64, 64, 3, 1, 1, 64, 128), (128, 128, 3, 1, 1, 32, 64), (256, 256, 3, 1, 1, 16, 32), (256, 128, 3, 1, 1, 64, 128), (128, 256, 3, 1, 1, 64, 128),
         (64, 64, 3, 1, 2, 64, 128), (64, 64, 3, 1, 3, 64, 128)]
    V = [('mid', {'midx_mode': 0}), ('midx', {'midx_mode': 1}), ('x1', {'midx_mode': 1, 'midx_var': 1}), ('x2', {'midx_mode': 1, 'midx_var': 2}), ('x3', {'midx_mode': 1, 'midx_var': 3})]
    for shp in P:
        row = []
        for name, opts in V:
            for kk, vv in {'midx_mode': 0, 'midx_var': 0, **opts}.items():
                lib.myolo_set_option(kk.encode(), vv)
            row.append(f'{name} {run(*shp):.1f}/{run(*shp, stats_on=False):.1f}')
        for kk, vv in {'midx_mode': 1, 'midx_var': 0}.items():
            lib.myolo_set_option(kk.encode(), vv)
        print(f'{str(shp):34s} ' + ' | '.join(row), '(us with stats / without)', flush=True)
    sys.exit(0)
if os.environ.get('PROBE') == 'mid_dbg':
    # where conv_mid's time goes: dbg bits 1 no steady-state loads, 2 no fragment reads, 4 no MFMAs, 8 no stores; tile variants
    P = [(128, 128, 3, 1, 1, 32, 64), (256, 128, 3, 1, 1, 64, 128), (256, 256, 1, 1, 1, 32, 64), (512, 512, 1, 1, 1, 16, 32), (128, 128, 1, 1, 1, 32, 64)]
    V = [('v1', 1, 0), ('noload', 1, 1), ('noread', 1, 2), ('nomfma', 1, 4), ('nostore', 1, 8), ('load only', 1, 6), ('mfma only', 1, 3),
         ('skeleton', 1, 15), ('skel-nobar', 1, 31), ('nobar', 1, 16), ('mfma only nobar', 1, 19), ('v4 256x128', 4, 0), ('v4 noload', 4, 1), ('v4 nomfma', 4, 4), ('v5 4-stage', 5, 0), ('v3 64x128', 3, 0)]
    for shp in P:
        row = []
        for name, var, dbg in V:
            for kk, vv in {'mid_mode': 2, 'mid_var': var, 'mid_dbg': dbg}.items():
                lib.myolo_set_option(kk.encode(), vv)
            row.append(f'{name} {run(*shp):.1f}/{run(*shp, stats_on=False):.1f}')
        for kk, vv in {'mid_mode': 1, 'mid_var': 0, 'mid_dbg': 0}.items():
            lib.myolo_set_option(kk.encode(), vv)
        print(f'{str(shp):30s} ' + ' | '.join(row), flush=True)
    sys.exit(0)
if os.environ.get('PROBE') == 'igemm':
    # LDS-tiled kernel on the mid-size layers: tile shape / grid experiments
    for shp in [(256, 256, 1, 1, 1, 32, 64), (128, 128, 1, 1, 1, 32, 64), (256, 128, 1, 1, 1, 32, 64), (512, 512, 1, 1, 1, 16, 32), (512, 256, 1, 1, 1, 16, 32),
                (256, 256, 1, 1, 1, 16, 32), (1024, 512, 1, 1, 1, 16, 32), (512, 256, 1, 1, 1, 32, 64), (128, 128, 3, 1, 1, 32, 64), (256, 256, 3, 1, 1, 16, 32)]:
        row = []
        for bm, bn, wgs in [(0, 0, 0), (64, 0, 0), (128, 0, 0), (64, 64, 0), (128, 64, 0), (64, 0, 1024), (64, 64, 1536), (128, 0, 512)]:
            for kk, vv in (('igemm_bm', bm), ('igemm_bn', bn), ('igemm_wgs', wgs)):
                lib.myolo_set_option(kk.encode(), vv)
            row.append(f'{bm}/{bn}/{wgs}: {run(*shp):.1f}')
        for kk in ('igemm_bm', 'igemm_bn', 'igemm_wgs'):
            lib.myolo_set_option(kk.encode(), 0)
        print(shp, ' | '.join(row), flush=True)
    sys.exit(0)
if os.environ.get('PROBE') == 'mintiles':
    # mid-size layers: LDS-tiled kernel (stream_min_tiles 2048) against the streaming kernel (stream_min_tiles 1)
    for shp in [(256, 256, 1, 1, 1, 32, 64), (128, 128, 1, 1, 1, 32, 64), (256, 128, 1, 1, 1, 32, 64), (512, 512, 1, 1, 1, 16, 32), (512, 256, 1, 1, 1, 16, 32),
                (256, 256, 1, 1, 1, 16, 32), (1024, 512, 1, 1, 1, 16, 32), (512, 256, 1, 1, 1, 32, 64), (128, 128, 3, 1, 1, 32, 64), (256, 256, 3, 1, 1, 16, 32)]:
        row = []
        for mt in (2048, 1):
            lib.myolo_set_option(b'stream_min_tiles', mt)
            row.append(f'min_tiles {mt}: {run(*shp):.1f}')
        lib.myolo_set_option(b'stream_min_tiles', 2048)
        print(shp, ' | '.join(row), flush=True)
    sys.exit(0)
if os.environ.get('PROBE') == 'dbg':
    # streaming kernel with its profiling switches: 1 = no stores, 2 = no activation loads, 3 = neither
    for shp in [(256, 256, 1, 1, 1, 32, 64), (128, 128, 1, 1, 1, 64, 128), (64, 64, 1, 1, 1, 128, 256), (512, 512, 1, 1, 1, 16, 32)]:
        row = []
        for dbg in (0, 1, 2, 3, 4, 6):
            lib.myolo_set_option(b'stream_dbg', dbg)
            row.append(f'dbg{dbg} {run(*shp, stats_on=False):.1f}')
        lib.myolo_set_option(b'stream_dbg', 0)
        print(shp, ' | '.join(row), flush=True)
    sys.exit(0)
if os.environ.get('PROBE'):
    # a few mid-size shapes under each kernel family, with and without the statistics epilogue
    P = [(256, 256, 1, 1, 1, 32, 64), (64, 64, 3, 1, 1, 64, 128), (128, 128, 1, 1, 1, 64, 128), (64, 64, 1, 1, 1, 128, 256), (128, 128, 3, 1, 1, 32, 64),
         (512, 512, 1, 1, 1, 16, 32)]
    V = [('default', {}), ('no_halo', {'halo_off': 1}), ('no_stream', {'stream_off': 1}), ('igemm', {'halo_off': 1, 'stream_off': 1, 'small_off': 1})]
    for shp in P:
        row = []
        for name, opts in V:
            for kk, vv in {'halo_off': 0, 'stream_off': 0, 'small_off': 0, **opts}.items():
                lib.myolo_set_option(kk.encode(), vv)
            row.append(f'{name} {run(*shp):.1f}/{run(*shp, stats_on=False):.1f}')
        print(shp, ' | '.join(row), '(us with stats / without)', flush=True)
    sys.exit(0)
print(f'batch {B}, {cfg}: {len(shapes)} distinct conv shapes ({sum(shapes.values())} launches)')
print(f'{"shape":34s} {"n":>2s} {"M":>8s} | {"us":>7s} {"TB/s":>5s} {"TF/s":>6s} | {"hbm us":>6s} {"mfma us":>7s} | x ideal')
tot, tot_ideal = 0.0, 0.0
for (cin, cout, k, s, d, H, W), cnt in sorted(shapes.items(), key=lambda kv: -kv[0][5] * kv[0][6] * kv[0][0]):
    Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
    M = B * Ho * Wo
    byt = B * H * W * cin * 2 + M * cout * 2 + cout * cin * k * k * 2
    fl = 2.0 * M * cout * cin * k * k
    us = run(cin, cout, k, s, d, H, W)
    hb, mf = byt / 5e12 * 1e6, fl / 1.25e15 * 1e6
    ideal = max(hb, mf, 4.7)
    tot += us * cnt
    tot_ideal += ideal * cnt
    print(f'{cin:4d}->{cout:4d} k{k} s{s} d{d} {H:4d}x{W:4d}        {cnt:2d} {M:8d} | {us:7.1f} {byt / us / 1e6:5.2f} {fl / us / 1e6:6.1f} | {hb:6.1f} {mf:7.1f} | '
          f'{us / ideal:4.1f}', flush=True)
print(f'sum over the forward: {tot:.0f} us measured, {tot_ideal:.0f} us at the rooflines')
