"""BatchNorm+SiLU forward / backward-reduce / backward-apply launches of the training step (batch 16, 512x1024) through the raw C ABI,
hipGraph-timed over rotating buffers (> the 256 MB MALL): us per launch and the HBM rate each reaches (fp16: fwd 4 B/element,
reduce 4 B, apply 6 B; +2 B with the residual / residual-gradient tensor).  usage: python scripts/bn_ubench.py [quick]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, '.')
from multiyolov5_amd import _lib as L

lib = L.lib()
dev = 'cuda'
SHAPES = [(16, 256, 512, 32), (16, 128, 256, 64), (16, 64, 128, 128), (16, 64, 128, 64), (16, 32, 64, 256), (16, 32, 64, 128), (16, 16, 32, 512),
          (16, 16, 32, 256), (16, 16, 32, 128)]
if len(sys.argv) > 1 and sys.argv[1] == 'quick':
    SHAPES = SHAPES[1::3]


def td(t):
    n, h, w, c = t.shape
    return L.Tensor(t.data_ptr(), n, h, w, c, h * w * c, w * c, c, L.F16, 0)


def timeit(fn, nbuf, iters=24):
    sp = L.stream_ptr()
    for i in range(min(2, nbuf)):
        fn(i, sp)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        spc = L.stream_ptr()
        for i in range(iters):
            fn(i % nbuf, spc)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * iters)


tot = [0.0, 0.0, 0.0, 0.0]
print(f'{"shape":22s} {"MB":>7s} | {"fwd us":>7s} {"TB/s":>5s} | {"fwd+res":>7s} {"TB/s":>5s} | {"reduce":>7s} {"TB/s":>5s} | {"apply":>7s} {"TB/s":>5s}')
for (n, h, w, c) in SHAPES:
    torch.manual_seed(0)
    by = n * h * w * c * 2
    nbuf = max(3, min(24, int(600e6 // (3 * by)) + 1))
    ys = [(torch.randn(n, h, w, c, device=dev) * 1.5 + 0.3).half() for _ in range(nbuf)]
    gs = [(torch.randn(n, h, w, c, device=dev) * 0.01).half() for _ in range(nbuf)]
    outs = [torch.empty(n, h, w, c, device=dev, dtype=torch.float16) for _ in range(nbuf)]
    M = n * h * w
    stats = torch.zeros(L.STAT_COPIES, 2, c, device=dev)
    yf = ys[0].float().view(-1, c)
    stats[0, 0], stats[0, 1] = yf.sum(0), (yf * yf).sum(0)
    gam, bet = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
    rm, rv, nbt = torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.zeros(1, device=dev, dtype=torch.int64)
    saved = torch.zeros(2, c, device=dev)
    dsum = torch.zeros(L.STAT_COPIES, 2, c, device=dev)
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    yd, gd, od = [td(t) for t in ys], [td(t) for t in gs], [td(t) for t in outs]
    null = L.Tensor(None, 0, 0, 0, 0, 0, 0, 0, 0, 0)

    def fwd(i, sp, res=False):
        L.check(lib.myolo_bn_act_fwd(C.byref(yd[i]), L.ptr(stats), L.ptr(gam), L.ptr(bet), L.ptr(rm), L.ptr(rv), L.ptr(nbt), L.ptr(saved),
                                     1e-3, 0.03, L.ACT_SILU, C.byref(gd[i]) if res else C.byref(null), C.byref(od[i]), sp))

    def red(i, sp):
        L.check(lib.myolo_bn_act_bwd_reduce(C.byref(gd[i]), C.byref(yd[i]), L.ptr(saved), L.ptr(gam), L.ptr(bet), L.ACT_SILU, L.ptr(dsum), sp))

    def app(i, sp):
        L.check(lib.myolo_bn_act_bwd_apply(C.byref(gd[i]), C.byref(yd[i]), L.ptr(saved), L.ptr(gam), L.ptr(bet), L.ACT_SILU, L.ptr(dsum), L.ptr(dg),
                                           L.ptr(db), C.byref(od[i]), C.byref(null), 0, sp))

    t = [timeit(fwd, nbuf), timeit(lambda i, sp: fwd(i, sp, True), nbuf), timeit(red, nbuf), timeit(app, nbuf)]
    bts = [2 * by, 3 * by, 2 * by, 3 * by]
    for k in range(4):
        tot[k] += t[k]
    print(f'{n}x{h}x{w}x{c:<10d} {by / 1e6:7.1f} | ' + ' | '.join(f'{t[k]:7.1f} {bts[k] / t[k] / 1e6:5.2f}' for k in range(4)))
print('sum us: fwd %.1f fwd+res %.1f reduce %.1f apply %.1f' % tuple(tot))
