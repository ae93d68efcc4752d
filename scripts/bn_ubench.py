"""graph-captured timing of the three BatchNorm passes (no host launch overhead) next to a plain device copy of the same bytes --
what a streaming kernel of that size can reach on this part.  usage: python scripts/bn_ubench.py"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiyolov5_amd import _lib as L

dev = torch.device('cuda:0')
lib = L.lib()
SHAPES = [(16, 256, 512, 32), (16, 128, 256, 64), (16, 64, 128, 128), (16, 64, 128, 64), (16, 32, 64, 128), (16, 16, 32, 512), (16, 16, 32, 256)]
NB = 6


def view(t):
    n, h, w, c = t.shape
    return L.Tensor(L.ptr(t), n, h, w, c, h * w * c, w * c, c, L.DT[t.dtype], 0)


def graph_time(fn, iters=24):
    fn(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (3 * iters)


print('shape                 MB/tensor |  copy(2 passes)   fwd(2)   fwd-noprologue   reduce(2)   apply(3)  apply-noprologue  [us, GB/s]')
for (n, h, w, c) in SHAPES:
    ys = [(torch.randn(n, h, w, c, device=dev) * 0.5).half() for _ in range(NB)]
    gs = [(torch.randn(n, h, w, c, device=dev) * 0.1).half() for _ in range(NB)]
    os_ = [torch.empty_like(ys[0]) for _ in range(NB)]
    stats = torch.rand(L.STAT_COPIES * 2 * c, device=dev) + 1.0
    gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
    saved = torch.cat([torch.randn(c, device=dev) * 0.1, torch.rand(c, device=dev) + 0.5])
    dsum = torch.zeros(L.STAT_COPIES * 2 * c, device=dev)
    dgam, dbet = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    null = L.Tensor()
    yd, gd, od = [view(t) for t in ys], [view(t) for t in gs], [view(t) for t in os_]

    def cp(i):
        os_[i % NB].copy_(ys[i % NB])

    def fwd(i):
        k = i % NB
        L.check(lib.myolo_bn_act_fwd(C.byref(yd[k]), L.ptr(stats), L.ptr(gamma), L.ptr(beta), None, None, None, L.ptr(saved),
                                     C.c_float(1e-3), C.c_float(0.03), 1, C.byref(null), C.byref(od[k]), L.stream_ptr()), 'fwd')

    def red(i):
        k = i % NB
        L.check(lib.myolo_bn_act_bwd_reduce(C.byref(gd[k]), C.byref(yd[k]), L.ptr(saved), L.ptr(gamma), L.ptr(beta), 1, L.ptr(dsum),
                                            L.stream_ptr()), 'red')

    def app(i):
        k = i % NB
        L.check(lib.myolo_bn_act_bwd_apply(C.byref(gd[k]), C.byref(yd[k]), L.ptr(saved), L.ptr(gamma), L.ptr(beta), 1, L.ptr(dsum),
                                           L.ptr(dgam), L.ptr(dbet), C.byref(od[k]), C.byref(null), 0, L.stream_ptr()), 'app')
    def fwd0(i):           # gamma = NULL: activation only, no statistics prologue -> the floor of the streaming part
        k = i % NB
        L.check(lib.myolo_bn_act_fwd(C.byref(yd[k]), None, None, None, None, None, None, None, C.c_float(0), C.c_float(0), 1,
                                     C.byref(null), C.byref(od[k]), L.stream_ptr()), 'fwd0')

    def app0(i):
        k = i % NB
        L.check(lib.myolo_bn_act_bwd_apply(C.byref(gd[k]), C.byref(yd[k]), None, None, None, 1, None, None, None, C.byref(od[k]),
                                           C.byref(null), 0, L.stream_ptr()), 'app0')
    e = n * h * w * c * 2
    out = []
    for f, units in ((cp, 2), (fwd, 2), (fwd0, 2), (red, 2), (app, 3), (app0, 3)):
        t = graph_time(f)
        out.append(f'{t:6.1f} {e * units / t / 1e3:5.0f}')
    print(f'{str((n, h, w, c)):22s} {e / 1e6:6.1f}   | ' + '   '.join(out), flush=True)
