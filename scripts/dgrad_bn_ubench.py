"""myolo_conv_dgrad_bn (BatchNorm-backward apply pass in the operand path of a 1x1 dgrad) against the two launches it replaces, per layer
shape of the training step (batch 16), hipGraph-timed over rotating buffers.  usage: python scripts/dgrad_bn_ubench.py"""
import ctypes as C
import sys

import torch

sys.path.insert(0, '.')
from multiyolov5_amd import _lib as L, engine as E

lib = L.lib()
dev = 'cuda'


def td(t):
    n, h, w, c = t.shape
    return L.Tensor(t.data_ptr(), n, h, w, c, h * w * c, w * c, c, L.F16, 0)


def run(K, N, H, W, B=16, iters=12, acc=True):
    torch.manual_seed(0)
    nbuf = max(2, min(8, int(400e6 // (B * H * W * (3 * K + N) * 2)) + 1))
    gs = [(torch.randn(B, H, W, K, device=dev) * 0.3).half() for _ in range(nbuf)]
    ys = [torch.randn(B, H, W, K, device=dev).half() for _ in range(nbuf)]
    dys = [torch.zeros(B, H, W, K, device=dev, dtype=torch.float16) for _ in range(nbuf)]
    gxs = [torch.zeros(B, H, W, N, device=dev, dtype=torch.float16) for _ in range(nbuf)]
    saved = torch.cat([torch.randn(K, device=dev) * 0.2, torch.rand(K, device=dev) + 0.5])
    gam, bet = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
    dsum = torch.randn(L.STAT_COPIES * 2 * K, device=dev)
    dg, db = torch.zeros(K, device=dev), torch.zeros(K, device=dev)
    wt = torch.randn(N, K, 1, 1, device=dev) * (1.0 / K ** 0.5)
    wp = torch.zeros(E.rup(N, 32), 1, K, device=dev, dtype=torch.float16)
    L.check(lib.myolo_pack_weight(L.ptr(wt), L.F32, N, K, 1, 1, L.ptr(wp), L.F16, E.rup(N, 32), K, 0, None, L.stream_ptr()))
    none = E.null_tensor()
    keep = []

    def descs(fused, i):
        d = L.ConvDesc()
        d.x, d.y, d.w = td(gs[i] if fused else dys[i]), td(gxs[i]), wp.data_ptr()
        d.cin_pad, d.cout_pad, d.wtaps, d.ntaps, d.stride, d.up_shift = K, E.rup(N, 32), 1, 1, 1, 0
        E.fill_taps(d, [0], [0], [0])
        d.res, d.act, d.accumulate = E.null_tensor(), L.ACT_NONE, int(acc)
        f = L.BnApplyFold()
        f.y, f.dy = td(ys[i]), td(dys[i])
        f.saved, f.gamma, f.beta, f.dsum, f.dgamma, f.dbeta, f.act = saved.data_ptr(), gam.data_ptr(), bet.data_ptr(), dsum.data_ptr(), dg.data_ptr(), \
            db.data_ptr(), L.ACT_SILU
        keep.append((d, f))
        return d, f

    def issue(fused, i, sp):
        d, f = descs(fused, i)
        if fused:
            L.check(lib.myolo_conv_dgrad_bn(C.byref(d), C.byref(f), sp))
        else:
            L.check(lib.myolo_bn_act_bwd_apply(C.byref(td(gs[i])), C.byref(td(ys[i])), L.ptr(saved), L.ptr(gam), L.ptr(bet), L.ACT_SILU, L.ptr(dsum),
                                               L.ptr(dg), L.ptr(db), C.byref(td(dys[i])), C.byref(none), 0, sp))
            L.check(lib.myolo_conv(C.byref(d), sp))
    out = []
    for fused in (False, True):
        issue(fused, 0, L.stream_ptr())
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(iters):
                issue(fused, i % nbuf, L.stream_ptr())
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / (3 * iters))
    return out


lib.myolo_set_option(b'mid_bna_strict', 1)
for K, N, H, W in [(128, 128, 32, 64), (256, 256, 32, 64), (128, 256, 32, 64), (512, 512, 16, 32), (256, 512, 16, 32), (512, 1024, 16, 32), (256, 256, 16, 32),
                   (128, 128, 64, 128), (128, 256, 64, 128)]:
    two, one = run(K, N, H, W)
    print(f'K {K:4d} -> N {N:4d} @ {H}x{W}: apply + dgrad {two:6.1f} us | dgrad_bn {one:6.1f} us', flush=True)
