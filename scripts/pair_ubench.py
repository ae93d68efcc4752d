"""GPU: myolo_conv_pair (csrc/conv_pair.hip) against the two-launch form, per Bottleneck shape of the detect.py frames (hipGraph-timed,
20 back-to-back calls per replay): decides the tile-count gate of the fused kernel.  python scripts/pair_ubench.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from multiyolov5_amd import _lib as L, engine as E  # noqa: E402

DEV = 'cuda:0'
lib = L.lib()


def view(t):
    n, h, w, c = t.shape
    sn, sh, sw, _ = t.stride()
    return L.Tensor(t.data_ptr(), n, h, w, c, sn, sh, sw, L.F16, 0)


def build(Cc, H, W):
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(1, H, W, Cc, generator=g) * 0.7).half().to(DEV)
    y = torch.zeros_like(x); t = torch.zeros_like(x)
    keep = [x, y, t]
    descs = []
    for k in (1, 3):
        w = (torch.randn(Cc, Cc, k, k, generator=g) / (Cc * k * k) ** 0.5).to(DEV)
        wp = torch.zeros(Cc, k * k, Cc, device=DEV, dtype=torch.float16)
        L.check(lib.myolo_pack_weight(L.ptr(w), L.F32, Cc, Cc, k, k, L.ptr(wp), L.F16, Cc, Cc, 0, None, L.stream_ptr()))
        sc, sh = (torch.rand(Cc, generator=g) + 0.5).to(DEV), (torch.randn(Cc, generator=g) * 0.1).to(DEV)
        d = L.ConvDesc()
        d.x, d.y, d.w = (view(x), view(t), wp.data_ptr()) if k == 1 else (view(t), view(y), wp.data_ptr())
        d.cin_pad, d.cout_pad, d.wtaps, d.ntaps, d.stride, d.up_shift = Cc, Cc, k * k, k * k, 1, 0
        E.fill_taps(d, *E.taps_fwd(k, 1, k // 2))
        d.scale, d.shift, d.act = sc.data_ptr(), sh.data_ptr(), L.ACT_SILU
        d.res = view(x) if k == 3 else E.null_tensor()
        keep += [w, wp, sc, sh]
        descs.append(d)
    return descs, keep


def timed(fn, reps=20, iters=30):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters * reps)


print(f'{"C":>4s} {"map":>9s} | {"two launches":>13s} {"fused th4":>10s} {"fused th8":>10s} {"fused auto":>11s}   (us per Bottleneck)')
for Cc, H, W in [(64, 128, 256), (128, 64, 128), (256, 32, 64), (64, 64, 128), (128, 32, 64), (256, 16, 32), (64, 32, 64), (128, 16, 32)]:
    (a, b), keep = build(Cc, H, W)
    res = []
    for mode, th in ((0, 0), (1, 4), (1, 8), (1, 0)):
        lib.myolo_set_option(b'pair_mode', mode)
        lib.myolo_set_option(b'pair_th', th)
        res.append(timed(lambda: L.check(lib.myolo_conv_pair(C.byref(a), C.byref(b), L.stream_ptr()))))
    lib.myolo_set_option(b'pair_mode', 1); lib.myolo_set_option(b'pair_th', 0)
    print(f'{Cc:4d} {H:4d}x{W:<4d} | {res[0]:13.1f} {res[1]:10.1f} {res[2]:10.1f} {res[3]:11.1f}')
