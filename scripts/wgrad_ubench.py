"""micro-benchmark of myolo_conv_wgrad through the raw C ABI (B=16, fp16): hipGraph-timed launches over rotating buffers, the LDS-tile
kernel (conv_wgrad_tile.hip) with its LDS-DMA loaders vs the register loaders (`wgrad_tile_dma` 0) [vs split / ring variants: `sweep`]
[vs the round-1 kernels (`wgrad_tile_off`): `r1`], numerics of every variant vs autograd (fp32).  usage: wgrad_ubench.py [quick] [sweep] [r1] [focus]"""
import ctypes as C
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, '.')
from multiyolov5_amd import _lib as L
from multiyolov5_amd import engine as E

lib = L.lib()
dev = 'cuda'
B = 16
# (cin, cout, k, dil, stride, Hin, Win)
SHAPES = [
    (64, 64, 3, 1, 1, 64, 128), (128, 128, 3, 1, 1, 32, 64), (32, 32, 3, 1, 1, 128, 256), (256, 128, 3, 1, 1, 64, 128),
    (64, 64, 3, 2, 1, 64, 128), (32, 64, 3, 1, 2, 256, 512), (64, 128, 3, 1, 2, 128, 256), (256, 256, 3, 1, 1, 16, 32),
    (64, 64, 1, 1, 1, 64, 128), (128, 128, 1, 1, 1, 64, 128), (256, 128, 1, 1, 1, 64, 128), (384, 64, 1, 1, 1, 64, 128),
    (64, 32, 1, 1, 1, 128, 256), (256, 128, 1, 1, 1, 32, 64), (512, 256, 1, 1, 1, 16, 32), (1024, 512, 1, 1, 1, 16, 32),
]
if 'focus' in sys.argv[1:]:          # the Focus layer's 3x3 (compact kernels of conv_wgrad.hip)
    SHAPES = [(16, 32, 3, 1, 1, 256, 512), (16, 32, 3, 1, 1, 64, 128)]
if 'quick' in sys.argv[1:]:
    SHAPES = SHAPES[:3] + SHAPES[8:10]


def tdesc(t):
    n, h, w, c = t.shape
    return L.Tensor(t.data_ptr(), n, h, w, c, h * w * c, w * c, c, L.F16, 0)


ws = torch.empty(48 << 20 >> 2, dtype=torch.float32, device=dev)


def run(cin, cout, k, d, s, H, W, nbuf, iters=10):
    torch.manual_seed(0)
    pad = d * (k // 2)
    Ho, Wo = (H + 2 * pad - d * (k - 1) - 1) // s + 1, (W + 2 * pad - d * (k - 1) - 1) // s + 1
    xs = [(torch.randn(B, H, W, cin, device=dev) * 0.5).half() for _ in range(nbuf)]
    dys = [(torch.randn(B, Ho, Wo, cout, device=dev) * 0.1).half() for _ in range(nbuf)]
    dw = torch.zeros(cout, cin, k, k, device=dev)
    descs = []
    for x, dy in zip(xs, dys):
        wd = L.WgradDesc()
        wd.x, wd.dy = tdesc(x), tdesc(dy)
        wd.dw, wd.db = dw.data_ptr(), None
        wd.ntaps, wd.stride, wd.up_shift, wd.ksplit, wd.cout, wd.cin = k * k, s, 0, 0, cout, cin
        wd.ws, wd.ws_bytes = ws.data_ptr(), ws.numel() * 4
        tdy, tdx, _ = E.taps_fwd(k, d, pad)
        E.fill_taps(wd, tdy, tdx)
        descs.append(wd)
    sp = L.stream_ptr()
    L.check(lib.myolo_conv_wgrad(C.byref(descs[0]), sp))
    torch.cuda.synchronize()
    xr = xs[0].permute(0, 3, 1, 2).float().requires_grad_(False)
    wref = torch.zeros(cout, cin, k, k, device=dev, requires_grad=True)
    y = F.conv2d(xr, wref, None, s, pad, d)
    y.backward(dys[0].permute(0, 3, 1, 2).float())
    err = ((dw - wref.grad).norm() / wref.grad.norm()).item()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        spc = L.stream_ptr()
        for i in range(iters):
            L.check(lib.myolo_conv_wgrad(C.byref(descs[i % nbuf]), spc))
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * iters), err


# variants: (label, {option: value}) -- A/B of the LDS-DMA loaders (round 4) against the register loaders of round 2 in one process
VARIANTS = [('dma', {}), ('reg', {'wgrad_tile_dma': 0})]
if 'sweep' in sys.argv[1:]:
    VARIANTS += [('dma mt3', {'wgrad_tile_min_tiles': 3}), ('dma mt2', {'wgrad_tile_min_tiles': 2}), ('dma mt4 wg192', {'wgrad_tile_min_tiles': 4, 'wgrad_tile_wg': 192}),
                 ('dma nst3', {'wgrad_tile_nst': 3}), ('dma nst4', {'wgrad_tile_nst': 4})]
if 'dbg' in sys.argv[1:]:            # where the time goes: the kernel without its LDS-DMA / without fragment reads + MFMAs / without result stores (err is meaningless there)
    VARIANTS = [('dma', {}), ('no dma', {'wgrad_tile_dbg': 1}), ('no mfma', {'wgrad_tile_dbg': 2}), ('no store', {'wgrad_tile_dbg': 4}), ('dma only', {'wgrad_tile_dbg': 6}),
                ('mfma only', {'wgrad_tile_dbg': 5}), ('empty', {'wgrad_tile_dbg': 7})]
if 'r1' in sys.argv[1:]:
    VARIANTS += [('r1 kern', {'wgrad_tile_off': 1})]
DEFAULTS = {'wgrad_tile_dbg': 0, 'wgrad_tile_dma': 1, 'wgrad_tile_min_tiles': 6, 'wgrad_tile_wg': 128, 'wgrad_tile_nst': 0, 'wgrad_tile_off': 0}
print(f'{"shape":34s} {"MB":>6s} {"GF":>6s} {"roof us":>8s} | ' + ' '.join(f'{v[0]:>13s}' for v in VARIANTS) + ' | best GB/s  frac   err per variant')
tot = [0.0] * len(VARIANTS)
for cin, cout, k, d, s, H, W in SHAPES:
    pad = d * (k // 2)
    Ho, Wo = (H + 2 * pad - d * (k - 1) - 1) // s + 1, (W + 2 * pad - d * (k - 1) - 1) // s + 1
    byt = B * (H * W * cin + Ho * Wo * cout) * 2
    fl = 2.0 * B * Ho * Wo * cin * cout * k * k
    roof = max(byt / 8e12, fl / 2.5e15) * 1e6
    nbuf = max(2, int(300e6 // byt) + 1)
    res = []
    for _, opts in VARIANTS:
        for o, v in {**DEFAULTS, **opts}.items():
            lib.myolo_set_option(o.encode(), v)
        res.append(run(cin, cout, k, d, s, H, W, nbuf))
    for o, v in DEFAULTS.items():
        lib.myolo_set_option(o.encode(), v)
    us = min(r[0] for r in res)
    for i, r in enumerate(res):
        tot[i] += r[0]
    print(f'{cin:4d}->{cout:4d} k{k} d{d} s{s} {H:4d}x{W:4d}     {byt / 1e6:6.1f} {fl / 1e9:6.2f} {roof:8.1f} | ' + ' '.join(f'{r[0]:13.1f}' for r in res) +
          f' | {byt / us / 1e3:5.0f} {roof / us:5.2f}  ' + ' '.join(f'{r[1]:.1e}' for r in res), flush=True)
print(f'{"sum us":58s} | ' + ' '.join(f'{t:13.1f}' for t in tot))
