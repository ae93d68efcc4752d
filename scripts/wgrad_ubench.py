"""micro-benchmark of myolo_conv_wgrad through the raw C ABI (B=16, fp16): hipGraph-timed launches over rotating buffers, the LDS-tile
kernel (conv_wgrad_tile.hip) vs the round-1 kernels (`wgrad_tile_off`), numerics vs autograd (fp32)."""
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
if 'focus' in sys.argv[1:]:          # the Focus layer's 3x3 (compact kernels of conv_wgrad.hip; MYOLO_WGRAD_NO_SMALL_HALO=1: the per-tap-load one)
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


print(f'{"shape":40s} {"MB":>6s} {"GF":>6s} {"roof us":>8s} | {"tile":>8s} {"r1 kern":>8s} | GB/s  TF/s  frac   err(tile) err(r1)')
for cin, cout, k, d, s, H, W in SHAPES:
    pad = d * (k // 2)
    Ho, Wo = (H + 2 * pad - d * (k - 1) - 1) // s + 1, (W + 2 * pad - d * (k - 1) - 1) // s + 1
    byt = B * (H * W * cin + Ho * Wo * cout) * 2
    fl = 2.0 * B * Ho * Wo * cin * cout * k * k
    roof = max(byt / 8e12, fl / 2.5e15) * 1e6
    nbuf = max(2, int(300e6 // byt) + 1)
    res = []
    for off in (0, 1):
        lib.myolo_set_option(b'wgrad_tile_off', off)
        res.append(run(cin, cout, k, d, s, H, W, nbuf))
    lib.myolo_set_option(b'wgrad_tile_off', 0)
    us = res[0][0]
    print(f'{cin:4d}->{cout:4d} k{k} d{d} s{s} {H:4d}x{W:4d}           {byt / 1e6:6.1f} {fl / 1e9:6.2f} {roof:8.1f} | {res[0][0]:8.1f} {res[1][0]:8.1f} | '
          f'{byt / us / 1e3:5.0f} {fl / us / 1e6:5.0f} {roof / us:5.2f}  {res[0][1]:.1e} {res[1][1]:.1e}', flush=True)
