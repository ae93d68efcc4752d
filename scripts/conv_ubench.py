"""micro-benchmark of myolo_conv through the raw C ABI on the real layer shapes (B=16, fp16): per-launch time with HIP events over
rotating buffers (working set > the 256 MB Infinity Cache), default dispatch vs `halo_off` / `stream_off`, plus a numerics check
against ATen's conv2d (fp32 accumulate) on the first buffer.  usage: python scripts/conv_ubench.py [quick]"""
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
# (cin, cout, k, dil, H, W, stats)
SHAPES = [
    (64, 64, 3, 1, 64, 128), (128, 128, 3, 1, 32, 64), (32, 32, 3, 1, 128, 256), (256, 128, 3, 1, 64, 128),
    (64, 64, 3, 2, 64, 128), (64, 64, 3, 3, 64, 128), (16, 32, 3, 1, 256, 512), (256, 256, 3, 1, 16, 32),
    (64, 64, 1, 1, 64, 128), (128, 128, 1, 1, 64, 128), (256, 128, 1, 1, 64, 128), (384, 64, 1, 1, 64, 128), (64, 384, 1, 1, 64, 128),
    (64, 32, 1, 1, 128, 256), (256, 128, 1, 1, 32, 64), (512, 256, 1, 1, 16, 32),
]
if 'quick' in sys.argv[1:]:
    SHAPES = SHAPES[:4]


def tdesc(t):
    n, h, w, c = t.shape
    return L.Tensor(t.data_ptr(), n, h, w, c, h * w * c, w * c, c, L.F16, 0)


def run(cin, cout, k, d, H, W, stats, nbuf, iters=20):
    torch.manual_seed(0)
    xs = [(torch.randn(B, H, W, cin, device=dev) * 0.5).half() for _ in range(nbuf)]
    ys = [torch.zeros(B, H, W, cout, device=dev, dtype=torch.float16) for _ in range(nbuf)]
    w = (torch.randn(cout, cin, k, k, device=dev) * (1.0 / (cin * k * k) ** 0.5))
    cin_pad, cout_pad = E.rup(cin, 32), E.rup(cout, 32)
    wp = torch.zeros(cout_pad, k * k, cin_pad, device=dev, dtype=torch.float16)
    L.check(lib.myolo_pack_weight(L.ptr(w), L.F32, cout, cin, k, k, L.ptr(wp), L.F16, cout_pad, cin_pad, 0, None, L.stream_ptr()))
    st = torch.zeros(L.STAT_COPIES * 2 * cout, device=dev) if stats else None
    descs = []
    for x, y in zip(xs, ys):
        dd = L.ConvDesc()
        dd.x, dd.y, dd.w = tdesc(x), tdesc(y), wp.data_ptr()
        dd.cin_pad, dd.cout_pad, dd.wtaps, dd.ntaps, dd.stride, dd.up_shift = cin_pad, cout_pad, k * k, k * k, 1, 0
        E.fill_taps(dd, *E.taps_fwd(k, d, d * (k // 2)))
        dd.res = E.null_tensor()
        dd.act = L.ACT_NONE
        dd.stats = st.data_ptr() if stats else None
        descs.append(dd)
    sp = L.stream_ptr()
    for dd in descs[:2]:
        L.check(lib.myolo_conv(C.byref(dd), sp))
    torch.cuda.synchronize()
    # the launches are captured into one hipGraph: a ctypes + hipLaunchKernel call costs ~12 us of host time, more than most of these
    # kernels run
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
    us = e0.elapsed_time(e1) * 1e3 / (3 * iters)
    ref = F.conv2d(xs[0].permute(0, 3, 1, 2).float(), w.half().float(), None, 1, d * (k // 2), d).permute(0, 2, 3, 1)
    err = ((ys[0].float() - ref).norm() / ref.norm()).item()
    if stats:
        st.zero_()
        L.check(lib.myolo_conv(C.byref(descs[0]), sp))
        s = st.view(L.STAT_COPIES, 2, cout).sum(0)
        serr = ((s[0] - ref.sum((0, 1, 2))).abs().max() / ref.sum((0, 1, 2)).abs().max().clamp_min(1e-6)).item()
        qerr = ((s[1] - (ref * ref).sum((0, 1, 2))).abs().max() / (ref * ref).sum((0, 1, 2)).abs().max()).item()
        err = max(err, serr, qerr)
    return us, err


print(f'{"shape":34s} {"MB":>6s} {"GF":>6s} {"roof us":>8s} | {"default":>9s} {"halo_off":>9s} {"both_off":>9s} | GB/s  TF/s  frac   err')
for cin, cout, k, d, H, W in SHAPES:
    byt = B * H * W * (cin + cout) * 2 + cout * cin * k * k * 2
    fl = 2.0 * B * H * W * cin * cout * k * k
    roof = max(byt / 8e12, fl / 2.5e15) * 1e6
    nbuf = max(2, int(300e6 // byt) + 1)
    res = []
    for opt in ((0, 0), (1, 0), (1, 1)):
        lib.myolo_set_option(b'halo_off', opt[0]); lib.myolo_set_option(b'stream_off', opt[1])
        res.append(run(cin, cout, k, d, H, W, True, nbuf))
    lib.myolo_set_option(b'halo_off', 0); lib.myolo_set_option(b'stream_off', 0)
    us = res[0][0]
    extra = ''
    if k > 1 and len(sys.argv) > 1 and 'dbg' in sys.argv[1:]:
        for name, bits in (('nostore', 1), ('noload', 2), ('nomma', 4), ('nopanel', 8), ('only_panel', 7), ('nothing', 15)):
            lib.myolo_set_option(b'halo_dbg', bits)
            extra += f' {name}={run(cin, cout, k, d, H, W, True, nbuf)[0]:.1f}'
        lib.myolo_set_option(b'halo_dbg', 0)
    if k == 1 and 'sdbg' in sys.argv[1:]:
        for name, opts in (('nostore', ((b'stream_dbg', 1),)), ('noload', ((b'stream_dbg', 2),)), ('neither', ((b'stream_dbg', 3),)),
                           ('percu1', ((b'stream_per_cu', 1),))):
            for o, v in opts:
                lib.myolo_set_option(o, v)
            extra += f' {name}={run(cin, cout, k, d, H, W, True, nbuf)[0]:.1f}'
            extra += f'/{run(cin, cout, k, d, H, W, False, nbuf)[0]:.1f}'
            lib.myolo_set_option(b'stream_dbg', 0); lib.myolo_set_option(b'stream_per_cu', 0)
        extra += f' nostats={run(cin, cout, k, d, H, W, False, nbuf)[0]:.1f}'
    print(f'{cin:4d}->{cout:4d} k{k} d{d} {H:4d}x{W:4d}        {byt / 1e6:6.1f} {fl / 1e9:6.2f} {roof:8.1f} | {res[0][0]:9.1f} {res[1][0]:9.1f} {res[2][0]:9.1f} | '
          f'{byt / us / 1e3:5.0f} {fl / us / 1e6:5.0f} {roof / us:5.2f}  {max(r[1] for r in res):.1e}{extra}', flush=True)
