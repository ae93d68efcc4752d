"""-m gpu: conv_wgrad_tile.hip through the raw C ABI (`myolo_conv_wgrad`, myolo.h) against torch's fp32 autograd weight gradient on the
CPU over the SAME fp16-rounded x / dy: the LDS-DMA loaders (3- and 4-stage ring, counted vmcnt), the register loaders they replaced,
the non-temporal DMA policy of single-block 1x1 layers, explicit split counts -- real layer shapes of SURVEY Appendix A plus
ragged maps (tile rows / columns past the image, halo outside the image on every side), stride 2, dilation, several gradient blocks,
channel counts below a block.  fp32 accumulation of fp16 products: tolerance 1e-3 relative L2 (measured 3e-7 .. 1.3e-6)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import check

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
DEFAULTS = {'wgrad_tile_dma': 1, 'wgrad_tile_nst': 0, 'wgrad_tile_min_tiles': 6, 'wgrad_tile_wg': 128,
            'wgrad_tile_dbg': 0, 'wgrad_tile_off': 0}


def _tdesc(L, t):
    n, h, w, c = t.shape
    sn, sh, sw, _ = t.stride()
    return L.Tensor(t.data_ptr(), n, h, w, c, sn, sh, sw, L.F16, 0)


def _run(cin, cout, k, d, s, B, H, W, opts, ksplit=0, accumulate_onto=False, seed=0):
    from multiyolov5_amd import _lib as L, engine as E
    lib = L.lib()
    g = torch.Generator().manual_seed(seed)
    pad = d * (k // 2)
    Ho, Wo = (H + 2 * pad - d * (k - 1) - 1) // s + 1, (W + 2 * pad - d * (k - 1) - 1) // s + 1
    x = (torch.randn(B, H, W, cin, generator=g) * 0.5).half()
    dy = (torch.randn(B, Ho, Wo, cout, generator=g) * 0.1).half()
    w0 = torch.zeros(cout, cin, k, k, requires_grad=True)
    F.conv2d(x.float().permute(0, 3, 1, 2), w0, None, s, pad, d).backward(dy.float().permute(0, 3, 1, 2))
    ref = w0.grad
    xd, dyd = x.to(DEV), dy.to(DEV)
    base = (torch.randn(cout, cin, k, k, generator=g) * 0.05) if accumulate_onto else torch.zeros(cout, cin, k, k)
    dw = base.to(DEV).clone()                              # the entry point ADDS its result (gradient accumulation, train.py:371/392)
    ws = torch.empty(16 << 20 >> 2, dtype=torch.float32, device=DEV)
    wd = L.WgradDesc()
    wd.x, wd.dy = _tdesc(L, xd), _tdesc(L, dyd)
    wd.dw, wd.db = dw.data_ptr(), None
    wd.ntaps, wd.stride, wd.up_shift, wd.ksplit, wd.cout, wd.cin = k * k, s, 0, ksplit, cout, cin
    wd.ws, wd.ws_bytes = ws.data_ptr(), ws.numel() * 4
    tdy, tdx, _ = E.taps_fwd(k, d, pad)
    E.fill_taps(wd, tdy, tdx)
    try:
        for o, v in {**DEFAULTS, **opts}.items():
            lib.myolo_set_option(o.encode(), v)
        L.check(lib.myolo_conv_wgrad(C.byref(wd), L.stream_ptr()))
        torch.cuda.synchronize()
    finally:
        for o, v in DEFAULTS.items():
            lib.myolo_set_option(o.encode(), v)
    tag = f'wgrad_tile/{cin}->{cout} k{k}d{d}s{s} {B}x{H}x{W} ' + ','.join(f'{a[11:]}={b}' for a, b in opts.items()) + (f' ks{ksplit}' if ksplit else '')
    check(tag, dw.cpu() - base, ref, 1e-3)


VARIANTS = {
    'dma': {},
    'dma_nst3': {'wgrad_tile_nst': 3},
    'dma_nst4': {'wgrad_tile_nst': 4},
    'dma_mt2': {'wgrad_tile_min_tiles': 2},
    'reg': {'wgrad_tile_dma': 0},              # the fallback when no LDS-DMA ring fits (one layer of the bs-16 step)
}
SHAPES = [
    # cin, cout, k, dil, stride, B, H, W
    (64, 64, 3, 1, 1, 2, 64, 128),           # 4.m.*.cv2: one 64 x 64 block, 9 taps
    (128, 128, 3, 1, 1, 2, 32, 64),          # 6.m.*.cv2: 2 x 2 blocks
    (256, 256, 3, 1, 1, 4, 16, 32),          # 9.m.0.cv2: 4 x 4 blocks, short maps
    (128, 128, 1, 1, 1, 2, 64, 128),         # 1x1, one 128 x 128 block
    (512, 256, 1, 1, 1, 2, 16, 32),          # 1x1, 2 x 4 blocks
    (384, 64, 1, 1, 1, 1, 64, 128),          # PSP head entry: 3 ci blocks, 64-wide co
    (64, 128, 3, 1, 2, 2, 64, 128),          # stride 2 (3.conv)
    (64, 64, 3, 2, 1, 1, 32, 64),            # dilation 2
    (32, 32, 3, 1, 1, 1, 37, 53),            # ragged map: rows and columns past the image, halo outside on every side
    (96, 40, 1, 1, 1, 1, 19, 70),            # channel counts below / between the block widths, ragged
    (48, 24, 3, 1, 2, 1, 31, 45),            # stride 2 on odd sizes, narrow blocks
]


@pytest.mark.parametrize('variant', list(VARIANTS))
@pytest.mark.parametrize('shape', SHAPES, ids=[f'{s[0]}-{s[1]}k{s[2]}d{s[3]}s{s[4]}_{s[5]}x{s[6]}x{s[7]}' for s in SHAPES])
def test_wgrad_tile_variants_match_autograd(shape, variant):
    _run(*shape, opts=VARIANTS[variant])


@pytest.mark.parametrize('variant', ['dma', 'reg'])
def test_wgrad_tile_explicit_splits_and_accumulation(variant):
    """caller-chosen split counts (1 = no split-K at all, a prime, more splits than workgroups aimed at) and accumulation onto an
    existing gradient"""
    for ks in (1, 7, 24):
        _run(128, 128, 3, 1, 1, 2, 32, 64, VARIANTS[variant], ksplit=ks)
        _run(256, 128, 1, 1, 1, 2, 32, 64, VARIANTS[variant], ksplit=ks)
    _run(64, 64, 3, 1, 1, 1, 32, 64, VARIANTS[variant], accumulate_onto=True)
    _run(128, 64, 1, 1, 1, 1, 32, 64, VARIANTS[variant], accumulate_onto=True)
