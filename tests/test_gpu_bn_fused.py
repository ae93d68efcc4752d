"""-m gpu: the one-launch BatchNorm backward of round 6 (`myolo_bn_act_bwd_fused`, csrc/bn_act.hip: partial sums -> device-wide barrier
inside the launch -> dx from the registers that still hold gout and y) through the raw C ABI against
  * fp32 first principles on the CPU (autograd of nn.BatchNorm2d (train mode) + nn.SiLU / Sigmoid over the same fp16-rounded operands:
    reference models/common.py:42-43), and
  * the two launches it replaces (`myolo_bn_act_bwd_reduce_split` + `myolo_bn_act_bwd_apply_split`) on the same inputs,
for every register-tile size (NP = 2 / 4 / 8 pixels per thread), ragged pixel counts, channel slices (64 / 128 / 256 / 512 channels), a
split parameter set (C3's merged cv1 | cv2), the shortcut gradient pass-through (overwrite and accumulate), fp32, many launches sharing ONE
barrier block (it resets itself), and a tensor that does not fit the resident grid (must fall back to the two launches).  The barrier's
timeout word must stay 0."""
import ctypes as C

import pytest
import torch

from tests.gpu_util import check

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _td(L, t, dt):
    n, h, w, c = t.shape
    sn, sh, sw, _ = t.stride()
    return L.Tensor(t.data_ptr(), n, h, w, c, sn, sh, sw, dt, 0)


def _ref(gout, y, gamma, beta, mean, invstd, act):
    xhat = (y.double() - mean.double()) * invstd.double()
    z = (xhat * gamma.double() + beta.double())
    s = torch.sigmoid(z)
    dact = s * (1 + z * (1 - s)) if act == 1 else s * (1 - s)
    dz = gout.double() * dact
    M = y.shape[0] * y.shape[1] * y.shape[2]
    d0, d1 = dz.sum((0, 1, 2)), (dz * xhat).sum((0, 1, 2))
    dx = (gamma.double() * invstd.double()) * (dz - d0 / M - xhat * d1 / M)
    return dx.float(), d0.float(), d1.float()


def _case(n, h, w, c, dtype=torch.float16, act=1, split=0, gres=0, seed=0, bar=None, expect_fused=True):
    from multiyolov5_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(seed)
    dt = L.F16 if dtype == torch.float16 else L.F32
    y = (torch.randn(n, h, w, c, generator=g) * 0.7 + torch.randn(c, generator=g) * 0.5).to(dtype)
    gout = (torch.randn(n, h, w, c, generator=g) * 0.1).to(dtype)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2
    yf = y.float().reshape(-1, c)
    mean = yf.mean(0)
    invstd = 1.0 / torch.sqrt(yf.var(0, unbiased=False) + 1e-3)
    r0 = (torch.randn(n, h, w, c, generator=g) * 0.1).to(dtype)
    dx_ref, d0_ref, d1_ref = _ref(gout.float(), y.float(), gamma, beta, mean, invstd, act)
    assert bool(lib.myolo_bn_act_bwd_fused_ok(dt, n * h * w, c)) == expect_fused

    def run(fused):
        yd, gd = y.to(DEV), gout.to(DEV)
        dy = torch.full((n, h, w, c), float('nan'), device=DEV, dtype=dtype)
        rd = r0.to(DEV).clone()
        saved = torch.cat([mean, invstd]).to(DEV)
        dsum = torch.zeros(L.STAT_COPIES * 2 * c, device=DEV)
        cs = split if split else c
        ga, be = gamma[:cs].contiguous().to(DEV), beta[:cs].contiguous().to(DEV)
        ga2, be2 = gamma[cs:].contiguous().to(DEV), beta[cs:].contiguous().to(DEV)
        dga, dbe = torch.ones(cs, device=DEV), torch.ones(cs, device=DEV)             # (accumulated into: start from 1)
        dga2, dbe2 = torch.ones(max(c - cs, 1), device=DEV), torch.ones(max(c - cs, 1), device=DEV)
        sp = L.BnSplit()
        sp.c_split, sp.count_scale = cs, 1
        sp.gamma2, sp.beta2, sp.dgamma2, sp.dbeta2 = ga2.data_ptr(), be2.data_ptr(), dga2.data_ptr(), dbe2.data_ptr()
        spp = C.byref(sp) if split else None
        tg, ty, tdy, tr = _td(L, gd, dt), _td(L, yd, dt), _td(L, dy, dt), _td(L, rd, dt)
        none = L.Tensor(0, 0, 0, 0, 0, 0, 0, 0, dt, 0)
        trp = C.byref(tr) if gres else C.byref(none)
        st = L.stream_ptr()
        if fused:
            L.check(lib.myolo_bn_act_bwd_fused(C.byref(tg), C.byref(ty), L.ptr(saved), L.ptr(ga), L.ptr(be), act, L.ptr(dsum), L.ptr(dga),
                                               L.ptr(dbe), C.byref(tdy), trp, int(gres == 2), spp, L.ptr(bar), st), 'fused')
        else:
            L.check(lib.myolo_bn_act_bwd_reduce_split(C.byref(tg), C.byref(ty), L.ptr(saved), L.ptr(ga), L.ptr(be), act, L.ptr(dsum), spp, st))
            L.check(lib.myolo_bn_act_bwd_apply_split(C.byref(tg), C.byref(ty), L.ptr(saved), L.ptr(ga), L.ptr(be), act, L.ptr(dsum), L.ptr(dga),
                                                     L.ptr(dbe), C.byref(tdy), trp, int(gres == 2), spp, st))
        torch.cuda.synchronize()
        ds = dsum.view(L.STAT_COPIES, 2, c).sum(0).cpu()
        dg = torch.cat([dga.cpu(), dga2.cpu()[:c - cs]]) - 1
        db = torch.cat([dbe.cpu(), dbe2.cpu()[:c - cs]]) - 1
        return dy.float().cpu(), ds, dg, db, rd.float().cpu()

    bad = []
    tag = f'bn_fused/{n}x{h}x{w}x{c}/{"f16" if dtype == torch.float16 else "f32"}/act{act}' + (f'/split{split}' if split else '') + (f'/gres{gres}' if gres else '')
    tol = 3e-3 if dtype == torch.float16 else 2e-5
    dy, ds, dg, db, rd = run(True)
    assert int(bar[18 * 32]) == 0, tag + ': the grid barrier timed out'
    check(tag + '/dx', dy, dx_ref, tol, collect=bad)
    check(tag + '/dsum0', ds[0], d0_ref, 1e-3, collect=bad)
    check(tag + '/dsum1', ds[1], d1_ref, 1e-3, collect=bad)
    check(tag + '/dgamma', dg, d1_ref, 1e-3, collect=bad)
    check(tag + '/dbeta', db, d0_ref, 1e-3, collect=bad)
    if gres:
        check(tag + '/gres', rd, gout.float() + (r0.float() if gres == 2 else 0), tol, collect=bad)
    # against the two launches: same arithmetic, only the order of the partial sums differs
    dy2, ds2, dg2, db2, rd2 = run(False)
    check(tag + '/dx_vs_two_launches', dy, dy2, 2e-3 if dtype == torch.float16 else 2e-5, collect=bad)
    check(tag + '/dsum_vs_two_launches', ds, ds2, 1e-4, collect=bad)
    if gres:
        assert torch.equal(rd, rd2), tag + ': shortcut gradient differs from the two-launch form'
    assert not bad, '\n'.join(bad)


CASES = [
    # n, h, w, c                      resident grid of bn_fused_np (csrc/bn_act.hip)
    (2, 16, 32, 256),                 # 1024 pixels, 4 slices: NP 2
    (1, 31, 37, 64),                  # ragged: 1147 pixels
    (16, 16, 32, 128),                # 14.bn / m32-sized: 1 M elements
    (16, 16, 32, 512),                # 8 slices, 4 M elements: NP 8
    (16, 32, 64, 128),                # 4 M elements, two slices: NP 8, 256 workgroups
    (16, 64, 128, 64),                # 8 M elements: NP 8, 512 workgroups (the largest tensor that is fused)
    (16, 32, 64, 256),                # 8 M elements in four slices
    (3, 50, 70, 32),                  # 32 channels: G = 4 (64-pixel rows), ragged
    (2, 24, 40, 16),                  # 16 channels
]


@pytest.fixture(scope='module')
def bar():
    """one barrier block for the whole module; the resident-grid bound is lifted to the kernel's own limits (the library's default of 256
    workgroups is a scheduling choice of the training step -- the weight-gradient stream shares the chip -- not a correctness bound)"""
    from multiyolov5_amd import _lib as L
    L.lib().myolo_set_option(b'bn_fused_cap', 1 << 20)
    L.lib().myolo_set_option(b'bn_fused', 1)
    yield torch.zeros(19 * 32, dtype=torch.int32, device=DEV)
    L.lib().myolo_set_option(b'bn_fused_cap', 256)


@pytest.mark.parametrize('shape', CASES, ids=['x'.join(map(str, s)) for s in CASES])
def test_fused_bn_backward_matches_first_principles_and_the_two_launches(shape, bar):
    _case(*shape, bar=bar)


def test_fused_bn_backward_variants(bar):
    _case(16, 32, 64, 128, gres=1, bar=bar)                            # Bottleneck shortcut: gres = gout
    _case(16, 32, 64, 128, gres=2, seed=3, bar=bar)                    # ... accumulated
    _case(16, 32, 64, 256, split=128, seed=4, bar=bar)                 # C3's merged cv1 | cv2: two parameter sets
    _case(4, 32, 64, 128, act=2, seed=5, bar=bar)                      # Sigmoid (run-time activation switch)
    _case(2, 32, 64, 128, dtype=torch.float32, seed=6, bar=bar)        # fp32 parity mode
    _case(4, 32, 64, 64, dtype=torch.float32, gres=2, split=32, seed=7, bar=bar)


def test_tensors_beyond_the_resident_grid_take_the_two_launches(bar):
    from multiyolov5_amd import _lib as L
    _case(16, 64, 128, 128, bar=bar, expect_fused=False)                # 16 M elements: 1024 workgroups at NP 8
    tr0 = None
    L.lib().myolo_trace_start(1)
    _case(8, 64, 128, 128, bar=bar, seed=2)                            # 8 M elements in two slices: fused
    tr0 = L.launch_trace()
    L.lib().myolo_trace_start(0)
    assert any('bn_act_bwd_fused' in k for k in tr0), tr0
    # the library's default bound (one workgroup per CU): the same tensor runs the two launches, a 4 M element one stays fused
    L.lib().myolo_set_option(b'bn_fused_cap', 256)
    try:
        _case(8, 64, 128, 128, bar=bar, seed=2, expect_fused=False)
        _case(16, 32, 64, 128, bar=bar, seed=3)
    finally:
        L.lib().myolo_set_option(b'bn_fused_cap', 1 << 20)


def test_many_launches_share_one_self_resetting_barrier_block(bar):
    """40 fused launches of different grid sizes back to back on one stream over one state block: counters return to zero, generations
    advance, nothing times out"""
    for i in range(20):
        _case(2, 16, 32, 256, seed=10 + i, bar=bar)
        _case(16, 32, 64, 128, seed=40 + i, bar=bar)
    st = bar.cpu()
    assert int(st[18 * 32]) == 0
    assert all(int(st[g * 32]) == 0 for g in range(8)) and int(st[16 * 32]) == 0
