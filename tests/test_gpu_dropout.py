"""-m gpu: train-mode nn.Dropout(0.1) of the Base / BiSe heads (reference models/yolo.py:65,140) -- csrc/dropout.hip.

RNG streams cannot match the reference's (SURVEY 8c dropout caveat), so parity is pinned in two parts: (1) the keep-mask the
kernel drew is read back and REPLAYED in the oracle, which makes forward and backward of the whole head comparable at the usual
tolerances (scale 1/(1-p), the backward uses the forward's mask); (2) the mask itself is checked statistically (keep rate within
4 sigma, fresh mask on every forward)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import model_ref
from tests.gpu_util import TOL, check
from tests.test_gpu_ops import _oracle, _randomize

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
P = 0.1


def _dropout_ops(mod):
    from multiyolov5_amd import engine as E
    ops = []
    for h in mod.__dict__['_plans'].values():
        ops += [op for op in h.plan.ops if isinstance(op, E.DropoutOp)]
    return ops


def _replay(mask_nchw, p):
    return lambda x: x * mask_nchw.to(x.dtype) / (1.0 - p)


HEADS = {
    'base': (lambda Y: Y.SegMaskBase(19, 1, 64, False, [64]), [(2, 64, 16, 32)],
             lambda c, p, xs: model_ref.seg_base(c, p, xs if isinstance(xs, list) else [xs], 1)),
    'bise': (lambda Y: Y.SegMaskBiSe(19, 1, 64, False, [64, 128, 256]), [(2, 64, 16, 32), (2, 128, 8, 16), (2, 256, 4, 8)],
             lambda c, p, xs: model_ref.seg_bise(c, p, xs)),
}


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16], ids=['f32', 'f16'])
@pytest.mark.parametrize('head', list(HEADS))
def test_head_train_mode_with_dropout_vs_oracle_replaying_the_mask(head, dtype):
    from multiyolov5_amd.models import yolo as Y
    from multiyolov5_amd.utils.torch_utils import initialize_weights
    ctor, shapes, fn = HEADS[head]
    torch.manual_seed(1)
    mod = ctor(Y)
    initialize_weights(mod)
    _randomize(mod)
    assert any(isinstance(m, torch.nn.Dropout) and m.p == P for m in mod.modules())       # the shipped default
    mod = mod.to(DEV).train()
    g = torch.Generator().manual_seed(3)
    xs_cpu = [torch.randn(s, generator=g) for s in shapes]
    xs = [x.to(DEV, dtype).requires_grad_() for x in xs_cpu]
    out = mod(xs)
    outs = out if isinstance(out, list) else [out]
    ops = _dropout_ops(mod)
    assert len(ops) == 1
    s = ops[0].src
    mask = ops[0].mask.view(s.n, s.h, s.w, s.c).permute(0, 3, 1, 2).cpu().clone()
    # (2) statistics of the drawn mask
    n = mask.numel()
    keep = float(mask.float().mean())
    assert set(np.unique(mask.numpy()).tolist()) <= {0, 1}
    assert abs(keep - (1 - P)) < 4 * (P * (1 - P) / n) ** 0.5, (keep, n)
    # (1) replay in the oracle
    sd = {'m.' + k: (v.detach().cpu().float().clone() if v.dtype.is_floating_point else v.detach().cpu().clone())
          for k, v in mod.state_dict().items()}
    params = {k: v.requires_grad_() for k, v in sd.items() if v.dtype.is_floating_point and 'running' not in k}
    ctx = model_ref.Ctx(sd, True, dropout_p=P)
    ctx.dropout_fn = _replay(mask, P)
    xin = [x.to(dtype).float().clone().requires_grad_() for x in xs_cpu]
    ref = fn(ctx, 'm', xin)
    refs = ref if isinstance(ref, list) else [ref]
    tol = TOL[dtype]
    bad = []
    rs = [torch.randn(r.shape, generator=g) * 0.1 for r in refs]
    for j, (o, r) in enumerate(zip(outs, refs)):
        check(f'dropout/{head}/out{j}', o, r, tol, collect=bad)
    sum((r * c).sum() for r, c in zip(refs, rs)).backward()
    sum((o.float() * c.to(DEV)).sum() for o, c in zip(outs, rs)).backward()
    for i, x in enumerate(xs):
        check(f'dropout/{head}/dx{i}', x.grad, xin[i].grad, tol * 3, collect=bad)
    for k, p in mod.named_parameters():
        # (fp16: x6 -- the final runs of round 5 measured 7.3e-2 on dm.1.cv1.bn.bias of the Base head against the former 8e-2: a BatchNorm bias
        #  gradient of a 16-channel layer a few fp16 layers deep; the fp32 case keeps x4 = 8e-4)
        check(f'dropout/{head}/d{k}', p.grad, params['m.' + k].grad, tol * (6 if dtype == torch.float16 else 4), collect=bad)
    assert not bad, '\n'.join(bad)
    # a new forward draws a new mask (counter-based RNG advanced by its own launch: also true for a graph replay)
    mod(xs)
    mask2 = ops[0].mask.view(s.n, s.h, s.w, s.c).permute(0, 3, 1, 2).cpu()
    differ = float((mask2 != mask).float().mean())
    assert abs(differ - 2 * P * (1 - P)) < 6 * (2 * P * (1 - P) / n) ** 0.5 + 0.01, differ
    # eval mode: identity (no DropoutOp in the plan)
    mod.eval()
    with torch.no_grad():
        mod([x.detach() for x in xs])
    assert len(_dropout_ops(mod)) == 1


def test_dropout_kernel_scale_mask_reuse_and_graph_replay():
    """C ABI: kept elements are scaled by exactly 1/(1-p), dropped are 0, the backward applies the forward's mask (+accumulate),
    and a captured hipGraph draws a fresh mask on every replay"""
    from multiyolov5_amd import _lib as L
    lib = L.lib()
    for dt in (torch.float16, torch.float32):
        n, h, w, c = 2, 24, 40, 64
        x = (torch.rand(n, h, w, c, device=DEV) + 0.5).to(dt)
        out = torch.zeros_like(x)
        mask = torch.zeros(x.numel(), dtype=torch.uint8, device=DEV)
        counter = torch.tensor([12345], dtype=torch.int64, device=DEV)
        d = lambda t: L.Tensor(L.ptr(t), n, h, w, c, h * w * c, w * c, c, L.DT[dt], 0)
        xd, od = d(x), d(out)
        L.check(lib.myolo_dropout_fwd(C.byref(xd), C.byref(od), L.ptr(mask), C.c_float(P), L.ptr(counter), L.stream_ptr()))
        torch.cuda.synchronize()
        assert int(counter.item()) == 12346
        mk = mask.view(n, h, w, c).bool()
        scale = 1.0 / (1.0 - P)
        exp = torch.where(mk, x.float() * scale, torch.zeros((), device=DEV)).to(dt)
        assert torch.equal(out, exp)
        keep = float(mk.float().mean())
        assert abs(keep - (1 - P)) < 4 * (P * (1 - P) / x.numel()) ** 0.5
        g = torch.randn(n, h, w, c, device=DEV).to(dt)
        base = torch.randn(n, h, w, c, device=DEV).to(dt)
        for acc in (0, 1):
            gx = base.clone()
            gd, gxd = d(g), d(gx)
            L.check(lib.myolo_dropout_bwd(C.byref(gd), L.ptr(mask), C.byref(gxd), C.c_float(P), acc, L.stream_ptr()))
            ref = torch.where(mk, g.float() * scale, torch.zeros((), device=DEV)) + (base.float() if acc else 0)
            check(f'dropout/bwd/acc{acc}/{dt}', gx, ref, 1e-3 if dt == torch.float16 else 1e-6)
        # graph replay: fresh mask each time
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                L.check(lib.myolo_dropout_fwd(C.byref(xd), C.byref(od), L.ptr(mask), C.c_float(P), L.ptr(counter),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            gr.replay(); torch.cuda.synchronize(); m1 = mask.clone()
            gr.replay(); torch.cuda.synchronize(); m2 = mask.clone()
        torch.cuda.current_stream().wait_stream(s)
        assert float((m1 != m2).float().mean()) > 0.1
