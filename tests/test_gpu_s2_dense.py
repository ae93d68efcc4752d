"""-m gpu: the dense stride-2 dgrad of round 5.  (1) csrc/pack.hip `transpose = 2 + py` through `myolo_pack_weights_mt` is bit-identical to
tests/test_s2_dense_cpu.pack_s2_ref (the layout whose arithmetic the CPU test proves against autograd); (2) a Conv(k=3, s=2) layer
(reference models/common.py:34-46) at real shapes, fp16 training step: the input gradient against torch fp32 on the same fp16-rounded
operands (5e-3), and the launch trace shows two stride-1 conv launches instead of the round-2 parity kernel."""
import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import check
from tests.test_s2_dense_cpu import pack_s2_ref

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('shape', [(64, 32), (128, 64), (256, 256), (96, 40)], ids=lambda s: f'{s[1]}->{s[0]}')
@pytest.mark.parametrize('src_dtype', [torch.float32, torch.float16], ids=['f32', 'f16'])
def test_s2_dgrad_pack_matches_the_reference_layout(shape, src_dtype):
    from multiyolov5_amd import _lib as L
    lib = L.lib()
    cout, cin = shape
    g = torch.Generator().manual_seed(1)
    w = torch.randn(cout, cin, 3, 3, generator=g).to(src_dtype)
    wd = w.to(DEV)
    rows_pad, cols_pad = (2 * cin + 63) // 64 * 64, (cout + 31) // 32 * 32
    for py in range(2):
        T = 4 if py else 2
        dst = torch.zeros(rows_pad, T, cols_pad, dtype=torch.float16, device=DEV)
        jobs = torch.tensor([[wd.data_ptr(), dst.data_ptr(), cout, cin, T, rows_pad, cols_pad, 2 + py, L.DT[src_dtype], L.F16, 0, 0]],
                            dtype=torch.int64, device=DEV)
        n = rows_pad * T * cols_pad
        chunks = torch.tensor([[0, s0] for s0 in range(0, n, 8192)], dtype=torch.int32, device=DEV)
        L.check(lib.myolo_pack_weights_mt(L.ptr(jobs), L.ptr(chunks), chunks.shape[0], -9, L.stream_ptr()))
        torch.cuda.synchronize()
        ref = torch.zeros(rows_pad, T, cols_pad, dtype=torch.float16)
        ref[:2 * cin, :, :cout] = pack_s2_ref(w.float(), py).half()
        assert torch.equal(dst.cpu(), ref), (py, float((dst.cpu().float() - ref.float()).abs().max()))


@pytest.mark.parametrize('layer', [(32, 64, 2, 128, 256), (64, 128, 2, 64, 128), (128, 128, 4, 32, 64), (256, 256, 16, 16, 32)],
                         ids=['1.conv', '3.conv', '18.conv', '21.conv_b16'])
def test_stride2_conv_input_gradient_through_the_dense_form(layer):
    from multiyolov5_amd import _lib as L
    from multiyolov5_amd.models.common import Conv
    from tests.test_gpu_ops import _randomize
    cin, cout, B, H, W = layer
    torch.manual_seed(3)
    mod = Conv(cin, cout, 3, 2)
    _randomize(mod)
    with torch.no_grad():
        mod.conv.weight.copy_(mod.conv.weight.half().float())
    mod.bn.eps, mod.bn.momentum = 1e-3, 0.03
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, cin, H, W, generator=g).half()
    xr = x.float().requires_grad_()
    wr = mod.conv.weight.detach().clone().requires_grad_()
    yr = F.silu(F.batch_norm(F.conv2d(xr, wr, None, 2, 1), None, None, mod.bn.weight.detach(), mod.bn.bias.detach(), True, 0.03, 1e-3))
    go = torch.randn(yr.shape, generator=g)
    yr.backward(go)
    mod = mod.to(DEV).train()
    xg = x.to(DEV).requires_grad_()
    y = mod(xg)
    L.lib().myolo_trace_start(1)
    y.backward(go.to(DEV, torch.float16))
    torch.cuda.synchronize()
    sites = L.launch_trace()
    L.lib().myolo_trace_start(0)
    assert not any('S2 = 1' in s for s in sites), sorted(sites)          # not the round-2 parity kernel ...
    assert sum(n for s, n in sites.items() if s.startswith('int mid::launch') or s.startswith('int midx::launch')) >= 2, sorted(sites)
    bad = []
    check(f's2dense/{cin}->{cout}/y', y, yr, 5e-3, collect=bad)
    check(f's2dense/{cin}->{cout}/dx', xg.grad, xr.grad, 5e-3, collect=bad)
    check(f's2dense/{cin}->{cout}/dw', mod.conv.weight.grad, wr.grad, 5e-3, collect=bad)
    assert not bad, '\n'.join(bad)
