"""CPU: the dense stride-2 dgrad form of round 5 (engine.ConvOp._build_bwd `dense2`, csrc/pack.hip pack_s2_dgrad_body) restated in torch:
the two stride-1 convolutions over dy with the zero-slotted weight tensors W'[py] reproduce autograd's input gradient of
nn.Conv2d(k=3, s=2, p=1) (reference models/common.py:38) exactly (fp32: 1e-5).  `pack_s2_ref` is the layout contract the device pack kernel is
compared with in tests/test_gpu_ops.py."""
import pytest
import torch
import torch.nn.functional as F


def pack_s2_ref(w, py):
    """w: OIHW [cout, cin, 3, 3] -> W'[2*cin rows (px, ci)][T slots][cout]   (pack.hip pack_s2_dgrad_body)"""
    cout, cin = w.shape[:2]
    T = 4 if py else 2
    out = torch.zeros(2 * cin, T, cout, dtype=w.dtype)
    for t in range(T):
        dyi, dxi = ((t >> 1), (t & 1)) if py else (0, t)
        ky = (0 if dyi else 2) if py else 1
        for px in range(2):
            kx = (0 if px else -1) if dxi else (2 if px else 1)
            if kx < 0:
                continue
            out[px * cin:(px + 1) * cin, t, :] = w[:, :, ky, kx].t()
    return out


@pytest.mark.parametrize('shape', [(2, 8, 16, 12, 20), (1, 16, 8, 6, 10)], ids=['8->16', '16->8'])
def test_two_stride1_convolutions_over_dy_give_the_stride2_input_gradient(shape):
    torch.manual_seed(0)
    n, cin, cout, H, W = shape
    x = torch.randn(n, cin, H, W, requires_grad=True)
    w = torch.randn(cout, cin, 3, 3)
    y = F.conv2d(x, w, None, 2, 1)
    dy = torch.randn_like(y)
    y.backward(dy)
    gx_ref = x.grad.permute(0, 2, 3, 1)                              # NHWC
    dyn = dy.permute(0, 2, 3, 1)                                     # [n, Ho, Wo, cout]
    Ho, Wo = dyn.shape[1:3]
    gx = torch.zeros(n, H, W, cin)
    for py in range(2):
        wp = pack_s2_ref(w, py)
        T = wp.shape[1]
        taps = [(0, 0), (0, 1), (1, 0), (1, 1)][:T]
        dpad = F.pad(dyn, (0, 0, 0, 1, 0, 1))                        # taps past the map read zeros (myolo.h:87)
        acc = torch.zeros(n, Ho, Wo, 2 * cin)
        for t, (ddy, ddx) in enumerate(taps):
            acc += dpad[:, ddy:ddy + Ho, ddx:ddx + Wo, :] @ wp[:, t, :].t()
        # the launch's output view: pixel (a, b) of the virtual tensor = gradient pixels (2a+py, 2b), (2a+py, 2b+1)
        gx[:, py::2, :, :] = acc.reshape(n, Ho, Wo, 2, cin).reshape(n, Ho, 2 * Wo, cin)
    assert float((gx - gx_ref).abs().max()) <= 1e-5 * float(gx_ref.abs().max())
