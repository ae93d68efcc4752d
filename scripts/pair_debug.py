"""GPU debug of csrc/conv_pair.hip: isolate GEMM 1 (W2 = centre-tap identity) and GEMM 2 (W1 = identity) and print where the error sits."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from multiyolov5_amd import _lib as L, engine as E
DEV = 'cuda:0'
lib = L.lib()


def view(t, c):
    n, h, w, cc = t.shape
    sn, sh, sw, _ = t.stride()
    return L.Tensor(t.data_ptr(), n, h, w, c, sn, sh, sw, L.F16, 0)


def run(Cc, B, H, W, mode, th=0):
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(B, H, W, Cc, generator=g) * 0.7).half()
    w1 = (torch.randn(Cc, Cc, 1, 1, generator=g) * (1.0 / Cc ** 0.5)).half()
    w2 = (torch.randn(Cc, Cc, 3, 3, generator=g) * (1.0 / (Cc * 9) ** 0.5)).half()
    act1 = act2 = L.ACT_SILU
    if mode == 'gemm2':            # W1 = identity, no activation: t = x
        w1 = torch.eye(Cc).view(Cc, Cc, 1, 1).half(); act1 = L.ACT_NONE
    if mode == 'gemm1':            # W2 = centre identity, no activation: y = t
        w2 = torch.zeros(Cc, Cc, 3, 3).half(); w2[:, :, 1, 1] = torch.eye(Cc).half(); act2 = L.ACT_NONE
    xn = x.float().permute(0, 3, 1, 2)
    t = F.conv2d(xn, w1.float())
    t = (F.silu(t) if act1 else t).half().float()
    r = F.conv2d(t, w2.float(), None, 1, 1)
    ref = (F.silu(r) if act2 else r).permute(0, 2, 3, 1).contiguous()
    xb = x.to(DEV); yb = torch.zeros(B, H, W, Cc, dtype=torch.float16, device=DEV); tb = torch.zeros_like(yb)

    def pack(w, k):
        wp = torch.zeros(Cc, k * k, Cc, device=DEV, dtype=torch.float16)
        L.check(lib.myolo_pack_weight(L.ptr(w.float().to(DEV)), L.F32, Cc, Cc, k, k, L.ptr(wp), L.F16, Cc, Cc, 0, None, L.stream_ptr()))
        return wp
    wp1, wp2 = pack(w1, 1), pack(w2, 3)
    a, b = L.ConvDesc(), L.ConvDesc()
    a.x, a.y, a.w = view(xb, Cc), view(tb, Cc), wp1.data_ptr()
    a.cin_pad, a.cout_pad, a.wtaps, a.ntaps, a.stride, a.up_shift = Cc, Cc, 1, 1, 1, 0
    E.fill_taps(a, *E.taps_fwd(1, 1, 0)); a.act, a.res = act1, E.null_tensor()
    b.x, b.y, b.w = view(tb, Cc), view(yb, Cc), wp2.data_ptr()
    b.cin_pad, b.cout_pad, b.wtaps, b.ntaps, b.stride, b.up_shift = Cc, Cc, 9, 9, 1, 0
    E.fill_taps(b, *E.taps_fwd(3, 1, 1)); b.act, b.res = act2, E.null_tensor()
    lib.myolo_set_option(b'pair_th', th)
    L.check(lib.myolo_conv_pair(C.byref(a), C.byref(b), L.stream_ptr()))
    torch.cuda.synchronize()
    got = yb.float().cpu()
    err = (got - ref).abs()
    rel = float((got - ref).norm() / ref.norm())
    print(f'== C={Cc} {B}x{H}x{W} mode={mode} th={th}: rel {rel:.3e}')
    if rel > 1e-2:
        print('  by row mod 8   :', [round(float(err[:, i::8].mean()), 3) for i in range(8)])
        print('  by col mod 16  :', [round(float(err[:, :, i::16].mean()), 3) for i in range(16)])
        print('  by chan // 8   :', [round(float(err[..., i * 8:(i + 1) * 8].mean()), 3) for i in range(Cc // 8)])
        print('  by chan mod 8  :', [round(float(err[..., i::8].mean()), 3) for i in range(8)])
        print('  ref rms', round(float(ref.pow(2).mean().sqrt()), 3), 'got rms', round(float(got.pow(2).mean().sqrt()), 3))
        # is got a permutation of ref along channels?  best matching channel for got channel 0..7 at one pixel block
        gm, rm = got[0, 8:24, 16:48].reshape(-1, Cc), ref[0, 8:24, 16:48].reshape(-1, Cc)
        cc = (gm.t() @ rm) / (gm.norm(dim=0).view(-1, 1) * rm.norm(dim=0).view(1, -1) + 1e-9)
        print('  got chan -> best ref chan (first 16):', cc.argmax(1)[:16].tolist(), [round(float(v), 2) for v in cc.max(1).values[:16]])


for mode in ('gemm1', 'gemm2', 'full'):
    run(64, 1, 32, 48, mode, 8)
    run(128, 1, 32, 48, mode, 8)
run(64, 1, 32, 48, 'gemm1', 4)
