"""per-shape microbenchmark of myolo_conv (eval-mode Conv: conv + BN affine + SiLU epilogue) at the yolov5s+PSP layer shapes,
B=16, 512x1024.  prints time, algorithmic GB/s (in+w+out) and TFLOP/s per shape."""
import sys, time
import torch
sys.path.insert(0, '.')
from multiyolov5_amd.models import common as C
from multiyolov5_amd import engine as E

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NSH = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dt = torch.float16
# (cin, cout, k, s, d, H, W) of the input map
SHAPES = [
    (64, 384, 1, 1, 1, 64, 128), (64, 256, 1, 1, 1, 64, 128), (128, 512, 1, 1, 1, 32, 64), (128, 256, 1, 1, 1, 32, 64), (256, 512, 1, 1, 1, 16, 32),
    (256, 128, 3, 1, 1, 64, 128), (64, 64, 3, 1, 1, 64, 128), (128, 128, 3, 1, 1, 32, 64), (32, 64, 3, 2, 1, 256, 512),
    (64, 128, 3, 2, 1, 128, 256), (128, 256, 3, 2, 1, 64, 128), (256, 512, 3, 2, 1, 32, 64), (256, 256, 3, 1, 1, 16, 32),
    (256, 128, 1, 1, 1, 64, 128), (16, 32, 3, 1, 1, 256, 512), (256, 128, 1, 1, 1, 32, 64), (256, 256, 1, 1, 1, 32, 64),
    (512, 256, 1, 1, 1, 16, 32), (384, 64, 1, 1, 1, 64, 128), (32, 32, 3, 1, 1, 128, 256), (64, 64, 3, 1, 2, 64, 128),
    (128, 128, 1, 1, 1, 64, 128), (1024, 512, 1, 1, 1, 16, 32), (512, 512, 1, 1, 1, 16, 32), (512, 128, 1, 1, 1, 32, 64),
    (256, 64, 1, 1, 1, 64, 128), (128, 128, 1, 1, 1, 32, 64), (64, 32, 1, 1, 1, 128, 256), (64, 64, 1, 1, 1, 128, 256),
    (128, 64, 1, 1, 1, 64, 128), (64, 64, 1, 1, 1, 64, 128), (32, 32, 1, 1, 1, 128, 256),
]
tot_t = tot_b = tot_f = 0
print(f'{"cin":>5}{"cout":>5} k s d {"HxW":>9} {"us":>8} {"GB/s":>8} {"TF/s":>7}  ideal_us(6.3TB/s|2.5PF)')
for cin, cout, k, s, d, H, W in SHAPES[:NSH]:
    m = C.Conv(cin, cout, k, s).to('cuda')
    if d != 1:
        m.conv.dilation, m.conv.padding = (d, d), (d, d)
    m.eval()
    x = torch.randn(B, cin, H, W, device='cuda', dtype=dt).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        m(x)
        plans = list(m._plans.values())
        plan = plans[0].plan
        conv_calls = [c for op in plan.ops for c in op.fwd_calls if c.name == 'myolo_conv']
        st = torch.cuda.current_stream().cuda_stream
        import ctypes
        stp = ctypes.c_void_p(st)
        for _ in range(3):
            for c in conv_calls: c(stp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            for c in conv_calls: c(stp)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e-3
    by = sum(E.conv_call_bytes(c) for c in conv_calls); fl = sum(E.conv_call_flops(c) for c in conv_calls)
    tot_t += t; tot_b += by; tot_f += fl
    print(f'{cin:5d}{cout:5d} {k} {s} {d} {H:4d}x{W:<4d} {t*1e6:8.1f} {by/t/1e9:8.0f} {fl/t/1e12:7.1f}  {max(by/6.3e12, fl/2.5e15)*1e6:6.1f}')
print(f'TOTAL {tot_t*1e6:.0f} us  {tot_b/tot_t/1e9:.0f} GB/s  {tot_f/tot_t/1e12:.1f} TF/s')
