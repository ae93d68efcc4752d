"""per-launch table of the myolo_conv kernels (forward convs + dgrad) of the last profiled step: shape, kernel, duration,
algorithmic GB/s and TFLOP/s.  usage: python scripts/conv_trace.py gpurun_out/prof_<tag>/train_kernel_trace.csv"""
import csv, sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiyolov5_amd import runtime as R, _lib as L, engine as E
from multiyolov5_amd.models.yolo import Model
from tests.util import CFG, TAGS
m = Model(os.path.join(CFG, TAGS['s_psp'])); m.train(True)
h = R.PlanHolder(m, [torch.zeros(16, 3, 512, 1024)], ('t', 0), torch.float16, True)
calls = []
for o in h.plan.ops:
    calls += [('f', c) for c in o.fwd_calls if c.name in ('myolo_conv', 'myolo_conv_dgrad_s2', 'myolo_conv_bn_act')]
for o in reversed(h.plan.ops):
    calls += [('b', c) for c in o.bwd_calls if c.name in ('myolo_conv', 'myolo_conv_dgrad_s2', 'myolo_conv_dgrad_bn')]
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sgd = [i for i, r in enumerate(rows) if 'mt_sgd' in r['Kernel_Name']]
step = rows[sgd[-2] + 1:sgd[-1] + 1]
ks = [r for r in step if any(n in r['Kernel_Name'] for n in ('conv_igemm_kernel', 'conv_stream_kernel', 'conv_halo_kernel', 'conv_mid_kernel', 'conv_midx_kernel'))]
print(len(calls), len(ks))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, ''])
import re
pairs, ki = [], 0
for ph, c in calls:                     # a stride-2 dgrad is one parity-fused halo kernel (template arg S2 = 1) or four plain launches
    if c.name == 'myolo_conv_dgrad_s2':
        m_ = re.search(r'conv_halo_kernel<\d+, \d+, \d+, \d+, \d+, (\d)', ks[ki]['Kernel_Name'])
        nk = 1 if (m_ and m_.group(1) == '1') else c.args[1]
    else:
        nk = 1
    grp = ks[ki:ki + nk]
    ki += nk
    r = dict(grp[0])
    r['Start_Timestamp'] = 0
    r['End_Timestamp'] = sum(int(g['End_Timestamp']) - int(g['Start_Timestamp']) for g in grp)
    pairs.append(((ph, c), r))
print('kernels consumed', ki, 'of', len(ks))
for (ph, c), r in pairs:
    d = E._s2_descs(c)[0] if c.name == 'myolo_conv_dgrad_s2' else E._conv_desc_of(c)
    t = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    kn = 'stream' if 'stream' in r['Kernel_Name'] else ('halo' if 'halo' in r['Kernel_Name'] else ('midx' if 'conv_midx' in r['Kernel_Name'] else ('mid' if 'conv_mid' in r['Kernel_Name'] else 'igemm')))
    key = (ph, d.x.h, d.x.w, d.x.c, d.y.h, d.y.w, d.y.c, d.ntaps, kn)
    a = agg[key]; a[0] += 1; a[1] += t; a[2] += E.conv_call_bytes(c); a[3] += E.conv_call_flops(c)
tot = sum(a[1] for a in agg.values())
print(f'total {tot:.0f} us')
HBM, MFMA = 8.0e12, 2.5e15          # peaks used by bench.py (MI355X_MICROARCH.md)
ideal_tot = sum(max(a[2] / HBM, a[3] / MFMA) * 1e6 for a in agg.values())
print(f'per-layer roofline (max(bytes/8 TB/s, flops/2.5 PF/s)) summed: {ideal_tot:.0f} us -> achieved fraction {ideal_tot / tot:.3f}')
print('ph  in(HxWxC) -> out(HxWxC) taps kern   n   us/call  GB/s  TF/s  total_us  roofline_us/call  frac')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    ph, xh, xw, xc, yh, yw, yc, nt, kn = k
    print(f'{ph} {xh:4d}x{xw:4d}x{xc:4d} -> {yh:4d}x{yw:4d}x{yc:4d} t{nt} {kn:6s} x{a[0]:2d} {a[1]/a[0]:8.1f} {a[2]/a[1]/1e3:6.0f} {a[3]/a[1]/1e6:6.1f} {a[1]:8.0f} {max(a[2] / HBM, a[3] / MFMA) * 1e6 / a[0]:8.1f} {max(a[2] / HBM, a[3] / MFMA) * 1e6 / a[1]:6.2f}')
