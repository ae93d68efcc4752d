"""per-launch table of the weight-gradient kernels of the last profiled step.  usage: python scripts/wgrad_trace.py <kernel_trace.csv>"""
import csv, sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiyolov5_amd import runtime as R, _lib as L, engine as E
from multiyolov5_amd.models.yolo import Model
from tests.util import CFG, TAGS
m = Model(os.path.join(CFG, TAGS['s_psp'])); m.train(True)
h = R.PlanHolder(m, [torch.zeros(16, 3, 512, 1024)], ('t', 0), torch.float16, True)
calls = []
for o in reversed(h.plan.ops):
    calls += [c for c in o.bwd_calls if c.name == 'myolo_conv_wgrad']
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sgd = [i for i, r in enumerate(rows) if 'mt_sgd' in r['Kernel_Name']]
step = rows[sgd[-2] + 1:sgd[-1] + 1]
ks = [r for r in step if 'wgrad_kernel' in r['Kernel_Name'] or 'wgrad_fused_kernel' in r['Kernel_Name']]
red = [r for r in step if 'wgrad_reduce' in r['Kernel_Name']]
print(len(calls), len(ks), len(red))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0])
for c, r in zip(calls, ks):
    d = None
    for a in c.args:
        ob = getattr(a, '_obj', None)
        if isinstance(ob, L.WgradDesc): d = ob
    t = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    kn = 'fused9' if 'fused' in r['Kernel_Name'] else 'pertap'
    key = (d.x.h, d.x.w, d.cin, d.dy.h, d.dy.w, d.cout, d.ntaps, d.ksplit, kn, int(r['Grid_Size_X']) // 256, int(r['Grid_Size_Y']))
    by = (d.x.n * d.x.h * d.x.w * d.cin + d.dy.n * d.dy.h * d.dy.w * d.cout) * 2
    fl = 2.0 * d.dy.n * d.dy.h * d.dy.w * d.cout * d.cin * d.ntaps
    a = agg[key]; a[0] += 1; a[1] += t; a[2] += by; a[3] += fl
print('total', sum(a[1] for a in agg.values()), 'reduce', sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in red))
print(' x(HxWxC)  dy(HxWxC) taps ks kern gridx gridy  n  us/call GB/s TF/s total')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(k, f'x{a[0]} {a[1]/a[0]:7.1f} {a[2]/a[1]/1e3:6.0f} {a[3]/a[1]/1e6:6.1f} {a[1]:7.0f}')
