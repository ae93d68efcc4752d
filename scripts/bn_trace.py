"""per-layer achieved bandwidth of the BN kernels from a rocprofv3 kernel trace (last profiled step).
usage: python scripts/bn_trace.py gpurun_out/prof_<tag>/train_kernel_trace.csv"""
import csv, sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiyolov5_amd import runtime as R, _lib as L
from multiyolov5_amd.models.yolo import Model
from tests.util import CFG, TAGS
m = Model(os.path.join(CFG, TAGS['s_psp'])); m.train(True)
h = R.PlanHolder(m, [torch.zeros(16, 3, 512, 1024)], ('t', 0), torch.float16, True)
shapes = collections.defaultdict(list)
def rec(calls):
    for c in calls:
        if c.name.startswith('myolo_bn_act'):
            for a in c.args:
                obj = getattr(a, '_obj', None)
                if isinstance(obj, L.Tensor):
                    shapes[c.name].append((obj.n, obj.h, obj.w, obj.c, obj.sw)); break
for o in h.plan.ops: rec(o.fwd_calls)
for o in reversed(h.plan.ops): rec(o.bwd_calls)
rows = list(csv.DictReader(open(sys.argv[1])))
kmap = {'bn_act_fwd_kernel': ('myolo_bn_act_fwd', 4), 'bn_act_bwd_reduce_kernel': ('myolo_bn_act_bwd_reduce', 4), 'bn_act_bwd_apply_kernel': ('myolo_bn_act_bwd_apply', 6)}
for kn, (cn, bpe) in kmap.items():
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if kn in r['Kernel_Name']]
    n = len(shapes[cn]); d = d[-n:]
    print(f'== {kn}: {n} calls, {sum(d):.0f} us')
    agg = collections.defaultdict(lambda: [0, 0.0, 0])
    for s, t in zip(shapes[cn], d):
        a = agg[s]; a[0] += 1; a[1] += t; a[2] += s[0]*s[1]*s[2]*s[3]*bpe
    for s, (cnt, t, b) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'  {s}: x{cnt} {t/cnt:7.1f} us/call  {b/t/1e3:6.0f} GB/s   total {t:6.0f} us')
