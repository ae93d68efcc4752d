"""chronological listing of the LAST complete training step of a rocprofv3 kernel trace of bench.py: per kernel queue, start offset, duration,
gap to the previous kernel of the same queue.  usage: python scripts/trace_list.py <kernel_trace.csv> [marker]
(marker: substring of the kernel that ends a step; default mt_ema_kernel = the training step; `seg_argmax` = one detect.py frame)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
MARK = sys.argv[2] if len(sys.argv) > 2 else 'mt_ema_kernel'
ends = [i for i, r in enumerate(rows) if MARK in r['Kernel_Name']]
a, b = ends[-2] + 1, ends[-1] + 1
step = rows[a:b]
t0 = step[0]['s']
qkey = 'Queue_Id' if 'Queue_Id' in step[0] else 'Stream_Id'
qs = sorted({r[qkey] for r in step}, key=lambda q: -sum(1 for r in step if r[qkey] == q))
last = {}
for r in step:
    q = qs.index(r[qkey])
    gap = (r['s'] - last[q]) / 1e3 if q in last else 0.0
    last[q] = r['e']
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '')
    print(f"q{q} +{(r['s'] - t0) / 1e3:8.1f} {(r['e'] - r['s']) / 1e3:7.1f} us  gap {gap:6.1f}  {name[:110]}")
