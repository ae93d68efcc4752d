"""timeline analysis of a rocprofv3 kernel trace of bench.py: for the LAST complete training step, per stream (queue): busy time,
gaps, and the overlap between streams; top kernels by time.  usage: python scripts/trace_timeline.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
# step boundaries: the optimizer kernel mt_sgd_kernel ends a step
ends = [i for i, r in enumerate(rows) if 'mt_ema_kernel' in r['Kernel_Name']]
if len(ends) < 2:
    print('need >= 2 steps'); sys.exit(0)
a, b = ends[-2] + 1, ends[-1] + 1
step = rows[a:b]
t0, t1 = step[0]['s'], max(r['e'] for r in step)
print(f'step: {len(step)} kernels, wall {(t1 - t0) / 1e3:.1f} us')
qkey = 'Queue_Id' if 'Queue_Id' in step[0] else 'Stream_Id'
byq = defaultdict(list)
for r in step:
    byq[r[qkey]].append(r)


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot


allbusy = union([(r['s'], r['e']) for r in step])
print(f'GPU busy (any queue) {allbusy / 1e3:.1f} us, idle {(t1 - t0 - allbusy) / 1e3:.1f} us')
for q, rs in byq.items():
    busy = union([(r['s'], r['e']) for r in rs])
    summ = sum(r['e'] - r['s'] for r in rs)
    gaps = []
    rs2 = sorted(rs, key=lambda r: r['s'])
    for x, y in zip(rs2, rs2[1:]):
        gaps.append(max(0, y['s'] - x['e']))
    print(f'queue {q}: {len(rs)} kernels, busy {busy / 1e3:.1f} us (sum of durations {summ / 1e3:.1f}), first {(rs2[0]["s"] - t0) / 1e3:.1f} last end {(max(r["e"] for r in rs) - t0) / 1e3:.1f}, '
          f'gaps: sum {sum(gaps) / 1e3:.1f} us, >5us: {sum(1 for g in gaps if g > 5000)}')
# phases on the main queue (largest kernel count)
mainq = max(byq, key=lambda q: len(byq[q]))
main = sorted(byq[mainq], key=lambda r: r['s'])
agg = defaultdict(lambda: [0, 0])
for r in step:
    n = r['Kernel_Name'].split('(')[0][:60]
    agg[n][0] += 1
    agg[n][1] += r['e'] - r['s']
print('top kernels (sum of durations, us):')
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f'  {t / 1e3:9.1f}  x{c:4d}  avg {t / c / 1e3:7.1f}  {n}')
# main-queue time when the other queues are idle vs busy
others = [(r['s'], r['e']) for q, rs in byq.items() if q != mainq for r in rs]
ob = union(others)
print(f'other queues busy {ob / 1e3:.1f} us of the step')
# the 12 largest gaps on the main queue with the kernels around them
g = [(y['s'] - x['e'], x, y) for x, y in zip(main, main[1:])]
print('largest main-queue gaps:')
for gap, x, y in sorted(g, key=lambda t: -t[0])[:12]:
    print(f'  {gap / 1e3:7.1f} us at +{(x["e"] - t0) / 1e3:8.1f}: {x["Kernel_Name"][:40]} -> {y["Kernel_Name"][:40]}')
# duration histogram of the main queue (where does the dependent chain spend its time: many short launches or few long ones?)
bins = [(0, 6), (6, 9), (9, 12), (12, 16), (16, 24), (24, 40), (40, 80), (80, 1e9)]
print('main-queue duration histogram (us): count, total us')
for lo, hi in bins:
    rs = [r for r in main if lo * 1e3 <= r['e'] - r['s'] < hi * 1e3]
    fam = defaultdict(int)
    for r in rs:
        fam[r['Kernel_Name'].split('<')[0].split('(')[0].replace('void ', '')[:28]] += 1
    top = ', '.join(f'{k} x{v}' for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:5])
    print(f'  [{lo:3.0f},{hi:5.0f}): {len(rs):4d}  {sum(r["e"] - r["s"] for r in rs) / 1e3:8.1f}   {top}')
# the end of the backward: the last kernels of every queue before the optimizer (the weight-gradient queue's exposed tail)
opt = [i for i, r in enumerate(step) if 'mt_' in r['Kernel_Name'] or 'scaler' in r['Kernel_Name']]
if opt:
    first_opt = step[opt[0]]
    print(f'end of the backward (optimizer starts at +{(first_opt["s"] - t0) / 1e3:.1f} us):')
    for q, rs in byq.items():
        pre = sorted([r for r in rs if r['e'] <= first_opt['s']], key=lambda r: r['s'])[-7:]
        for r in pre:
            print(f'  q{q} +{(r["s"] - t0) / 1e3:8.1f} .. +{(r["e"] - t0) / 1e3:8.1f} ({(r["e"] - r["s"]) / 1e3:6.1f} us)  {r["Kernel_Name"][:70]}')
