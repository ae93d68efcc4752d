"""timeline of the detect.py frame loop from a rocprofv3 kernel trace of `bench.py --stage infer`: per steady-state frame (one
seg_argmax kernel ends a frame) the wall time, the time any queue is busy, the main queue's idle gaps and the kernels around the largest
of them.  usage: python scripts/trace_infer_timeline.py <kernel_trace.csv> [frames to skip at both ends]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else 'Stream_Id'
ends = [i for i, r in enumerate(rows) if 'seg_argmax' in r['Kernel_Name']]
ends = ends[skip:-skip]
if len(ends) < 3:
    print('too few frames'); sys.exit(0)


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
    return tot + (ce - cs if cs is not None else 0)


walls, busys, mains, gapsum = [], [], [], defaultdict(lambda: [0, 0.0])
for a, b in zip(ends, ends[1:]):
    fr = rows[a + 1:b + 1]
    t0, t1 = rows[a]['e'], rows[b]['e']
    walls.append(t1 - t0)
    busys.append(union([(max(r['s'], t0), r['e']) for r in fr]))
    byq = defaultdict(list)
    for r in fr:
        byq[r[qkey]].append(r)
    mq = max(byq, key=lambda q: len(byq[q]))
    m = sorted(byq[mq], key=lambda r: r['s'])
    mains.append(sum(r['e'] - r['s'] for r in m))
    prev_e, prev_n = t0, 'seg_argmax (previous frame)'
    for r in m:
        g = r['s'] - prev_e
        if g > 2000:
            k = (prev_n.split('(')[0].split('<')[0][-34:], r['Kernel_Name'].split('(')[0].split('<')[0][-34:])
            gapsum[k][0] += 1
            gapsum[k][1] += g
        prev_e, prev_n = max(prev_e, r['e']), r['Kernel_Name']
n = len(walls)
print(f'{n} frames: wall {sum(walls) / n / 1e3:.1f} us, GPU busy (any queue) {sum(busys) / n / 1e3:.1f} us, idle {(sum(walls) - sum(busys)) / n / 1e3:.1f} us, '
      f'main-queue kernel time {sum(mains) / n / 1e3:.1f} us')
print('main-queue gaps > 2 us, per frame (count, us), by the kernels around them:')
for k, (c, t) in sorted(gapsum.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f'  {c / n:5.2f} x  {t / n / 1e3:7.1f} us   {k[0]}  ->  {k[1]}')
