"""host API calls and kernels of the LAST complete detect.py frame on one clock: rocprofv3 --kernel-trace --hip-runtime-trace csv pair.
usage: python scripts/trace_api_list.py <kernel_trace.csv> <hip_api_trace.csv> [marker]"""
import csv
import sys

kr = list(csv.DictReader(open(sys.argv[1])))
ar = list(csv.DictReader(open(sys.argv[2])))
MARK = sys.argv[3] if len(sys.argv) > 3 else 'seg_argmax'
for r in kr:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
kr.sort(key=lambda r: r['s'])
ends = [i for i, r in enumerate(kr) if MARK in r['Kernel_Name']]
a, b = ends[-2] + 1, ends[-1] + 1
step = kr[a:b]
t0, t1 = step[0]['s'] - 150000, step[-1]['e']
qkey = 'Queue_Id' if 'Queue_Id' in step[0] else 'Stream_Id'
qs = sorted({r[qkey] for r in step}, key=lambda q: -sum(1 for r in step if r[qkey] == q))
ev = [(r['s'], f"   GPU q{qs.index(r[qkey])} {(r['e'] - r['s']) / 1e3:6.1f} us  {r['Kernel_Name'][:70]}") for r in step]
for r in ar:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if t0 <= s <= t1:
        ev.append((s, f"HOST t{r.get('Thread_Id', '?')[-3:]} {(e - s) / 1e3:6.1f} us  {r['Function']}"))
ev.sort()
for t, s in ev:
    print(f'+{(t - step[0]["s"]) / 1e3:8.1f}  {s}')
