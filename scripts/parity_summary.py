"""Condense gpurun_out/parity_log.jsonl (every comparison tests/gpu_util.check made during the last `pytest -m gpu` run) into a small
committed record: per test family the number of comparisons, how many failed, and the worst error relative to its tolerance.
python scripts/parity_summary.py [log] > profiles/<tag>_parity_summary.md"""
import json
import sys
from collections import defaultdict

path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/parity_log.jsonl'
fam = defaultdict(lambda: {'n': 0, 'bad': 0, 'worst': 0.0, 'worst_name': '', 'worst_rel': 0.0, 'tol': 0.0})
n = 0
for line in open(path):
    try:
        r = json.loads(line)
    except ValueError:
        continue
    n += 1
    key = r['name'].split('/')[0]
    f = fam[key]
    f['n'] += 1
    f['bad'] += 0 if r.get('ok', True) else 1
    ratio = r['rel_l2'] / r['tol'] if r.get('tol') else 0.0
    if ratio > f['worst']:
        f.update(worst=ratio, worst_name=r['name'], worst_rel=r['rel_l2'], tol=r['tol'])
print(f'# parity log of the last `pytest -m gpu` run: {n} comparisons (tests/gpu_util.check: relative L2 error against the oracle / torch fp32)\n')
print('| family | comparisons | failed | worst error / tolerance | worst case (rel L2, tol) |\n|---|---|---|---|---|')
for k, f in sorted(fam.items(), key=lambda kv: -kv[1]['n']):
    print(f"| {k} | {f['n']} | {f['bad']} | {f['worst']:.2f} | `{f['worst_name'][:110]}` ({f['worst_rel']:.2e}, {f['tol']:.0e}) |")
