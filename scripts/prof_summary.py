#!/usr/bin/env python
"""rocprofv3 --kernel-trace --stats (csv) -> profiles/<tag>_kernel_stats.csv + a markdown table.
usage: python scripts/prof_summary.py gpurun_out/prof_<tag>/train_kernel_stats.csv <tag> <steps_profiled> [bench json line file]
       [unit (step|frame)] [the profiled command]"""
import csv
import json
import os
import re
import shutil
import sys

src, tag, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
bench = sys.argv[4] if len(sys.argv) > 4 else None
unit = sys.argv[5] if len(sys.argv) > 5 else 'step'
cmdline = sys.argv[6] if len(sys.argv) > 6 else ('rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 5 --warmup 2 '
                                                 '--no-kernel-timing --no-infer --no-cpu-baseline')
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(root, 'profiles'), exist_ok=True)
dst = os.path.join(root, 'profiles', f'{tag}_kernel_stats.csv')
shutil.copyfile(src, dst)
rows = list(csv.DictReader(open(src)))
tot = sum(float(r['TotalDurationNs']) for r in rows)


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    m = re.match(r'_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I(DF16_|f)', n)
    if m:
        return m.group(1) + ('<f16>' if m.group(2) == 'DF16_' else '<f32>') + (re.search(r'Li(\d+)E', n).group(0) if 'Li' in n else '')
    return n.split('(')[0][:60]


with open(os.path.join(root, 'profiles', f'{tag}_summary.md'), 'w') as f:
    f.write(f'# rocprofv3 --kernel-trace --stats, tag `{tag}`\n\n')
    f.write(f'command: `{cmdline}` ({steps} {unit}s profiled incl. warm-up); '
            f'total kernel time {tot / steps / 1e6:.3f} ms/{unit}\n\n')
    if bench and os.path.exists(bench):
        line = [l for l in open(bench) if l.startswith('{')][-1]
        j = json.loads(line)
        f.write('bench.py line of the same build:\n\n```json\n' + json.dumps(j, indent=1) + '\n```\n\n')
    f.write(f'| kernel | calls/{unit} | avg µs | ms/{unit} | % |\n|---|---|---|---|---|\n')
    for r in rows[:40]:
        f.write(f"| {short(r['Name'])} | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | "
                f"{float(r['TotalDurationNs']) / steps / 1e6:.3f} | {float(r['Percentage']):.1f} |\n")
print('wrote', dst)
