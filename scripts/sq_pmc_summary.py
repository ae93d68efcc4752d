"""per-kernel means of the SQ counters of one rocprofv3 --pmc pass (x_counter_collection.csv).  usage: sq_pmc_summary.py <csv> [name filter]"""
import collections
import csv
import re
import sys

per = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.Counter())
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])[:70]
    if flt and flt not in k:
        continue
    per[k][r['Counter_Name']] += float(r['Counter_Value'])
    n[k][r['Counter_Name']] += 1
for k in sorted(per):
    c = {a: per[k][a] / max(1, n[k][a]) for a in per[k]}
    wc = c.get('SQ_WAVE_CYCLES', 0) or 1
    print(k, 'launches', max(n[k].values()))
    print('   ' + ' '.join(f'{a}={v:.3g}' for a, v in sorted(c.items())))
    if 'SQ_LDS_IDX_ACTIVE' in c:
        print(f"   lds conflict share {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, c['SQ_LDS_IDX_ACTIVE']):.2f}  wait_any/wave {c.get('SQ_WAIT_ANY', 0) / wc:.2f} "
              f"wait_inst/wave {c.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} (lds {c.get('SQ_WAIT_INST_LDS', 0) / wc:.2f}) active/wave {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f}")
