"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (scripts/gpu_pmc.sh) -> profiles/<tag>_pmc.json: HBM bytes per training step and per
myolo_conv launch, plus a per-kernel-family table.  hbm bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB
and FETCH_SIZE counts 64-byte requests as 32 on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section).
Round 6: only COMPLETE steady-state steps are counted -- the dispatches up to the `skip`-th mt_ema_kernel (the step
that builds the plan: ~240 one-off torch.zeros fills writing ~2 GB, and the warm-up) are dropped, the rest is divided by the number of
optimizer steps it contains.
usage: python scripts/pmc_summary.py <tag> [skip_steps=2]   (reads gpurun_out/pmc_<tag>_{FETCH,WRITE}_SIZE/x_counter_collection.csv)"""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 2             # bench.py --steps 5 --warmup 2: 7 steps in the trace, the first 2 dropped
steps = None


def fam(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(anonymous namespace\)::|stream::|midx::|mid::|halo::|wgt::|small::', '', name)
    m = re.search(r'(\w+_kernel|__amd_rocclr_\w+|at::native::\w+)', name)
    if 'N12_GLOBAL__N_' in name:
        m2 = re.search(r'N_1?\d+(\w+?_kernel)', name)
        if m2:
            return m2.group(1)
    return m.group(1) if m else name[:40]


def load(counter):
    p = os.path.join(ROOT, 'gpurun_out', f'pmc_{tag}_{counter}', 'x_counter_collection.csv')
    global steps
    per, n = collections.defaultdict(float), collections.Counter()
    rows_ = [r for r in csv.DictReader(open(p)) if r['Counter_Name'] == counter]
    key = 'Dispatch_Id' if rows_ and 'Dispatch_Id' in rows_[0] else None
    if key:
        rows_.sort(key=lambda r: int(r[key]))
    ends = [i for i, r in enumerate(rows_) if 'mt_ema_kernel' in r['Kernel_Name']]
    assert len(ends) > skip, f'{len(ends)} optimizer steps in the {counter} pass, cannot skip {skip}'
    rows_ = rows_[ends[skip - 1] + 1:ends[-1] + 1] if skip > 0 else rows_[:ends[-1] + 1]
    nsteps = len(ends) - skip
    assert steps in (None, nsteps), (steps, nsteps)
    steps = nsteps
    for r in rows_:
        f = fam(r['Kernel_Name'])
        per[f] += float(r['Counter_Value'])
        n[f] += 1
    return per, n


fetch, nf = load('FETCH_SIZE')
write, nw = load('WRITE_SIZE')
rows = []
for f in sorted(set(fetch) | set(write)):
    rd, wr = 2 * fetch.get(f, 0.0) * 1024 / steps, write.get(f, 0.0) * 1024 / steps
    rows.append({'kernel': f, 'launches_per_step': nf.get(f, nw.get(f, 0)) / steps, 'read_MB_per_step': rd / 1e6, 'write_MB_per_step': wr / 1e6})
rows.sort(key=lambda r: -(r['read_MB_per_step'] + r['write_MB_per_step']))
conv = [r for r in rows if r['kernel'] in ('conv_igemm_kernel', 'conv_stream_kernel', 'conv_halo_kernel', 'conv_mid_kernel', 'conv_midx_kernel')]
conv_bytes = sum((r['read_MB_per_step'] + r['write_MB_per_step']) * 1e6 for r in conv)
conv_launches = sum(r['launches_per_step'] for r in conv)
out = {
    'source': f'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 5 --warmup 2, the {steps} steady-state steps after '
              f'the first {skip} (plan build + warm-up dropped); hbm = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE half-count correction, '
              'MI355X_MICROARCH.md)',
    'steps_counted': steps,
    'conv_launches_per_step': conv_launches,
    'conv_hbm_bytes_per_launch': conv_bytes / max(conv_launches, 1),
    'step_hbm_read_bytes': sum(r['read_MB_per_step'] for r in rows) * 1e6,
    'step_hbm_write_bytes': sum(r['write_MB_per_step'] for r in rows) * 1e6,
    'per_kernel': rows[:40],
}
dst = os.path.join(ROOT, 'profiles', f'{tag}_pmc.json')
json.dump(out, open(dst, 'w'), indent=1)
print('wrote', dst, {k: (round(v / 1e9, 2) if 'bytes' in k and 'launch' not in k else v) for k, v in out.items() if k not in ('per_kernel', 'source')})
for r in rows[:14]:
    print(f"  {r['kernel']:34s} x{r['launches_per_step']:6.1f}  read {r['read_MB_per_step']:9.1f} MB  write {r['write_MB_per_step']:9.1f} MB")
