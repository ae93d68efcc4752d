# per-kernel times of the pooling / resize / loss passes inside the bench's training step (rocprofv3 kernel stats over 7 steps)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pool -o train -- python bench.py --steps 5 --warmup 2 --no-kernel-timing --no-infer --no-cpu-baseline --no-stock-baseline > gpurun_out/prof_pool.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_pool/**/train_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if any(k in n for k in ('spp', 'bilinear', 'aap', 'pyr_up', 'copy_up', 'gate', 'fill_words', 'det_obj', 'det_cand', 'focus', 'seg_upce', 'tiny', 'cast', 'ohem', 'ce_')):
        print(f"{float(r['TotalDurationNs']) / 7 / 1e3:8.1f} us/step  x{int(r['Calls']) / 7:5.1f}  avg {float(r['AverageNs']) / 1e3:7.1f}  {n[:90]}")
PY
rm -rf gpurun_out/prof_pool
