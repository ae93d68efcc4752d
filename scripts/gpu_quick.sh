cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for sl in 0 64 0 64 0 64 0 64 0 64; do
rm -f gpurun_out/parity_log.jsonl
MYOLO_BN_SLICE=$sl timeout 300 python -m pytest tests/test_gpu_configs.py -x -q -k config1 2>&1 | tail -1 | tr '\n' ' '
python - <<'PY'
import json
rows=[json.loads(l) for l in open('gpurun_out/parity_log.jsonl')]
c=[r['rel_l2'] for r in rows if r['name'].startswith('cfg1/grad') and r['tol']==1e-2]
print(' max %.2e' % max(c))
PY
done
