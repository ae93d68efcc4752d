cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash scripts/gpu_inf.sh r3c 2>&1 | grep -v "^wrote" | cut -c1-260
echo "--- A/B stream / halo thresholds @2048"
for v in "X=0" "MYOLO_STREAM_MIN_TILES=1024" "MYOLO_STREAM_MIN_TILES=512" "MYOLO_SMALL_MAX_TILES=512" "MYOLO_NO_AAP_MULTI=1"; do
echo -n "$v: "; env $v timeout 300 python bench.py --stage infer --no-cpu-baseline --steps 60 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value'],1), j['stage_ms']['forward'])"
done
echo "--- host time"; timeout 300 python scripts/host_time.py 2>&1 | tail -3 | cut -c1-300
