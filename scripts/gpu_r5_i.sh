# round 5, call I: training launch lists replayed as hipGraphs (last measured in round 2 at 10.4 ms per step), and the Python launch loop for reference
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for E in "A=0" "MYOLO_GRAPH_TRAIN=1" "MYOLO_GRAPH_TRAIN=1 MYOLO_BWD_SEGMENTS=4" "MYOLO_GRAPH_TRAIN=1 MYOLO_GRAPH_BWD=fork" "MYOLO_NATIVE_EXEC=0" "A=1"; do
  R=$(env $E timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-infer --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.3f ms  %.0f img/s' % (j['ms_per_step'], j['value']))" 2>&1 | tail -1)
  echo "[$E] train: $R" | tee -a gpurun_out/r5i_graph.txt
done
