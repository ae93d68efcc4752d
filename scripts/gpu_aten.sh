cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
MYOLO_GRAPH_TRAIN=0 timeout 600 python scripts/aten_ops.py 2>&1 | tail -45 | cut -c1-260
