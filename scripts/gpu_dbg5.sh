cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in 4 8; do echo "MAXNT=$v"; MYOLO_STREAM_MAXNT=$v MYOLO_STREAM_MIN_TILES=${MT:-2048} timeout 200 python scripts/conv_bench.py 16 5 2>&1 | tail -6; done
echo "MIN_TILES=512 MAXNT=8"; MYOLO_STREAM_MAXNT=8 MYOLO_STREAM_MIN_TILES=512 timeout 200 python scripts/conv_bench.py 16 5 2>&1 | tail -6
