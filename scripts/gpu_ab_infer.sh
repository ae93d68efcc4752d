# same-session A/B of environment switches on the detect.py path: bash scripts/gpu_ab_infer.sh "ENV=a" "ENV=b" ...
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "$@"; do
echo -n "$v: "; env $v timeout 300 python bench.py --stage infer 2>&1 | tail -1 | cut -c1-200
done; done
