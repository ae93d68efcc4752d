"""A/B of the reference loop body as written (train.py:364-401: detection pass + segmentation pass per iteration) with the pruned
one-loss backward lists on and off (engine.PRUNE_BWD), same process, same plan.  usage: python scripts/train_py_ab.py [iters]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from multiyolov5_amd import engine as E  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
sys.argv = sys.argv[:1]
args = bench.parse()
dev = torch.device('cuda', 0)
tr = bench.Trainer(args, 1, 0, dev)
out = {}
for rnd in range(2):
    for name, flag in (('pruned', True), ('full', False)):
        E.PRUNE_BWD = flag
        r = bench.train_py_rate(tr, iters=iters, warm=4)
        out[f'{name}_{rnd}'] = {'pairs_per_s': round(r['pairs_per_s'], 1), 'ms_per_iteration': round(r['ms_per_iteration'], 3)}
E.PRUNE_BWD = True
plan = next(iter(tr.model.__dict__['_plans'].values())).plan
out['programs'] = {str(sorted(k[1]) if k[1] is not None else None): (v.n if v else None) for k, v in plan.__dict__['_nprog_bwd'].items()}
print(json.dumps(out))
