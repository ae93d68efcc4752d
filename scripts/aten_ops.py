"""which ATen ops does one bench step still issue?  torch.profiler over 3 steps -> per-op counts (per step) with call sites"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
sys.argv = ['bench.py']
args = bench.parse()
dev = torch.device('cuda', 0)
tr = bench.Trainer(args, 1, 0, dev)
for _ in range(5):
    tr.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(3):
        tr.step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=4)
rows = [e for e in ka if e.key.startswith('aten::') and e.count >= 3]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    st = ' <- '.join(s.split('/')[-1][:60] for s in (e.stack or [])[:3])
    print(f'{e.count / 3:7.1f}/step  {e.key:34s} cuda {e.device_time_total / 3:8.1f} us  {st}')
