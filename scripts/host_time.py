import sys, time, torch
sys.path.insert(0,'.')
import bench
class A: pass
a=A(); a.batch=16; a.img=(512,1024); a.cfg='yolov5s_city_seg.yaml'; a.dtype='f16'; a.stage='train'
dev=torch.device('cuda',0)
tr=bench.Trainer(a,1,0,dev)
for _ in range(5): tr.step()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(20): tr.step()
t1=time.perf_counter()
torch.cuda.synchronize()
t2=time.perf_counter()
print('host enqueue per step %.2f ms ; total per step %.2f ms' % ((t1-t0)/20*1e3, (t2-t0)/20*1e3))
# break down host time of phases
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for _ in range(5): tr.step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
# the native executor alone: one forward launch list, host time of the C call
from multiyolov5_amd import engine as E
plan=[h.plan for h in tr.model.__dict__['_plans'].values() if h.plan.training][0]
np_=plan._native_fwd()
if np_ is not None:
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): np_.run()
    t1=time.perf_counter(); torch.cuda.synchronize()
    print('native forward program: %d ops, host %.3f ms per run (%.2f us per op)' % (np_.n, (t1-t0)/10*1e3, (t1-t0)/10/np_.n*1e6))
