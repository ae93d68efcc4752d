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
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
