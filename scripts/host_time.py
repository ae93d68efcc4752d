"""host-side cost of one training step, phase by phase.  The launch queue is drained (device synchronize) before every phase, so the times are
what the host needs to ENQUEUE the phase -- not how long it is blocked behind a full queue (a back-to-back loop measures the GPU's pace:
8.0 ms 'host' per 9.4 ms step in rounds 2-3 whatever the launch path was)."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
class A: pass
a = A(); a.batch = 16; a.img = (512, 1024); a.cfg = sys.argv[1] if len(sys.argv) > 1 else 'yolov5s_city_seg.yaml'; a.dtype = 'f16'; a.stage = 'train'; a.sync_bn = False; a.ddp = 'reducer'
dev = torch.device('cuda', 0)
tr = bench.Trainer(a, 1, 0, dev)
for _ in range(5):
    tr.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    tr.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('back-to-back: loop returns after %.2f ms per step (queue back-pressure included); GPU done after %.2f ms per step' % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
m, B = tr.model, a.batch
ph = {'forward': 0.0, 'losses': 0.0, 'backward': 0.0, 'optimizer+ema': 0.0}
N = 10
for _ in range(N):
    sync = torch.cuda.synchronize
    sync(); t = time.perf_counter()
    pred = m(tr.imgs)
    ph['forward'] += time.perf_counter() - t
    sync(); t = time.perf_counter()
    loss, items = tr.compute_loss(pred[0], tr.targets)
    segloss = tr.compute_seg_loss(pred[1], tr.mask) * B
    total = tr.scaler.scale(loss * 0.6 + segloss * 0.35)
    ph['losses'] += time.perf_counter() - t
    sync(); t = time.perf_counter()
    total.backward()
    ph['backward'] += time.perf_counter() - t
    sync(); t = time.perf_counter()
    tr.scaler.step(tr.opt); tr.scaler.update(); tr.opt.zero_grad(); tr.ema.update(m)
    ph['optimizer+ema'] += time.perf_counter() - t
sync()
print('host enqueue time per phase (ms):', {k: round(v / N * 1e3, 3) for k, v in ph.items()}, 'sum %.3f ms per step' % (sum(ph.values()) / N * 1e3))
from multiyolov5_amd import engine as E
plan = [h.plan for h in m.__dict__['_plans'].values() if h.plan.training][0]
np_ = plan._native_fwd()
if np_ is not None:
    sync(); t = time.perf_counter(); np_.run(); t1 = time.perf_counter(); sync()
    print('native forward program: %d ops, %.3f ms host (%.2f us per op)' % (np_.n, (t1 - t) * 1e3, (t1 - t) / np_.n * 1e6))
