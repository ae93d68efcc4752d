"""where the HOST spends a detect.py frame (bench.py --stage infer): duration of every hipGraph replay call, of the forward call as a whole, of NMS
(with its device->host read) and of the arg-max call, against the frame's wall time.  usage: python scripts/ubench/frame_host_timing.py [H W]"""
import os
import sys
import time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from multiyolov5_amd.models.yolo import Model                          # noqa: E402
from multiyolov5_amd.utils.general import non_max_suppression, seg_argmax  # noqa: E402
from multiyolov5_amd import synth                                      # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 2048)
dev = torch.device('cuda:0')
m = Model(os.path.join(ROOT, 'multiyolov5_amd', 'cfg', 'yolov5s_city_seg.yaml'))
synth.randomize_(m, seed=0)
m = m.to(dev).half().fuse().eval()
img = synth.images(1, H, W, seed=7).to(dev, torch.float16)
na = 3 * ((H // 8) * (W // 8) + (H // 16) * (W // 16) + (H // 32) * (W // 32))
pred = synth.nms_pred(1, na, 10, seed=3, img_w=W, img_h=H).to(dev, torch.float16)
acc = {}
_replay = torch.cuda.CUDAGraph.replay


def replay(self):
    t = time.perf_counter()
    _replay(self)
    acc.setdefault(id(self), []).append(time.perf_counter() - t)


def frame(rec=None):
    with torch.no_grad():
        t0 = time.perf_counter()
        out = m(img)
        t1 = time.perf_counter()
        non_max_suppression(pred, 0.25, 0.45)
        t2 = time.perf_counter()
        seg_argmax(out[1], H, W)
        t3 = time.perf_counter()
    if rec is not None:
        rec.append((t1 - t0, t2 - t1, t3 - t2))


for _ in range(20):
    frame()
torch.cuda.synchronize()
torch.cuda.CUDAGraph.replay = replay
rec = []
t0 = time.perf_counter()
for _ in range(200):
    frame(rec)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 200
print(f'{H}x{W}: frame wall {wall * 1e6:.1f} us; host: forward call {sum(r[0] for r in rec) / len(rec) * 1e6:.1f} us, NMS (incl. its read-back wait) '
      f'{sum(r[1] for r in rec) / len(rec) * 1e6:.1f} us, arg-max call {sum(r[2] for r in rec) / len(rec) * 1e6:.1f} us')
for i, (k, v) in enumerate(acc.items()):
    print(f'  graph {i}: replay() {sum(v) / len(v) * 1e6:.1f} us on the host, {len(v) / 200:.0f} per frame')
