"""host-side cost of one detect.py frame (graph replay + NMS + argmax) at 1x3x512x1024: cProfile of 200 frames"""
import sys, time, torch, cProfile, pstats
sys.path.insert(0, '.')
import os
from multiyolov5_amd.models.yolo import Model
from multiyolov5_amd.utils.general import non_max_suppression, seg_argmax
from multiyolov5_amd import synth
dev = torch.device('cuda', 0)
H, W = 512, 1024
m = Model(os.path.join('multiyolov5_amd', 'cfg', 'yolov5s_city_seg.yaml'))
synth.randomize_(m, seed=0)
m = m.to(dev).half().fuse().eval()
img = synth.images(1, H, W, seed=7).to(dev, torch.float16)
na = 3 * ((H // 8) * (W // 8) + (H // 16) * (W // 16) + (H // 32) * (W // 32))
pred = synth.nms_pred(1, na, 10, seed=3, img_w=W, img_h=H).to(dev, torch.float16)
def frame():
    with torch.no_grad():
        out = m(img)
        det = non_max_suppression(pred, 0.25, 0.45)
        lab = seg_argmax(out[1], H, W)
for _ in range(10): frame()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): frame()
torch.cuda.synchronize()
print('wall per frame %.3f ms' % ((time.perf_counter() - t0) / 200 * 1e3))
ph = [0.0, 0.0, 0.0]
for _ in range(100):
    torch.cuda.synchronize(); t = time.perf_counter()
    with torch.no_grad():
        out = m(img)
    ph[0] += time.perf_counter() - t
    torch.cuda.synchronize(); t = time.perf_counter()
    det = non_max_suppression(pred, 0.25, 0.45)
    ph[1] += time.perf_counter() - t
    torch.cuda.synchronize(); t = time.perf_counter()
    lab = seg_argmax(out[1], H, W)
    ph[2] += time.perf_counter() - t
print('host per phase (ms; nms includes its own GPU time: it ends with a device->host read):', [round(p / 100 * 1e3, 3) for p in ph])
pr = cProfile.Profile(); pr.enable()
for _ in range(200): frame()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(16)
