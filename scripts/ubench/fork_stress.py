"""the detect.py frame loop (forward -> NMS -> resize + arg-max) at 64x128, NF frames over 12 different images, every frame's label map and
logits against the same model run without graphs / without a second stream: the stress that exposed round 6's kernel-made semaphore
(profiles/r6_fork_stress.txt) -- small tensors stay in the L2s, the head's adaptive-pool accumulators are not idempotent, and the
runtime's kernel-argument pool wraps every ~265 frames.  usage: [FORK=sem|event|joined] [NF=1200] python scripts/ubench/fork_stress.py [f16]"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from multiyolov5_amd.models.yolo import Model                               # noqa: E402
from multiyolov5_amd.utils.general import non_max_suppression, seg_argmax  # noqa: E402
from multiyolov5_amd import synth, runtime as R                            # noqa: E402

dev = torch.device('cuda:0')
H, W = 64, 128
half = len(sys.argv) > 1 and sys.argv[1] == 'f16'
NF = int(os.environ.get('NF', '1200'))


import copy
_raw = Model(os.path.join(ROOT, 'multiyolov5_amd', 'cfg', 'yolov5s_city_seg.yaml'))
synth.randomize_(_raw, seed=0)


def mk():
    m = copy.deepcopy(_raw).to(dev)                 # (the SAME parameters for both models: Model() initialises some at random)
    if half:
        m = m.half()
    return m.fuse().eval()


imgs = [synth.images(1, H, W, seed=s).to(dev) for s in range(12)]
if half:
    imgs = [x.half() for x in imgs]
na = 3 * ((H // 8) * (W // 8) + (H // 16) * (W // 16) + (H // 32) * (W // 32))
pred = synth.nms_pred(1, na, 10, seed=3, img_w=W, img_h=H).to(dev, torch.float16 if half else torch.float32)
R.GRAPH_EVAL = False
m = mk()
with torch.no_grad():
    ref = []
    for x in imgs:
        o = m(x)
        ref.append((seg_argmax(o[1], H, W).clone(), o[0][0].float().clone()))
torch.cuda.synchronize()
R.GRAPH_EVAL = True
FORK = os.environ.get('FORK', R.EVAL_FORK)
if FORK == 'joined':
    R.SPLIT_EVAL = False                             # one graph, the head joined inside it
else:
    R.EVAL_FORK = FORK
m2 = mk()
bad = 0
with torch.no_grad():
    for it in range(NF):
        i = it % len(imgs)
        out = m2(imgs[i])
        non_max_suppression(pred, 0.25, 0.45)
        lab = seg_argmax(out[1], H, W)
        nd = int((lab != ref[i][0]).sum())
        pd = float((out[0][0].float() - ref[i][1]).abs().max())
        if nd > 2 or pd > 1e-3:
            bad += 1
            if bad < 4:
                print('frame', it, 'label pixels that differ', nd, 'max |decoded prediction diff|', pd)
            if bad == 1 and os.environ.get('DUMP'):
                # which activations of THIS frame differ from the eager one-model run of the same image (every Buf of both plans, in allocation order)
                torch.cuda.synchronize()
                m(imgs[i])
                torch.cuda.synchronize()
                pa = [h.plan for h in m2.__dict__['_plans'].values()][0]
                pb = [h.plan for h in m.__dict__['_plans'].values()][0]
                owner = {}
                for k, op in enumerate(pa.ops):
                    o = getattr(op, 'out', None)
                    if o is not None and getattr(o, 'buf', None) is not None:
                        owner.setdefault(id(o.buf), []).append(f'{k}:{type(op).__name__}{"*" if getattr(op, "branch", None) else ""}')
                for k, (ba, bb) in enumerate(zip(pa.bufs, pb.bufs)):
                    if ba.t is None or bb.t is None:
                        continue
                    d = (ba.t.float() - bb.t.float()).abs()
                    d = torch.nan_to_num(d, nan=1e9)
                    if float(d.max()) > 1e-3:
                        print(f'   buf {k} {tuple(ba.t.shape)} max|diff| {float(d.max()):.4g} frac differing {float((d > 1e-3).float().mean()):.4f} nan {int(torch.isnan(ba.t.float()).sum())} written by {owner.get(id(ba), [])[:6]}')
hs = list(m2.__dict__.get('_plans', {}).values())
print('fork', FORK, 'f16' if half else 'f32', ': bad frames', bad, 'of', NF, '| un-joined head', [h.__dict__.get('_graph_c') is not None for h in hs],
      'semaphore', [(h.__dict__['_sem'][:33:32].tolist() if h.__dict__.get('_sem') is not None else None) for h in hs])
