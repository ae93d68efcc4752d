"""Write tests/golden/ref_tiny_ckpt.pt (+ .npz): a checkpoint PICKLED BY THE REAL REFERENCE CLASSES in the layout train.py:482-494
saves, for a narrow yolov5s_city_seg (width_multiple 0.125, 0.5 M parameters, fp16 -> ~1 MB), and the reference's own outputs
for it (`attempt_load` semantics: EMA weights, fp32, fused, eval) on a fixed image.  Build-container only (needs
/root/reference); the files it writes are committed and pin multiyolov5_amd.models.experimental.attempt_load.

    python -m oracle.make_ckpt_fixture
"""
import copy
import os
import sys

import numpy as np
import torch
import yaml

from . import ref_shim, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
CFG = os.path.join(ROOT, 'multiyolov5_amd', 'cfg')


def main():
    ref = ref_shim.install()
    with open(os.path.join(CFG, 'yolov5s_city_seg.yaml')) as f:
        cfg = yaml.safe_load(f)
    cfg['width_multiple'] = 0.125
    torch.manual_seed(0)
    m = ref.yolo.Model(copy.deepcopy(cfg))
    m.load_state_dict(synth.synth_state_dict({k: v.clone() for k, v in m.state_dict().items()}, seed=0), strict=True)
    m.names = [f'class{i}' for i in range(cfg['nc'])]
    m.nc, m.gr = cfg['nc'], 1.0
    m.hyp = {'box': 0.05, 'cls': 0.5, 'obj': 1.0}
    ema = copy.deepcopy(m)
    ema.load_state_dict(synth.synth_state_dict({k: v.clone() for k, v in m.state_dict().items()}, seed=7), strict=True)
    ckpt = {'epoch': 12, 'best_fitness': 0.5, 'training_results': 'n/a',
            'model': copy.deepcopy(m).half(), 'ema': copy.deepcopy(ema).half(), 'updates': 345,
            'optimizer': None, 'wandb_id': None}
    path = os.path.join(GOLD, 'ref_tiny_ckpt.pt')
    torch.save(ckpt, path)
    # what the reference itself computes from that file (experimental.py:114-134): EMA, fp32, fused, eval
    r = copy.deepcopy(ckpt['ema']).float().fuse().eval()
    x = synth.synth_images(1, 64, 128, seed=3)
    with torch.no_grad():
        det, seg = r(x)
    segs = seg if isinstance(seg, (list, tuple)) else [seg]
    np.savez_compressed(os.path.join(GOLD, 'ref_tiny_ckpt.npz'), z=det[0].numpy(), seg=segs[0].numpy()[:, :, ::4, ::4],
                        n_params=np.array(sum(p.numel() for p in m.parameters())), stride=m.stride.numpy(),
                        ema_first=next(iter(ckpt['ema'].float().state_dict().values())).reshape(-1)[:8].numpy())
    print('wrote', path, os.path.getsize(path), 'bytes;', 'params', sum(p.numel() for p in m.parameters()))


if __name__ == '__main__':
    sys.exit(main())
