"""shared helpers for the tests (CPU side)."""
import os

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, 'multiyolov5_amd', 'cfg')
GOLD = os.path.join(ROOT, 'tests', 'golden')
TAGS = {'s_psp': 'yolov5s_city_seg.yaml', 's_base': 'yolov5s_city_seg_base.yaml', 's_lab': 'yolov5s_city_seg_lab.yaml',
        's_bise': 'yolov5s_city_seg_bise.yaml', 'm_lab': 'yolov5m_city_seg_lab.yaml'}


def load_cfg(tag):
    with open(os.path.join(CFG, TAGS[tag])) as f:
        return yaml.safe_load(f)


def golden(name):
    return np.load(os.path.join(GOLD, name + '.npz'))


def tap(t, k=6):
    f = t.detach().float().reshape(-1).cpu()
    idx = torch.linspace(0, f.numel() - 1, k).long()
    return np.concatenate(([f.mean().item(), f.std().item() if f.numel() > 1 else 0.0, f.abs().max().item()],
                           f[idx].numpy())).astype(np.float32)


def synth_sd(tag, seed=0):
    from oracle import shapes, synth
    return synth.synth_state_dict(shapes.template_state_dict(load_cfg(tag)), seed)


class fp16_storage:
    """context manager: the oracle's convs/activations round their operands and results to fp16 (fp32 math in between),
    i.e. the storage points of the product's fp16 mode.  The deviation of this run from the fp32 oracle is the intrinsic
    fp16 noise of the (random-weight, tiny-batch BatchNorm) network; fp16 whole-model tests are judged against it."""

    def __enter__(self):
        import torch.nn.functional as F
        from oracle import model_ref
        self.mr, self.F = model_ref, F
        self.conv, self.silu = F.conv2d, F.silu
        q = lambda t: t.half().float()
        conv, silu = self.conv, self.silu

        class FQ:
            def __getattr__(_, name):
                return getattr(F, name)

            @staticmethod
            def conv2d(x, w, b=None, *a, **k):
                return q(conv(q(x), q(w), b, *a, **k))

            @staticmethod
            def silu(x):
                return q(silu(x))
        model_ref.F = FQ()
        return self

    def __exit__(self, *exc):
        self.mr.F = self.F
