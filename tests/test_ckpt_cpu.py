"""checkpoint bridge (models/experimental.py): a checkpoint pickled by the REAL reference classes (tests/golden/ref_tiny_ckpt.pt,
written by oracle/make_ckpt_fixture.py in the layout of train.py:482-494) loads without the reference on sys.path."""
import os
import sys

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CKPT = os.path.join(GOLD, 'ref_tiny_ckpt.pt')


def test_reference_checkpoint_loads_as_shells_then_model():
    from multiyolov5_amd.models import experimental as X
    from multiyolov5_amd.models.yolo import Model
    assert 'models' not in sys.modules and 'models.yolo' not in sys.modules       # the reference is not importable here
    ck = X.load_checkpoint(CKPT, map_location='cpu')
    assert ck['epoch'] == 12 and ck['updates'] == 345
    assert isinstance(ck['model'], X._Shell) and isinstance(ck['ema'], X._Shell)
    g = np.load(os.path.join(GOLD, 'ref_tiny_ckpt.npz'))
    ema = X.model_from_reference(ck['ema'])
    assert isinstance(ema, Model)
    assert sum(p.numel() for p in ema.parameters()) == int(g['n_params'])
    first = next(iter(ema.state_dict().values())).reshape(-1)[:8]
    assert first.dtype == torch.float32 and np.array_equal(first.numpy(), g['ema_first'])      # fp16 -> fp32 is exact
    assert list(ema.state_dict()) == list(ck['ema'].state_dict())                              # the reference's keys, in order
    assert np.array_equal(ema.stride.numpy(), g['stride']) and ema.names[3] == 'class3' and ema.hyp['obj'] == 1.0


def test_attempt_load_semantics():
    """experimental.py:114-134: EMA preferred over model, fp32, fused, eval; a list gives an Ensemble carrying names / stride"""
    from multiyolov5_amd.models import experimental as X
    from multiyolov5_amd.models.common import Conv
    m = X.attempt_load(CKPT, map_location='cpu')
    assert not m.training and all(p.dtype == torch.float32 for p in m.parameters())
    assert all(not hasattr(c, 'bn') for c in m.modules() if type(c) is Conv)                   # fused
    ck = X.load_checkpoint(CKPT, map_location='cpu')
    det_key = [k for k in m.state_dict() if k.endswith('.m.0.weight')][0]                      # Detect conv: untouched by fuse()
    assert torch.equal(m.state_dict()[det_key], ck['ema'].state_dict()[det_key].float())
    assert not torch.equal(m.state_dict()[det_key], ck['model'].state_dict()[det_key].float())
    e = X.attempt_load([CKPT, CKPT], map_location='cpu')
    assert isinstance(e, X.Ensemble) and len(e) == 2 and e.names == m.names and torch.equal(e.stride, m.stride)


def test_strip_optimizer_roundtrip(tmp_path):
    from multiyolov5_amd.models import experimental as X
    from multiyolov5_amd.models.yolo import Model
    out = str(tmp_path / 'stripped.pt')
    X.strip_optimizer(CKPT, out)
    x = X.load_checkpoint(out, map_location='cpu')
    assert x['epoch'] == -1 and x['ema'] is None and x['optimizer'] is None
    assert isinstance(x['model'], Model) and all(p.dtype == torch.float16 and not p.requires_grad for p in x['model'].parameters())
    m = X.attempt_load(out, map_location='cpu')                                              # and it loads again
    assert isinstance(m, Model) and not m.training
