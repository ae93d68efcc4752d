"""Checkpoint bridge and model ensemble -- host-side mirror of the reference's models/experimental.py:98-134
(`Ensemble`, `attempt_load`) and of the checkpoint layout written by train.py:482-494 / utils/general.py:512-525.

A reference `.pt` is a pickle of whole `models.yolo.Model` objects (`{'epoch', 'model': Model.half(), 'ema', ...}`), i.e. it
names the reference's own classes (`models.common.Conv`, ...).  This library does not execute those classes, so the
bridge unpickles them as inert `nn.Module` shells (only `_parameters` / `_buffers` / `_modules` and plain attributes are
restored -- no reference code runs), takes what defines the network -- the parsed yaml dict (`Model.yaml`, yolo.py:237-244),
the `state_dict()`, `names`, `nc`, `hyp`, `gr` -- and builds a `multiyolov5_amd.models.yolo.Model` from it.  The weights stay
OIHW fp32 masters in the Parameters (checkpoint compatible both ways: `state_dict` keys are the reference's); the packed
MFMA layouts are produced per launch plan.
"""
import io
import pickle

import torch
import torch.nn as nn

from .yolo import Model

_REF_PREFIXES = ('models.', 'utils.')           # the reference's own top-level packages (models/, utils/)
_REF_MODULES = ('models', 'utils')


class _Shell(nn.Module):
    """inert stand-in for one reference class: holds the pickled module state, never runs."""

    def forward(self, *a, **k):  # pragma: no cover - shells are never executed
        raise RuntimeError('reference module shell: build a multiyolov5_amd Model from the checkpoint instead (attempt_load)')


_shells = {}


def _shell_class(module, name):
    key = (module, name)
    cls = _shells.get(key)
    if cls is None:
        cls = _shells[key] = type(name, (_Shell,), {'__module__': 'multiyolov5_amd.models.experimental', '_ref_class': f'{module}.{name}'})
    return cls


class _BridgeUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module in _REF_MODULES or module.startswith(_REF_PREFIXES):
            return _shell_class(module, name)
        return super().find_class(module, name)


class _bridge_pickle:
    """`pickle_module` for torch.load: stock pickle with the reference namespaces mapped to shells."""
    __name__ = 'multiyolov5_amd.models.experimental._bridge_pickle'
    Unpickler = _BridgeUnpickler
    load = staticmethod(lambda f, **kw: _BridgeUnpickler(f, **kw).load())
    loads = staticmethod(lambda b, **kw: _BridgeUnpickler(io.BytesIO(b), **kw).load())
    dumps, dump, Pickler = pickle.dumps, pickle.dump, pickle.Pickler
    HIGHEST_PROTOCOL, DEFAULT_PROTOCOL = pickle.HIGHEST_PROTOCOL, pickle.DEFAULT_PROTOCOL
    PickleError, UnpicklingError, PicklingError = pickle.PickleError, pickle.UnpicklingError, pickle.PicklingError


def load_checkpoint(path, map_location=None):
    """torch.load of a reference checkpoint without the reference on sys.path (its Model objects arrive as shells)."""
    return torch.load(path, map_location=map_location, pickle_module=_bridge_pickle, weights_only=False)


def model_from_reference(ref):
    """shell of a pickled reference `Model` -> multiyolov5_amd `Model` with the same yaml, weights, names, hyp."""
    if isinstance(ref, Model):
        return ref
    cfg = getattr(ref, 'yaml', None)
    if not isinstance(cfg, dict):
        raise ValueError('checkpoint model carries no parsed yaml dict (Model.yaml): not a multiyolov5 / YOLOv5 v4 Model pickle')
    cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}
    m = Model(cfg, ch=cfg.get('ch', 3), nc=cfg.get('nc'))
    sd = {k: (v.float() if v.is_floating_point() else v) for k, v in ref.state_dict().items()}
    m.load_state_dict(sd, strict=True)
    for k in ('names', 'nc', 'hyp', 'gr', 'class_weights'):
        if hasattr(ref, k):
            setattr(m, k, getattr(ref, k))
    return m


class Ensemble(nn.ModuleList):
    """experimental.py:98-111: NMS ensemble -- detections of every member concatenated along the anchor axis.

    The reference line `module(x, augment)[0]` predates this fork's `[det, seg]` model output: there `[0]` is Detect's
    `(z, raw)` tuple and `torch.cat` raises TypeError.  The intended tensor -- the decoded detections `z` [B,A,5+nc] -- is
    taken here (the upstream YOLOv5 behaviour the class was written for)."""

    def __init__(self):
        super().__init__()

    def forward(self, x, augment=False):
        y = []
        for module in self:
            det = module(x, augment)[0]
            # forward outputs alias plan-owned static buffers: a module listed twice would overwrite its first result
            y.append((det[0] if isinstance(det, (tuple, list)) else det).clone())
        y = torch.cat(y, 1)  # nms ensemble
        return y, None  # inference, train output


def attempt_load(weights, map_location=None):
    """experimental.py:114-134: one model (weights=a or [a]) or an Ensemble (weights=[a,b,...]); every member is the
    checkpoint's EMA (if present) or model, as fp32, fused, in eval mode.  No download is attempted (no network)."""
    model = Ensemble()
    for w in weights if isinstance(weights, list) else [weights]:
        ckpt = load_checkpoint(w, map_location=map_location)
        ref = ckpt['ema' if ckpt.get('ema') else 'model'] if isinstance(ckpt, dict) else ckpt
        m = model_from_reference(ref)
        if map_location is not None:
            m = m.to(map_location)
        model.append(m.float().fuse().eval())
    if len(model) == 1:
        return model[-1]
    print('Ensemble created with %s\n' % weights)
    for k in ['names', 'stride']:
        setattr(model, k, getattr(model[-1], k))
    return model


def strip_optimizer(f='best.pt', s=''):
    """utils/general.py:512-525 for checkpoints written by THIS library (a dict whose 'model'/'ema' are multiyolov5_amd Models):
    keep the EMA as the model, drop optimizer state, fp16 weights, requires_grad off."""
    x = load_checkpoint(f, map_location=torch.device('cpu'))
    if x.get('ema'):
        x['model'] = x['ema']
    for k in 'optimizer', 'training_results', 'wandb_id', 'ema', 'updates':
        x[k] = None
    x['epoch'] = -1
    if isinstance(x['model'], _Shell):
        x['model'] = model_from_reference(x['model'])
    x['model'].half()
    for p in x['model'].parameters():
        p.requires_grad = False
    torch.save(x, s or f)
