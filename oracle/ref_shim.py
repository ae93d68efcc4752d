"""Import the real reference (pure Python, /root/reference) in the BUILD CONTAINER ONLY, to validate the
restatement and to generate tests/golden (TEST INFRASTRUCTURE ONLY; never imported on the GPU box).

Stubs the modules the reference imports unconditionally but that are not installed here:
cv2, onnx, torchvision, seaborn.  `torchvision.ops.nms` is served by oracle.nms_ref.greedy_nms.
"""
import os
import sys
import types

REF = os.environ.get('MYOLO_REFERENCE', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REF, 'models'))


def install():
    if not available():
        raise RuntimeError(f'reference tree not found at {REF}')
    import torch
    from . import nms_ref

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def _nms(boxes, scores, thr):
        keep = nms_ref.greedy_nms(boxes.detach().cpu().numpy(), scores.detach().cpu().numpy(), thr)
        return torch.from_numpy(keep).to(boxes.device)
    if 'cv2' not in sys.modules:
        stub('cv2', setNumThreads=lambda n: None)
    for n in ('onnx', 'onnx.external_data_helper', 'seaborn'):
        if n not in sys.modules:
            stub(n)
    if 'torchvision' not in sys.modules:
        tv = stub('torchvision')
        tv.ops = stub('torchvision.ops', nms=_nms)
        tv.transforms = stub('torchvision.transforms')
        tv.models = stub('torchvision.models')
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import models.yolo as ryolo          # noqa: E402  (the reference's package names)
    import utils.loss as rloss
    import utils.general as rgeneral
    import utils.torch_utils as rtu
    # compatibility patch (semantics-preserving): loss.py:212 clamp_ with float-tensor bounds
    # raises on torch>=1.10; re-bind build_targets with int bounds via a wrapper on Tensor.clamp_ is invasive,
    # so patch the two call sites' operands instead.
    _orig = torch.Tensor.clamp_

    def _clamp_(self, min=None, max=None):
        if torch.is_tensor(max) and not self.dtype.is_floating_point:
            max = int(max)
        if torch.is_tensor(min) and not self.dtype.is_floating_point:
            min = int(min)
        return _orig(self, min, max)
    torch.Tensor.clamp_ = _clamp_
    return types.SimpleNamespace(yolo=ryolo, loss=rloss, general=rgeneral, torch_utils=rtu, root=REF)
