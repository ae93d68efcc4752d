"""Host helpers mirrored from the reference's utils/torch_utils.py (fuse_conv_and_bn 182-202, initialize_weights
145-154, ModelEMA 270-304, time_synchronized 90-94, is_parallel, copy_attr, model_info 205-226)."""
import math
import time
from copy import deepcopy

import torch
import torch.nn as nn


def time_synchronized():  # torch_utils.py:90-94 (cuda.synchronize == hipDeviceSynchronize on ROCm)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.time()


def is_parallel(model):
    return type(model) in (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel)


def initialize_weights(model):  # torch_utils.py:145-154
    for m in model.modules():
        if type(m) is nn.BatchNorm2d:
            m.eps = 1e-3
            m.momentum = 0.03


def fuse_conv_and_bn(conv, bn):  # torch_utils.py:182-202 (one-off host transform at load time)
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, kernel_size=conv.kernel_size, stride=conv.stride,
                      padding=conv.padding, dilation=conv.dilation, groups=conv.groups, bias=True) \
        .requires_grad_(False).to(conv.weight.device, conv.weight.dtype)
    with torch.no_grad():
        scale = bn.weight.div(torch.sqrt(bn.eps + bn.running_var))
        fused.weight.copy_((scale.view(-1, 1) * conv.weight.view(conv.out_channels, -1)).view(fused.weight.shape))
        b_conv = torch.zeros(conv.weight.size(0), device=conv.weight.device, dtype=conv.weight.dtype) \
            if conv.bias is None else conv.bias
        b_bn = bn.bias - bn.weight.mul(bn.running_mean).div(torch.sqrt(bn.running_var + bn.eps))
        fused.bias.copy_(scale * b_conv + b_bn)
    return fused


def model_info(model, verbose=False, img_size=640):  # torch_utils.py:205-226 (FLOPs via thop omitted: not installed)
    n_p = sum(x.numel() for x in model.parameters())
    n_g = sum(x.numel() for x in model.parameters() if x.requires_grad)
    import logging
    logging.getLogger(__name__).info(f'Model Summary: {len(list(model.modules()))} layers, {n_p} parameters, {n_g} gradients')


def copy_attr(a, b, include=(), exclude=()):
    for k, v in b.__dict__.items():
        if (len(include) and k not in include) or k.startswith('_') or k in exclude:
            continue
        setattr(a, k, v)


class ModelEMA:
    """torch_utils.py:270-304.  `update` keeps the reference's semantics (every floating state_dict tensor, decay ramp);
    on the GPU the per-tensor python loop is replaced by one multi-tensor libmyolo launch (utils/optim.py)."""

    def __init__(self, model, decay=0.9999, updates=0):
        self.ema = deepcopy(model.module if is_parallel(model) else model).eval()
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / 2000))
        for p in self.ema.parameters():
            p.requires_grad_(False)

    def update(self, model):
        from .optim import ema_update
        self.updates += 1
        d = self.decay(self.updates)
        m = model.module if is_parallel(model) else model
        first = next(m.parameters(), None)
        stale = getattr(self, '_pairs_of', None) is not m or (first is not None and self._first_ptr != first.data_ptr())
        if stale:                                              # state_dict() walks ~360 modules: pair the tensors once
            msd, esd = m.state_dict(), self.ema.state_dict()
            self._pairs = [(esd[k], msd[k]) for k in esd if esd[k].dtype.is_floating_point]
            self._pairs_of = m
            self._first_ptr = first.data_ptr() if first is not None else 0
        ema_update(self._pairs, d)

    def update_attr(self, model, include=(), exclude=('process_group', 'reducer')):
        copy_attr(self.ema, model, include, exclude)
