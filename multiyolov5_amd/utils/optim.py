"""Optimizer-side host code of the training step on MI355X: one multi-tensor libmyolo launch per operation instead of a
python loop of per-tensor ATen kernels.

* `FusedSGD`   -- `torch.optim.SGD(momentum, nesterov, weight_decay)` semantics over param groups (reference
  train.py:121-137 builds three groups; train.py:344-352 rewrites `lr` / `momentum` per group during warm-up)
* `GradScaler` -- `torch.cuda.amp.GradScaler` surface used by train.py:265,371,397-398 (`scale`, `step`, `update`): the
  unscale is folded into the SGD kernel, the inf/nan check and the growth/backoff bookkeeping stay on device
* `ema_update` -- `ModelEMA.update` body (utils/torch_utils.py:296-300)
"""
import ctypes as C
import os

import torch

from .. import _lib as L

CHUNK = 8192     # elements per workgroup (yolov5s: 1109 workgroups; scripts/optim_ubench.py)


class _Table:
    """device pointer table + chunk list for a fixed list of fp32 tensors triples; rebuilt only when a pointer moves."""

    def __init__(self, device):
        self.device = device
        self.key = None
        self.table = self.chunks = None
        self.nchunks = 0

    def update(self, ptrs, numels, groups):
        """ptrs: list of (p0, p1, p2) ints."""
        key = (tuple(ptrs), tuple(numels), tuple(groups))
        if key == self.key:
            return
        rows, ch = [], []
        for i, ((a, b, c), n, g) in enumerate(zip(ptrs, numels, groups)):
            rows.append((a, b, c, n, g, 0))
            for s in range(0, n, CHUNK):
                ch.append((i, s))
        self.table = torch.tensor(rows, dtype=torch.int64).reshape(-1, 6).to(self.device)
        self.chunks = torch.tensor(ch, dtype=torch.int32).reshape(-1, 2).to(self.device)
        self.nchunks = len(ch)
        self.key = key


def _f32_cuda(t, what):
    L.require_gpu(t)
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise L.MyoloError(f'{what}: contiguous fp32 tensors expected (got {t.dtype}, contiguous={t.is_contiguous()})')


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=0.01, momentum=0.0, dampening=0, weight_decay=0.0, nesterov=False):
        if dampening != 0:
            raise NotImplementedError('dampening != 0 is not used by the reference (train.py:133)')
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov)
        super().__init__(params, defaults)
        if len(self.param_groups) > 8:
            raise L.MyoloError('FusedSGD supports at most 8 param groups')
        self._tab = None

    def add_param_group(self, g):
        super().add_param_group(g)
        self._tab = None

    @torch.no_grad()
    def step(self, closure=None, scale=None, found_inf=None):
        if closure is not None:
            raise NotImplementedError('closure')
        ps, groups = [], []
        hy = L.SgdHyper()
        nesterov = None
        for gi, grp in enumerate(self.param_groups):
            hy.lr[gi], hy.momentum[gi], hy.weight_decay[gi] = grp['lr'], grp['momentum'], grp['weight_decay']
            nesterov = grp['nesterov'] if nesterov is None else nesterov
            if grp['nesterov'] != nesterov:
                raise L.MyoloError('FusedSGD: nesterov must be the same in every param group')
            for p in grp['params']:
                if p.grad is None:
                    continue
                ps.append(p)
                groups.append(gi)
        if not ps:
            return None
        hy.nesterov = int(bool(nesterov))
        dev = ps[0].device
        ptrs, numels = [], []
        for p in ps:
            st = self.state[p]
            buf = st.get('momentum_buffer')
            if buf is None:
                _f32_cuda(p, 'FusedSGD param')
                buf = st['momentum_buffer'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous():
                raise L.MyoloError('FusedSGD: gradients must be contiguous fp32')
            ptrs.append((p.data_ptr(), g.data_ptr(), buf.data_ptr()))
            numels.append(p.numel())
        if self._tab is None:
            self._tab = _Table(dev)
        self._tab.update(ptrs, numels, groups)
        t = self._tab
        lib, st = L.lib(), L.stream_ptr()
        if found_inf is not None:
            L.check(lib.myolo_mt_check_finite(L.ptr(t.table), L.ptr(t.chunks), t.nchunks, CHUNK, 1, L.ptr(found_inf), st),
                    'myolo_mt_check_finite')
        L.check(lib.myolo_mt_sgd(L.ptr(t.table), L.ptr(t.chunks), t.nchunks, CHUNK, C.byref(hy), L.ptr(scale), L.ptr(found_inf),
                                 st), 'myolo_mt_sgd')
        _bump_version(ps[0])                                 # (raw-pointer update: see ema_update)
        return None


class GradScaler:
    """loss scaling for fp16 training with the reference's call pattern (train.py:265 `amp.GradScaler(enabled=cuda)`)."""

    def __init__(self, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self._enabled = enabled
        self._init = float(init_scale)
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self._scale = self._tracker = self._found = None

    def _lazy(self, dev):
        if self._scale is None:
            self._scale = torch.full((1,), self._init, dtype=torch.float32, device=dev)
            self._tracker = torch.zeros(1, dtype=torch.int32, device=dev)
            self._found = torch.zeros(1, dtype=torch.float32, device=dev)

    def is_enabled(self):
        return self._enabled

    def scale(self, loss):
        if not self._enabled:
            return loss
        self._lazy(loss.device)
        return loss * self._scale.to(loss.dtype) if loss.dim() else loss * self._scale[0].to(loss.dtype)

    def step(self, optimizer, *args, **kwargs):
        if not self._enabled:
            return optimizer.step(*args, **kwargs)
        if not isinstance(optimizer, FusedSGD):
            raise NotImplementedError('multiyolov5_amd GradScaler drives FusedSGD; use torch.cuda.amp.GradScaler with torch optimizers')
        if self._scale is None:
            raise L.MyoloError('GradScaler.step() before scale()')
        return optimizer.step(*args, scale=self._scale, found_inf=self._found, **kwargs)

    def update(self, new_scale=None):
        if not self._enabled or self._scale is None:
            return
        if new_scale is not None:
            self._scale.fill_(float(new_scale))
            return
        L.check(L.lib().myolo_scaler_update(L.ptr(self._scale), L.ptr(self._tracker), L.ptr(self._found),
                                            C.c_float(self.growth_factor), C.c_float(self.backoff_factor),
                                            int(self.growth_interval), L.stream_ptr()), 'myolo_scaler_update')

    def get_scale(self):
        return float(self._scale) if (self._enabled and self._scale is not None) else 1.0

    def state_dict(self):
        return {'scale': self.get_scale(), 'growth_factor': self.growth_factor, 'backoff_factor': self.backoff_factor,
                'growth_interval': self.growth_interval,
                '_growth_tracker': int(self._tracker) if self._tracker is not None else 0} if self._enabled else {}


_EMA_TABLES = {}


@torch.no_grad()
def _bump_version(t):
    try:
        torch.autograd.graph.increment_version(t)
    except AttributeError:                                   # (older torch)
        t.add_(0)


@torch.no_grad()
def ema_update(pairs, d, model_sd=None):
    """v = d*v + (1-d)*m for every (ema tensor, model tensor) pair (floating state_dict entries, buffers included), one
    launch.  `pairs` may also be the EMA state_dict with `model_sd` the model's (torch_utils.py:296-300 call shape)."""
    if model_sd is not None:
        pairs = [(v, model_sd[k]) for k, v in pairs.items() if v.dtype.is_floating_point]
    if not pairs:
        return
    ptrs, numels = [], []
    for v, m in pairs:
        ptrs.append((v.data_ptr(), m.data_ptr(), 0))
        numels.append(v.numel())
    tab = _EMA_TABLES.get(ptrs[0][0])
    if tab is None:
        for v, m in pairs:
            _f32_cuda(v, 'ema_update')
            _f32_cuda(m, 'ema_update')
        tab = _EMA_TABLES[ptrs[0][0]] = _Table(pairs[0][0].device)
    tab.update(ptrs, numels, [0] * len(ptrs))
    L.check(L.lib().myolo_mt_ema(L.ptr(tab.table), L.ptr(tab.chunks), tab.nchunks, CHUNK, C.c_float(d), L.stream_ptr()),
            'myolo_mt_ema')
    # the kernel wrote through raw pointers: tell autograd's version counters (an eval plan of the EMA model -- test.py runs ema.ema
    # every epoch -- re-derives its packed weights / folded BatchNorm constants when the parameter versions move; one bump is enough
    # for runtime.PlannedModule._param_version, which sums them)
    _bump_version(pairs[0][0])
