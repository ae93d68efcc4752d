"""Losses of the joint det+seg step on MI355X -- host-side mirror of the reference's utils/loss.py
(ComputeLoss 89-217, SegmentationLosses 221-263, OhemCELoss 303-328, smooth_BCE 11-13).

Same class names, constructor signatures and call conventions; the arithmetic is in libmyolo (csrc/loss.hip): one
fused launch sequence per head with no host synchronisation (no boolean-mask indexing, no `.item()`), gradients are
produced by the library and handed to autograd through `torch.autograd.Function`.  There is no CPU path.
"""
import ctypes as C
import math
import os

import torch
import torch.nn as nn

from .. import _lib as L
from .torch_utils import is_parallel


# MYOLO_FUSED_CE=0: keep the CE backward a separate pass (needed only if the logits feed another differentiable op as well)
FUSED_CE = os.environ.get('MYOLO_FUSED_CE', '1') != '0'
# MYOLO_FUSED_UPCE=0: keep the loss on the materialised full-resolution logits (myolo_seg_ce_fwd_grad + myolo_seg_upsample_bwd)
FUSED_UPCE = os.environ.get('MYOLO_FUSED_UPCE', '1') != '0'


def smooth_BCE(eps=0.1):  # loss.py:11-13
    return 1.0 - 0.5 * eps, 0.5 * eps


def _check_logits(x):
    L.require_gpu(x)
    if x.dim() != 4 or x.dtype not in (torch.float16, torch.float32):
        raise L.MyoloError('segmentation logits must be a [N,C,H,W] fp16/fp32 tensor')
    if x.shape[1] > 32:
        raise L.MyoloError('at most 32 segmentation classes are supported by the fused CE kernel')


def _check_target(t, x):
    L.require_gpu(t)
    if t.dtype != torch.int64 or tuple(t.shape) != (x.shape[0], x.shape[2], x.shape[3]):
        raise L.MyoloError(f'segmentation target must be int64 [N,H,W] matching the logits, got {t.dtype} {tuple(t.shape)}')
    return t.contiguous()


class _SegCE(torch.autograd.Function):
    """mean CE over non-ignored pixels (ohem_thresh None) or OhemCELoss.forward_once (ohem_thresh = -log(p))."""
    _warned_classes = False

    @staticmethod
    def forward(ctx, logits, target, ignore_index, ohem_thresh):
        _check_logits(logits)
        target = _check_target(target, logits)
        lib, st = L.lib(), L.stream_ptr()
        n, c, h, w = logits.shape
        dev = logits.device
        acc = torch.empty(2, dtype=torch.float64, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        pix = sel = None
        if ohem_thresh is not None:
            pix = torch.empty(n * h * w, dtype=torch.float32, device=dev)
        # the producing plan's gradient buffer (saves a 2nd 300 MB tensor + copy); one loss per forward may claim it
        ctx.grad_buf = None
        if getattr(logits, '_myolo_grad_buf', None) is not None and not getattr(logits, '_myolo_grad_claimed', False):
            ctx.grad_buf = logits._myolo_grad_buf
            logits._myolo_grad_claimed = True
        # plain mean CE on the plan's own channels-last logits: the forward pass also leaves softmax - onehot in the plan's
        # gradient buffer and the backward only publishes the scalar gout/n_valid to the buffer's consumer
        gs = getattr(logits, '_myolo_grad_scale', None)
        ctx.fused = None
        # K15: the plan's low-resolution class logits are at hand -> upsample + CE + gradient + transposed upsample in one pass over
        # them; the full-resolution logits are not read and their gradient is never formed
        low, g32 = getattr(logits, '_myolo_low', None), getattr(logits, '_myolo_low_grad', None)
        if low is not None and g32 is not None and low.shape[3] != 19 and FUSED_UPCE and not _SegCE._warned_classes:
            import warnings
            _SegCE._warned_classes = True            # (the perf cliff should be visible: +~0.3 ms per step at 16x512x1024)
            warnings.warn(f'multiyolov5_amd: the one-pass upsample + cross-entropy kernels are built for 19 classes (Cityscapes); '
                          f'{low.shape[3]} classes take the materialised-logits path (x8 upsample + CE + its backward as separate passes)')
        if (FUSED_UPCE and FUSED_CE and pix is None and ctx.grad_buf is not None and gs is not None and ctx.needs_input_grad[0]
                and low is not None and g32 is not None and low.shape[3] == 19 and low.stride(3) == 1 and low.dtype == logits.dtype
                and not gs[1].get('low', True)):
            ld = L.Tensor(low.data_ptr(), low.shape[0], low.shape[1], low.shape[2], low.shape[3], low.stride(0), low.stride(1),
                          low.stride(2), L.DT[low.dtype], 0)
            rc = lib.myolo_seg_upce_fwd_grad(C.byref(ld), h, w, L.ptr(target), int(ignore_index), L.ptr(acc), L.ptr(loss),
                                             L.ptr(g32), st)
            if rc == 0:
                ctx.fused = gs
                gs[1]['low'] = True
            elif rc != L.EINVAL:
                L.check(rc, 'myolo_seg_upce_fwd_grad')
        # K15 for OhemCELoss (loss.py:303-328): per-pixel losses from the low-resolution logits (the only full-resolution tensor: fp32
        # [n,H,W]), hard-pixel selection on the device, then the selected pixels' gradient folded into the low-resolution map
        ctx.ohem_sel = None
        if (FUSED_UPCE and FUSED_CE and pix is not None and ctx.grad_buf is not None and gs is not None and ctx.needs_input_grad[0]
                and low is not None and g32 is not None and low.shape[3] == 19 and low.stride(3) == 1 and low.dtype == logits.dtype
                and not gs[1].get('low', True)):
            ld = L.Tensor(low.data_ptr(), low.shape[0], low.shape[1], low.shape[2], low.shape[3], low.stride(0), low.stride(1),
                          low.stride(2), L.DT[low.dtype], 0)
            rc = lib.myolo_seg_upce_ohem_pix(C.byref(ld), h, w, L.ptr(target), int(ignore_index), L.ptr(acc), L.ptr(pix), st)
            if rc == 0:
                sel = torch.empty(4, dtype=torch.float32, device=dev)
                scratch = torch.empty(5, dtype=torch.float64, device=dev)
                ws = torch.empty(2052, dtype=torch.int32, device=dev)
                L.check(lib.myolo_ohem_select(L.ptr(pix), n * h * w, C.c_float(ohem_thresh), L.ptr(acc), L.ptr(scratch), L.ptr(ws),
                                              L.ptr(loss), L.ptr(sel), st), 'myolo_ohem_select')
                L.check(lib.myolo_seg_upce_ohem_grad(C.byref(ld), h, w, L.ptr(target), int(ignore_index), L.ptr(pix), L.ptr(sel),
                                                     C.c_float(ohem_thresh), L.ptr(g32), st), 'myolo_seg_upce_ohem_grad')
                ctx.fused, ctx.ohem_sel = gs, sel
                gs[1]['low'] = True
                ctx.save_for_backward(logits, target, acc)
                ctx.pix, ctx.sel, ctx.ignore, ctx.thresh = None, None, int(ignore_index), ohem_thresh
                return loss.view(())
            elif rc != L.EINVAL:
                L.check(rc, 'myolo_seg_upce_ohem_pix')
        if (ctx.fused is None and FUSED_CE and pix is None and ctx.grad_buf is not None and gs is not None and ctx.needs_input_grad[0]
                and logits.stride() == (h * w * c, 1, w * c, c) and ctx.grad_buf.stride() == logits.stride()
                and ctx.grad_buf.dtype == logits.dtype and logits.data_ptr() % 16 == 0 and ctx.grad_buf.data_ptr() % 16 == 0):
            rc = lib.myolo_seg_ce_fwd_grad(L.ptr(logits), L.ptr(ctx.grad_buf), L.DT[logits.dtype], n, c, h, w, L.ptr(target),
                                           int(ignore_index), L.ptr(acc), L.ptr(loss), st)
            if rc == 0:
                ctx.fused = gs
            elif rc != L.EINVAL:                      # EINVAL = layout the fused kernel does not take: unfused path below
                L.check(rc, 'myolo_seg_ce_fwd_grad')
        if ctx.fused is None:
            L.check(lib.myolo_seg_ce_fwd(L.ptr(logits), L.DT[logits.dtype], n, c, h, w, *logits.stride(), L.ptr(target),
                                         int(ignore_index), L.ptr(acc), L.ptr(pix), None if pix is not None else L.ptr(loss), st),
                    'myolo_seg_ce_fwd')
        if pix is not None:
            sel = torch.empty(4, dtype=torch.float32, device=dev)
            scratch = torch.empty(5, dtype=torch.float64, device=dev)
            ws = torch.empty(2052, dtype=torch.int32, device=dev)
            L.check(lib.myolo_ohem_select(L.ptr(pix), n * h * w, C.c_float(ohem_thresh), L.ptr(acc), L.ptr(scratch), L.ptr(ws),
                                          L.ptr(loss), L.ptr(sel), st), 'myolo_ohem_select')
        ctx.save_for_backward(logits, target, acc)
        ctx.pix, ctx.sel, ctx.ignore, ctx.thresh = pix, sel, int(ignore_index), ohem_thresh
        return loss.view(())

    @staticmethod
    def backward(ctx, go):
        logits, target, acc = ctx.saved_tensors
        n, c, h, w = logits.shape
        grad = ctx.grad_buf
        if ctx.fused is not None:
            scale, state = ctx.fused
            gout = go.detach().to(torch.float32).reshape(1).contiguous()
            if ctx.ohem_sel is not None:            # OHEM: the mean runs over the selected pixels (sel = mode, denom, kth, tie weight)
                scale.copy_(gout / ctx.ohem_sel[1:2])
            else:
                L.check(L.lib().myolo_seg_ce_scale(L.ptr(acc), L.ptr(gout), L.ptr(scale), L.stream_ptr()), 'myolo_seg_ce_scale')
            state['fresh'] = True                   # consumed (and reset) by the plan's backward
            return grad, None, None, None
        if grad is None or grad.shape != logits.shape or grad.stride() != logits.stride() or grad.dtype != logits.dtype:
            grad = torch.empty_strided(logits.shape, logits.stride(), dtype=logits.dtype, device=logits.device)
        gout = go.detach().to(torch.float32).reshape(1).contiguous()
        L.check(L.lib().myolo_seg_ce_bwd(L.ptr(logits), L.ptr(grad), L.DT[logits.dtype], n, c, h, w, *logits.stride(),
                                         *grad.stride(), L.ptr(target), ctx.ignore, L.ptr(acc), L.ptr(gout), L.ptr(ctx.pix),
                                         L.ptr(ctx.sel), C.c_float(ctx.thresh if ctx.thresh is not None else 0.0),
                                         L.stream_ptr()), 'myolo_seg_ce_bwd')
        return grad, None, None, None


def seg_cross_entropy(logits, target, ignore_index=-1):
    return _SegCE.apply(logits, target, ignore_index, None)


class SegmentationLosses(nn.Module):
    """2D cross entropy with auxiliary losses (loss.py:221-263).  `weight` (class weights) and the se_loss branches are
    unused by the reference's training scripts (train.py:278-288) and are not on the gfx950 path."""

    def __init__(self, se_loss=False, se_weight=0.2, nclass=-1, aux_num=2, aux=False, aux_weight=0.1, weight=None,
                 ignore_index=-1):
        super().__init__()
        if se_loss or weight is not None:
            raise NotImplementedError('se_loss / class weights: not used by train.py (loss.py:246 "目前未使用") and not on the hot path')
        self.se_loss, self.aux, self.nclass, self.se_weight = se_loss, aux, nclass, se_weight
        self.aux_weight, self.aux_num, self.ignore_index = aux_weight, aux_num, ignore_index

    def _ce(self, pred, target):
        return _SegCE.apply(pred, target, self.ignore_index, None)

    def forward(self, *inputs):
        if not self.aux:                                        # Base / PSP / Lab (loss.py:236-237)
            return self._ce(*inputs)
        if self.aux_num == 2:                                   # BiSe (loss.py:239-244)
            pred1, pred2, pred3, target = tuple(inputs)
            return self._ce(pred1, target) + self.aux_weight * 1.5 * self._ce(pred2, target) + \
                self.aux_weight / 2.0 * self._ce(pred3, target)
        assert self.aux_num == 1                                # loss.py:246-250
        pred1, pred2, target = tuple(inputs)
        return self._ce(pred1, target) + self.aux_weight * self._ce(pred2, target)


class OhemCELoss(nn.Module):
    """OHEM cross entropy with auxiliary heads (loss.py:303-328).  `.cuda()` in the reference ctor (306) is implicit here:
    the threshold is a host float baked into the launch."""

    def __init__(self, thresh=0.5, ignore_index=-1, aux=False, aux_weight=[0.15, 0.05]):
        super().__init__()
        self.thresh = -torch.log(torch.tensor(thresh, requires_grad=False, dtype=torch.float))
        self._th = float(self.thresh)
        self.ignore_index, self.aux, self.aux_weight = ignore_index, aux, aux_weight

    def forward(self, preds, labels):
        if not self.aux:
            return self.forward_once(preds, labels)
        return self.forward_once(preds[0], labels) + self.aux_weight[0] * self.forward_once(preds[1], labels) + \
            self.aux_weight[1] * self.forward_once(preds[2], labels)

    def forward_once(self, preds, labels):
        return _SegCE.apply(preds, labels, self.ignore_index, self._th)


class _DetLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cl, targets, *p):
        lib, st = L.lib(), L.stream_ptr()
        p = [q.contiguous() for q in p]
        for q in p:
            L.require_gpu(q)
        dt = p[0].dtype
        if dt not in (torch.float16, torch.float32) or any(q.dtype != dt or q.dim() != 5 for q in p):
            raise L.MyoloError('ComputeLoss: predictions must be fp16/fp32 [B,na,ny,nx,no] tensors of one dtype')
        dev = p[0].device
        targets = targets.to(dev, torch.float32).contiguous()
        bs, na, no = p[0].shape[0], p[0].shape[1], p[0].shape[4]
        nt = targets.shape[0]
        if na != cl.na or no != cl.nc + 5 or len(p) != cl.nl:
            raise L.MyoloError('ComputeLoss: prediction shape does not match the Detect head')
        d = L.DetLossDesc()
        d.nl, d.na, d.no, d.bs, d.nt, d.dtype = cl.nl, na, no, bs, nt, L.DT[dt]
        ncell = 0
        for i, q in enumerate(p):
            d.p[i] = q.data_ptr()
            d.ny[i], d.nx[i] = q.shape[2], q.shape[3]
            d.balance[i] = cl.balance[i]
            ncell += bs * na * q.shape[2] * q.shape[3]
        anchors = cl.anchors.to(dev, torch.float32).contiguous()
        h = cl.hyp
        d.anchors, d.targets = anchors.data_ptr(), targets.data_ptr() if nt else None
        d.box, d.obj, d.cls, d.cls_pw, d.obj_pw = h['box'], h['obj'], h['cls'], h['cls_pw'], h['obj_pw']
        d.anchor_t, d.gr, d.cp, d.cn = h['anchor_t'], float(cl.gr), cl.cp, cl.cn
        winner = torch.empty(ncell, dtype=torch.int32, device=dev)
        ciou = torch.empty(max(cl.nl * 5 * na * nt, 1), dtype=torch.float32, device=dev)
        acc = torch.empty(20, dtype=torch.float64, device=dev)
        out = torch.empty(5, dtype=torch.float32, device=dev)
        d.winner, d.ciou, d.acc, d.out = winner.data_ptr(), ciou.data_ptr(), acc.data_ptr(), out.data_ptr()
        L.check(lib.myolo_detloss_fwd(C.byref(d), st), 'myolo_detloss_fwd')
        ctx.d, ctx.keep, ctx.p, ctx.ncell = d, (winner, ciou, acc, out, anchors, targets), p, ncell
        ctx.mark_non_differentiable(out)
        return out[0:1].clone(), out

    @staticmethod
    def backward(ctx, gloss, _gitems):
        d, p = ctx.d, ctx.p
        dev = p[0].device
        gout = gloss.detach().to(torch.float32).reshape(1).contiguous()
        gps = [torch.empty_like(q) for q in p]
        for i, g in enumerate(gps):
            d.gp[i] = g.data_ptr()
        gp32 = None
        if p[0].dtype == torch.float16:
            gp32 = torch.empty(ctx.ncell * d.no, dtype=torch.float32, device=dev)
        d.gp32, d.gout = L.ptr(gp32), gout.data_ptr()
        L.check(L.lib().myolo_detloss_bwd(C.byref(d), L.stream_ptr()), 'myolo_detloss_bwd')
        return (None, None, *gps)


class ComputeLoss:
    """loss.py:89-162.  `__call__(p, targets)` -> (loss * batch_size, cat(lbox, lobj, lcls, loss).detach())."""

    def __init__(self, model, autobalance=False):
        h = model.hyp
        self.cp, self.cn = smooth_BCE(eps=h.get('label_smoothing', 0.0))
        if h['fl_gamma'] > 0:
            raise NotImplementedError('FocalLoss (fl_gamma > 0) is off in hyp.scratch.yaml:21 and not on the gfx950 hot path')
        if autobalance:
            raise NotImplementedError('autobalance needs a host sync per level (loss.py:152); not on the hot path')
        det = model.module.model[-1] if is_parallel(model) else model.model[-1]
        self.balance = {3: [4.0, 1.0, 0.4]}.get(det.nl, [4.0, 1.0, 0.25, 0.06, .02])
        self.ssi = 0
        self.gr, self.hyp, self.autobalance = model.gr, h, autobalance
        for k in 'na', 'nc', 'nl', 'anchors':
            setattr(self, k, getattr(det, k))

    def __call__(self, p, targets):
        loss, items = _DetLoss.apply(self, targets, *p)
        return loss, items[1:5].detach()
