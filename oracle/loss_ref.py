"""CPU fp32 restatement of the reference's losses (TEST INFRASTRUCTURE ONLY).

ComputeLoss (utils/loss.py:89-217), bbox_iou CIoU (utils/general.py:343-385),
SegmentationLosses (loss.py:221-263), OhemCELoss (loss.py:303-328).
Autograd-differentiable so tests can compare gradients.
"""
import math

import torch
import torch.nn.functional as F


def scaled_hyp(imgsz=1024, nc=10, nl=3, label_smoothing=0.0):
    """data/hyp.scratch.yaml:14-21 after train.py:248-250 scaling."""
    return dict(box=0.05 * 3. / nl, cls=0.5 * nc / 80. * 3. / nl, obj=1.0 * (imgsz / 640) ** 2 * 3. / nl,
                cls_pw=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=label_smoothing)


def ciou_xywh(b1, b2, eps=1e-7):
    """general.py:343-380 with x1y1x2y2=False, CIoU=True.  b1,b2: [n,4] (cx,cy,w,h)."""
    b1x1, b1x2 = b1[:, 0] - b1[:, 2] / 2, b1[:, 0] + b1[:, 2] / 2
    b1y1, b1y2 = b1[:, 1] - b1[:, 3] / 2, b1[:, 1] + b1[:, 3] / 2
    b2x1, b2x2 = b2[:, 0] - b2[:, 2] / 2, b2[:, 0] + b2[:, 2] / 2
    b2y1, b2y2 = b2[:, 1] - b2[:, 3] / 2, b2[:, 1] + b2[:, 3] / 2
    inter = (torch.min(b1x2, b2x2) - torch.max(b1x1, b2x1)).clamp(0) * \
            (torch.min(b1y2, b2y2) - torch.max(b1y1, b2y1)).clamp(0)
    w1, h1 = b1x2 - b1x1, b1y2 - b1y1 + eps
    w2, h2 = b2x2 - b2x1, b2y2 - b2y1 + eps
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.max(b1x2, b2x2) - torch.min(b1x1, b2x1)
    ch = torch.max(b1y2, b2y2) - torch.min(b1y1, b2y1)
    c2 = cw ** 2 + ch ** 2 + eps
    rho2 = ((b2x1 + b2x2 - b1x1 - b1x2) ** 2 + (b2y1 + b2y2 - b1y1 - b1y2) ** 2) / 4
    v = (4 / math.pi ** 2) * (torch.atan(w2 / h2) - torch.atan(w1 / h1)) ** 2
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


def build_targets(p, targets, anchors, anchor_t=4.0):
    """loss.py:164-217.  Returns per level: (b, a, gj, gi, tbox[n,4], anch[n,2], tcls[n]) in the reference's
    row order (offset-major, then anchor, then target) -- that order decides the duplicate-cell winner."""
    na, nt = anchors.shape[1], targets.shape[0]
    out = []
    off = torch.tensor([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]], dtype=torch.float32, device=targets.device) * 0.5
    ai = torch.arange(na, dtype=torch.float32, device=targets.device).view(na, 1).repeat(1, nt)
    t7 = torch.cat((targets.repeat(na, 1, 1), ai[:, :, None]), 2)          # [na, nt, 7]
    for i, pi in enumerate(p):
        ny, nx = pi.shape[2], pi.shape[3]
        gain = torch.tensor([1, 1, nx, ny, nx, ny, 1], dtype=torch.float32, device=targets.device)
        t = t7 * gain
        if nt:
            r = t[:, :, 4:6] / anchors[i][:, None]
            keep = torch.max(r, 1. / r).max(2)[0] < anchor_t
            t = t[keep]
            gxy = t[:, 2:4]
            gxi = gain[[2, 3]] - gxy
            j, k = ((gxy % 1. < 0.5) & (gxy > 1.)).T
            l, m = ((gxi % 1. < 0.5) & (gxi > 1.)).T
            sel = torch.stack((torch.ones_like(j), j, k, l, m))
            t = t.repeat((5, 1, 1))[sel]
            offsets = (torch.zeros_like(gxy)[None] + off[:, None])[sel]
        else:
            t = t7[0]
            offsets = 0
        b, c = t[:, :2].long().T
        gxy, gwh = t[:, 2:4], t[:, 4:6]
        gij = (gxy - offsets).long()
        gi, gj = gij.T                # views of gij: the reference clamps them IN PLACE (loss.py:212), so tbox (213) sees
        a = t[:, 6].long()            # the clamped cell too
        gj.clamp_(0, ny - 1)          # loss.py:212 (int bounds: the torch>=1.10 compatibility fix)
        gi.clamp_(0, nx - 1)
        out.append((b, a, gj, gi, torch.cat((gxy - gij, gwh), 1), anchors[i][a], c))
    return out


def compute_loss(p, targets, anchors, hyp, gr=1.0, balance=(4.0, 1.0, 0.4)):
    """loss.py:115-162.  p: list of [B,na,ny,nx,5+nc]; anchors: [nl,na,2] in grid units.
    Returns (loss*bs, [lbox,lobj,lcls,loss])."""
    nc = p[0].shape[-1] - 5
    cp, cn = 1.0 - 0.5 * hyp.get('label_smoothing', 0.0), 0.5 * hyp.get('label_smoothing', 0.0)
    dev = targets.device
    lcls, lbox, lobj = torch.zeros(1, device=dev), torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    tg = build_targets(p, targets, anchors, hyp['anchor_t'])
    for i, pi in enumerate(p):
        b, a, gj, gi, tbox, anch, tcls = tg[i]
        tobj = torch.zeros_like(pi[..., 0])
        n = b.shape[0]
        if n:
            ps = pi[b, a, gj, gi]
            pxy = ps[:, :2].sigmoid() * 2. - 0.5
            pwh = (ps[:, 2:4].sigmoid() * 2) ** 2 * anch
            iou = ciou_xywh(torch.cat((pxy, pwh), 1), tbox)
            lbox = lbox + (1.0 - iou).mean()
            # loss.py:137 `tobj[b, a, gj, gi] = ...` with duplicate cells: index_put_ without accumulate is sequential on
            # one CPU thread (rows < the parallel grain), i.e. the LAST row of a cell wins.  Stated explicitly so the
            # oracle stays deterministic when torch parallelises large scatters.
            val = (1.0 - gr) + gr * iou.detach().clamp(0).type(tobj.dtype)
            lin = ((b * pi.shape[1] + a) * pi.shape[2] + gj) * pi.shape[3] + gi
            win = torch.full((tobj.numel(),), -1, dtype=torch.long, device=dev).scatter_reduce(0, lin, torch.arange(n, device=dev), 'amax')
            rows = win[win >= 0]
            tobj.view(-1)[lin[rows]] = val[rows]
            if nc > 1:
                t = torch.full_like(ps[:, 5:], cn)
                t[range(n), tcls] = cp
                lcls = lcls + F.binary_cross_entropy_with_logits(ps[:, 5:], t)
        lobj = lobj + F.binary_cross_entropy_with_logits(pi[..., 4], tobj) * balance[i]
    lbox = lbox * hyp['box']
    lobj = lobj * hyp['obj']
    lcls = lcls * hyp['cls']
    bs = p[0].shape[0]
    loss = lbox + lobj + lcls
    return loss * bs, torch.cat((lbox, lobj, lcls, loss)).detach()


def seg_ce(logits, target, ignore_index=-1):
    """SegmentationLosses default path (loss.py:236-237): mean CE over non-ignored pixels."""
    return F.cross_entropy(logits, target, ignore_index=ignore_index)


def seg_ce_aux(preds, target, aux_weight=0.1, ignore_index=-1):
    """SegmentationLosses aux_num==2 (loss.py:239-244): l1 + 0.15*l2 + 0.05*l3 for aux_weight 0.1."""
    l1, l2, l3 = (F.cross_entropy(q, target, ignore_index=ignore_index) for q in preds)
    return l1 + aux_weight * 1.5 * l2 + aux_weight / 2.0 * l3


def ohem_ce(logits, target, thresh=0.7, ignore_index=-1):
    """OhemCELoss.forward_once (loss.py:321-328): keep losses > -log(thresh); fall back to top n_min."""
    th = -math.log(thresh)
    n_min = int((target != ignore_index).sum().item() // 16)
    loss = F.cross_entropy(logits, target, ignore_index=ignore_index, reduction='none').view(-1)
    hard = loss[loss > th]
    if hard.numel() < n_min:
        hard, _ = loss.topk(n_min)
    return hard.mean()
