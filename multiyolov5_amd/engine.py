"""Static launch plans over libmyolo (the host side of the MI355X hot path).

A reference module (`Conv`, `C3`, `SegMaskPSP`, ... `Model`) does not call ATen: its `emit()` appends kernel
launches to a `Plan` -- NHWC buffers laid out once (concat-free: producers write channel slices of the
consumer's buffer), weights pre-packed for MFMA, forward and backward launch lists fixed at build time.
Running a plan is a flat loop of C-ABI calls on the current HIP stream, so a whole training step is
hipGraph-capturable (torch.cuda.graph == hipGraph on ROCm).  There is no CPU path.
"""
import ctypes as C
import math
import os
import struct

import torch

from . import _lib as L
from ._lib import Tensor as CT

# MYOLO_GRAPH_TRAIN=1: training launch lists are replayed as hipGraphs instead of enqueued call by call (see Plan.run_fwd / run_bwd).
# Opt-in: the step is bound by the main stream'''s kernel time, not by the host (measured r2: eager 10.48 ms, graphs 10.59 ms per step)
GRAPH_TRAIN = os.environ.get('MYOLO_GRAPH_TRAIN', '0') != '0'
PACK_TILED = True            # (module attribute, no environment switch since round 6: measured 84 -> 62 us, round 3) per-forward weight repack through LDS tiles (coalesced OIHW reads)
LAZY_SEG = os.environ.get('MYOLO_LAZY_SEG', '1') != '0'            # training: materialise the x8-upsampled logits only on demand
LAZY_SEG_EVAL = True         # eval: same (detect.py's resize + argmax reads the low-resolution logits)
EVAL_BRANCH = True           # eval: the segmentation head runs on the side stream beside neck + Detect
BWD_SEGMENTS = 16
# 'seg': the backward is BWD_SEGMENTS pairs of single-stream graphs chained by events between launches; 'fork': ONE graph whose capture
# forks the weight-gradient stream per launch exactly like the eager loop (finer overlap; not used with a GradReducer: RCCL stays eager)
GRAPH_BWD = 'seg'            # ('fork': per-launch cross-stream forks inside one graph: 12.1 vs 10.5 ms, round 2)
WGRAD_WG = 0                 # 0: library default (128)
WGRAD_WG_TAIL = 0
WGRAD_TAIL_FRAC = 0.15
# round 4: the BatchNorm-backward apply pass of a 1x1 stride-1 Conv+BatchNorm layer rides in the operand path of the layer's own dgrad
# (myolo_conv_dgrad_bn, csrc/conv_mid.hip): one launch instead of two, dy still written for the weight gradient.  =0: the two-launch form
BN_APPLY_FOLD = os.environ.get('MYOLO_BN_APPLY_FOLD', '1') != '0'
# 1x1 Conv+BatchNorm+activation layers on maps of <= 1024 pixels (PyramidPooling's branches): one workgroup per layer, the layers of a
# module in ONE forward and ONE backward launch (csrc/tiny_conv.hip) instead of 2 + 3 launches per layer
CONV_PAIR = os.environ.get('MYOLO_CONV_PAIR', '1') != '0'       # eval: Bottleneck's 1x1 -> 3x3 in one launch (csrc/conv_pair.hip)
TINY_CONV = os.environ.get('MYOLO_TINY_CONV', '1') != '0'
# round 6: BatchNorm backward (reduce + apply) of a tensor the resident grid holds in registers as ONE launch with a device-wide barrier
# inside (csrc/bn_act.hip bn_act_bwd_fused_kernel).  1: layers that run the two passes as launches of their own; 2: also instead of the
# apply fold of a 1x1 dgrad (myolo_conv_dgrad_bn) where that layer's reduce pass is a launch of its own; 0: off
BN_BWD_FUSED = int(os.environ.get('MYOLO_BN_BWD_FUSED', '2'))
# round 6: training-mode Conv + BatchNorm + activation as ONE launch (myolo_conv_bn_act) for the layers whose tiles are all resident.  OPT-IN
# (MYOLO_CONV_BN_ACT=1): in the step it measures neutral (7.72-7.735 against 7.71-7.72 ms, profiles/r6_conv_bn_act_step_ab.txt) -- the barrier costs
# what the 6.4 us bn_act_fwd launch it replaces costs -- and a launch that needs every workgroup resident is the more fragile form
CONV_BN_ACT = os.environ.get('MYOLO_CONV_BN_ACT', '0') != '0'
# round 6: the first layer's BatchNorm backward + weight gradient as one pass over (gout, y, x) (myolo_bn_wgrad_stem)
STEM_WGRAD = os.environ.get('MYOLO_STEM_WGRAD', '1') != '0'

# MYOLO_NATIVE_EXEC=0: issue the launch lists one ctypes call at a time from Python (rounds 1-2) instead of through the native
# executor (csrc/plan_exec.hip: one C call per launch list)
NATIVE_EXEC = os.environ.get('MYOLO_NATIVE_EXEC', '1') != '0'
# MYOLO_PRUNE_BWD=0: a backward that starts from a subset of the outputs (train.py:371 / 392: the detection pass, the segmentation pass)
# runs the whole static launch list over zero gradients instead of the sub-list that can contribute (Plan.bwd_schedule)
PRUNE_BWD = os.environ.get('MYOLO_PRUNE_BWD', '1') != '0'

SEG = {torch.float16: 8, torch.float32: 4}
KC = {torch.float16: 32, torch.float32: 16}


# round 6: fp16 convolutions whose channel count is 32 modulo 64 (yolov5m's 96) get their packed weights zero-padded to the next multiple of
# 64 in K and in N, so that conv_mid's 128-byte K steps / 64-wide N tiles take the layer (ragged last chunk: the activation lanes past the
# tensor's channels fetch the zero page); 0: the round-5 padding (multiples of 32: conv_igemm's 96-wide tile / conv_stream run those layers)
MID_RAGGED = os.environ.get('MYOLO_MID_RAGGED', '1') != '0'
BN_STATS_IN_DGRAD = True            # fold bn_act_bwd_reduce into the dgrad that completes the layer's output gradient ...
BN_STATS_MAX_ELEMS = 4 << 20        # ... for maps up to this many elements (round 5 re-measured 9 M / 17 M: +0.2 / +1.1 %)


def barrier_launches_ok():
    """the one-launch forms with a device-wide barrier inside (myolo_conv_bn_act, myolo_bn_act_bwd_fused) need their whole grid resident.  With
    more than one rank RCCL's persistent collective kernels share the CUs during the backward (and their first call can sit in connection
    set-up for seconds): a barrier launch would wait for them, possibly past its spin bound.  Nothing with more than one GPU could be measured
    here, the forms are worth <= 0.3 % on one GPU -- so plans built under a process group of more than one rank use the two-launch forms."""
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)


def conv_pad(c, m, dt):
    """padded channel count of a packed weight operand (m: the kernel family's minimum granule)"""
    if MID_RAGGED and dt == torch.float16 and c >= 96 and c % 64 == 32:
        return rup(c, 64)
    return rup(c, m)


def rup(x, m):
    return (x + m - 1) // m * m


class Buf:
    """One NHWC allocation [n,h,w,c] (+ an optional gradient twin with the same geometry)."""

    def __init__(self, n, h, w, c, dtype):
        self.n, self.h, self.w, self.c, self.dtype = n, h, w, c, dtype
        self.t = None
        self.g = None
        self.gwritten = []      # list of (c0, c1) ranges of the gradient twin already written this backward
        self.gwriters = []      # (c0, c1, op | None) in backward execution order

    def alloc(self, device, with_grad):
        if self.t is None:
            self.t = torch.zeros(self.n, self.h, self.w, self.c, dtype=self.dtype, device=device)
        if with_grad and self.g is None:
            self.g = torch.zeros(self.n, self.h, self.w, self.c, dtype=self.dtype, device=device)


class TV:
    """Tensor view: channel slice [coff, coff+c) of a Buf.  `buf` may be unset until a concat claims it."""

    def __init__(self, plan, n, h, w, c, requires_grad=True):
        self.plan, self.n, self.h, self.w, self.c = plan, n, h, w, c
        self.buf, self.coff = None, 0
        self.requires_grad = requires_grad

    @property
    def shape(self):
        return (self.n, self.h, self.w, self.c)

    def place(self, buf, coff):
        assert self.buf is None
        self.buf, self.coff = buf, coff

    def desc(self, grad=False, c=None, pad_to=None):
        b = self.buf
        t = b.g if grad else b.t
        cc = self.c if c is None else c
        if pad_to:
            cc = min(rup(cc, pad_to), b.c - self.coff)
        es = t.element_size()
        return CT(t.data_ptr() + self.coff * es, b.n, b.h, b.w, cc, b.h * b.w * b.c, b.w * b.c, b.c,
                  L.DT[b.dtype], 0)

    def torch_view(self, grad=False):
        t = self.buf.g if grad else self.buf.t
        return t[..., self.coff:self.coff + self.c]


def null_tensor():
    return CT(None, 0, 0, 0, 0, 0, 0, 0, 0, 0)


class Call:
    """One C-ABI launch: function + argument tuple (stream appended at run time).  `side`: may run on the plan's side HIP
    stream (weight gradients: nothing in the backward chain depends on them)."""
    __slots__ = ('fn', 'args', 'name', 'keep', 'side')

    def __init__(self, name, args, keep=None, side=False):
        self.fn = getattr(L.lib(), name)
        self.name, self.args, self.keep, self.side = name, args, keep, side

    def __call__(self, st):
        e = self.fn(*self.args, st)
        if e:
            L.check(e, self.name)


class SwitchCall:
    """one of two launches, chosen when it is issued: `b` if state[key] (set by the consumer of the plan's output for this step) else
    `a`.  A captured backward graph bakes in the choice made at capture time."""
    __slots__ = ('a', 'b', 'state', 'key', 'name', 'side', 'args', 'keep', 'cell')

    def __init__(self, a, b, state, key):
        self.a, self.b, self.state, self.key = a, b, state, key
        self.name, self.side, self.args, self.keep = a.name, False, a.args, (a, b)
        self.cell = C.c_int32(0)          # the choice as the native executor reads it (refreshed from `state` before every run)

    def __call__(self, st):
        (self.b if self.state[self.key] else self.a)(st)


class SyncPoint:
    """a host-side step inside a launch list: all-reduce (SUM) of a plan-owned fp32 statistics array over the ranks of a process group,
    enqueued on the current stream between two launches (nn.SyncBatchNorm, train.py:190-193: the per-channel sums of the conv epilogue
    before BatchNorm's forward, the backward sums before its apply pass).  Behaves like a Call in the Python launch loops; the native
    executor's program is cut at it (NativeProg.run)."""
    __slots__ = ('t', 'group', 'name', 'side', 'args', 'keep')
    calls, host_s = 0, 0.0          # process-wide tally (bench.py --sync-bn reports collectives and their host cost per step)

    def __init__(self, t, group):
        self.t, self.group = t, group
        self.name, self.side, self.args, self.keep = 'sync_allreduce', False, (), t

    def __call__(self, st=None):
        import time
        import torch.distributed as dist
        t0 = time.perf_counter()
        dist.all_reduce(self.t, op=dist.ReduceOp.SUM, group=self.group)     # RCCL: stream-ordered behind / before the neighbouring launches
        SyncPoint.calls += 1
        SyncPoint.host_s += time.perf_counter() - t0


class Op:
    def build(self, plan):      # create Calls (buffers are allocated)
        self.fwd_calls, self.bwd_calls, self.prep_calls = [], [], []

    def plan_bwd(self, plan):   # decide accumulate flags; called in reverse op order before build()
        pass

    def grad_io(self):
        """(TVs whose gradient this op's backward launches READ, TVs whose gradient they WRITE) -- Plan.bwd_schedule's dataflow"""
        return [], []


def claim(tv, writer=None):
    """Gradient write planning for tv: returns (accumulate_flag, [zero-fill TV ranges needed first]).  `writer` (an Op whose dgrad
    launch does the write) is remembered per range: the LAST writer of a BatchNorm layer's output gradient can fold that layer's
    backward sums into its epilogue (Plan._plan_bn_stats)."""
    b = tv.buf
    lo, hi = tv.coff, tv.coff + tv.c
    b.gwriters.append((lo, hi, writer))
    log = getattr(tv.plan, '_claim_log', None)        # (claiming op, buffer, range, accumulate flag, zero-filled gaps) in backward order:
    covered = [(a, z) for a, z in b.gwritten if a < hi and z > lo]      # what Plan.bwd_schedule prunes a backward with
    if not covered:
        b.gwritten.append((lo, hi))
        if log is not None:
            log.append((tv.plan._claiming, b, lo, hi, 0, []))
        return 0, []
    # fully covered?
    pts = sorted(covered)
    cur, gaps = lo, []
    for a, z in pts:
        if a > cur:
            gaps.append((cur, min(a, hi)))
        cur = max(cur, z)
    if cur < hi:
        gaps.append((cur, hi))
    b.gwritten.append((lo, hi))
    if log is not None:
        log.append((tv.plan._claiming, b, lo, hi, 1, list(gaps)))
    return 1, gaps


def taps_fwd(k, d, pad):
    dy, dx, tw = [], [], []
    for kh in range(k):
        for kw in range(k):
            dy.append(kh * d - pad)
            dx.append(kw * d - pad)
            tw.append(kh * k + kw)
    return dy, dx, tw


def taps_dgrad(k, d, pad, s, py, px):
    """taps of dx[(s*oy'+py),(s*ox'+px)] = sum dy[oy'+off] W^T ; only taps with (p + pad - k*d) % s == 0."""
    dy, dx, tw = [], [], []
    for kh in range(k):
        ny = py + pad - kh * d
        if ny % s:
            continue
        for kw in range(k):
            nx = px + pad - kw * d
            if nx % s:
                continue
            dy.append(ny // s)
            dx.append(nx // s)
            tw.append(kh * k + kw)
    return dy, dx, tw


def fill_taps(desc, dy, dx, tw=None):
    for i in range(len(dy)):
        desc.tap_dy[i] = dy[i]
        desc.tap_dx[i] = dx[i]
        if tw is not None:
            desc.tap_w[i] = tw[i]


class FocusPackOp(Op):
    """common.py:550 -- space-to-depth of the NCHW image into NHWC16 (+cast)."""

    def __init__(self, plan, img_slot, out, mul=1.0):
        self.slot, self.out, self.mul = img_slot, out, mul

    def grad_io(self):
        return [self.out], []

    def build(self, plan):
        super().build(plan)
        meta = plan.in_meta[self.slot]
        n, _, h, w = meta['shape']
        self.d = self.out.desc()
        self.fwd_calls.append(Call('myolo_focus_pack', (plan.in_ptr[self.slot], L.DT[meta['dtype']], n, h, w,
                                                        C.c_float(self.mul), C.byref(self.d))))


class ImportOp(Op):
    """NCHW-logical torch tensor (any strides) -> NHWC TV (module-level boundary); bwd exports the gradient."""

    def __init__(self, plan, slot, out):
        self.slot, self.out = slot, out

    def grad_io(self):
        return [self.out], []

    def build(self, plan):
        super().build(plan)
        meta = plan.in_meta[self.slot]
        n, c, h, w = meta['shape']
        sn, sc, sh, sw = meta['stride']
        self.d = self.out.desc()
        self.fwd_calls.append(Call('myolo_seg_upsample_bwd', (plan.in_ptr[self.slot], L.DT[meta['dtype']], h, w, sn, sc, sh, sw,
                                                              C.byref(self.d), 0, None)))
        if plan.training and self.out.requires_grad:
            g = torch.empty(n, c, h, w, dtype=meta['dtype'], device=plan.device)
            plan.input_grads[self.slot] = g
            self.gd = self.out.desc(grad=True)
            gsn, gsc, gsh, gsw = g.stride()
            self.bwd_calls.append(Call('myolo_seg_upsample_fwd', (C.byref(self.gd), L.ptr(g), L.DT[g.dtype], h, w,
                                                                  gsn, gsc, gsh, gsw), keep=g))


class ConvOp(Op):
    """Conv2d (+BN) (+act) (+residual) / Detect conv; see include/myolo.h myolo_conv."""

    def __init__(self, plan, x, out, weight, bn=None, bias=None, k=1, s=1, d=1, act=L.ACT_NONE, res=None, det=None,
                 weight2=None, bn2=None, bias2=None):
        """weight2 / bn2 / bias2: a SECOND Conv(+BN) of the same geometry on the same input, run in the same launches; its output
        channels follow the first one's (`out` holds c1out + c2out channels; C3.cv2 | C3.cv1, common.py:137)."""
        # train.py:190-193 `--sync-bn`: convert_sync_batchnorm swaps the BatchNorm2d modules of the mirror for SyncBatchNorm.  With a process
        # group of more than one rank the per-channel sums are all-reduced between the producing launch and the BatchNorm pass
        # (SyncPoint) and the kernels count world x n*h*w samples (myolo_bn_split.count_scale); without one it IS BatchNorm2d, as in torch
        self.sync_world, self.sync_group = 1, None
        sync = [isinstance(b_, torch.nn.SyncBatchNorm) for b_ in (bn, bn2) if b_ is not None]
        if any(sync):
            import torch.distributed as dist
            if not all(sync) or (bn2 is not None and bn2.process_group is not bn.process_group):
                raise L.MyoloError('a merged convolution pair mixes BatchNorm2d and SyncBatchNorm (or two process groups)')
            if dist.is_available() and dist.is_initialized():
                self.sync_group = bn.process_group
                self.sync_world = dist.get_world_size(self.sync_group)
        self.x, self.out, self.weight, self.bn, self.bias = x, out, weight, bn, bias
        self.weight2, self.bn2, self.bias2 = weight2, bn2, bias2
        self.k, self.s, self.d, self.act, self.res, self.det = k, s, d, act, res, det
        self.pad = d * (k // 2)
        self.c1out = weight.shape[0]
        self.cout, self.cin = weight.shape[0] + (weight2.shape[0] if weight2 is not None else 0), weight.shape[1]
        if weight2 is not None:
            assert tuple(weight2.shape[1:]) == tuple(weight.shape[1:]) and (bn is None) == (bn2 is None) and \
                (bias is None) == (bias2 is None) and res is None and det is None and self.c1out % SEG[plan.dtype] == 0
        self.acc_x = 0
        self.zero_first = []
        self.res_acc = 0
        self.reduce_by = None        # the op whose dgrad launch produces this layer's BatchNorm-backward sums (else: own reduce launch)
        self.group = None            # tiny maps: the ConvOps sharing one myolo_tiny_conv_fwd / _bwd launch, forward order (Plan._plan_tiny_groups)
        self.bnb_targets = []        # [(layer op, c0, c1)]: BatchNorm layers whose output gradient this op's dgrad completes

    def grad_io(self):
        ins = [self.x] if self.x.requires_grad else []
        if self.res is not None and self.res.requires_grad:
            ins.append(self.res)
        return [self.out], ins

    def bn_reduce_call(self):
        """this layer's BatchNorm-backward reduce pass as its own launch (a pruned backward whose `reduce_by` dgrad does not run)"""
        bn = self.bn
        return Call('myolo_bn_act_bwd_reduce', (C.byref(self.god), C.byref(self.yd), L.ptr(self.saved), L.ptr(bn.weight), L.ptr(bn.bias),
                                                self.act, L.ptr(self.dsum)))

    def plan_bwd(self, plan):
        if self.res is not None and self.res.requires_grad:
            self.res_acc, z = claim(self.res)
            self.zero_first += [(self.res, a, b) for a, b in z]
        if self.x.requires_grad:
            self.acc_x, z = claim(self.x, writer=self)
            self.zero_first += [(self.x, a, b) for a, b in z]

    def tiny_ok(self, plan):
        """the conditions of myolo_tiny_conv_fwd / _bwd (include/myolo.h): a 1x1 Conv(+BatchNorm)(+activation) a single workgroup holds"""
        if not (TINY_CONV and plan.training) or self.k != 1 or self.s != 1 or self.res is not None or self.det or self.weight2 is not None:
            return False
        if self.bias is not None or self.sync_world > 1 or (self.bn is None and self.act == L.ACT_NONE):
            return False
        x, o, w = self.x, self.out, self.weight
        if (x.n, x.h, x.w) != (o.n, o.h, o.w) or x.c != self.cin or o.c != self.cout or w.dtype != torch.float32 or not w.is_contiguous():
            return False
        # pixel / channel limits and the workgroup's LDS budget: the library's own answer (csrc/tiny_conv.hip tiny_lds)
        return bool(L.lib().myolo_tiny_conv_ok(L.DT[plan.dtype], x.n * x.h * x.w, self.cin, self.cout))

    def _build_tiny(self, plan):
        """this layer's descriptor (forward AND backward fields); the group's LAST member (forward order) issues both launches: every
        member's input exists by then, and in the backward it comes first -- the consumers of every member's output ran before it"""
        dt, dev, g = plan.dtype, plan.device, self.group
        bn = self.bn
        self.yraw = Buf(self.out.n, self.out.h, self.out.w, self.cout, dt)
        self.yraw.alloc(dev, False)
        yv = TV(plan, *self.yraw.t.shape)
        yv.place(self.yraw, 0)
        self.yv = yv
        self.dy = Buf(self.out.n, self.out.h, self.out.w, self.cout, dt)
        self.dy.alloc(dev, False)
        dyv = TV(plan, *self.dy.t.shape)
        dyv.place(self.dy, 0)
        d = self.tdesc = L.TinyConvDesc()
        d.x, d.z, d.out, d.w = self.x.desc(), yv.desc(), self.out.desc(), self.weight.data_ptr()
        d.act, d.gx_accumulate = self.act, int(self.acc_x)
        if bn is not None:
            self.saved = torch.zeros(2 * self.cout, dtype=torch.float32, device=dev)
            d.gamma, d.beta, d.saved = bn.weight.data_ptr(), bn.bias.data_ptr(), self.saved.data_ptr()
            d.running_mean, d.running_var, d.nbt = bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr()
            d.eps, d.momentum = bn.eps, bn.momentum
            d.dgamma, d.dbeta = plan.pgrad(bn.weight).data_ptr(), plan.pgrad(bn.bias).data_ptr()
        d.gout, d.dy = self.out.desc(grad=True), dyv.desc()
        d.gx = self.x.desc(grad=True) if self.x.requires_grad else null_tensor()
        self.dy_desc = dyv.desc()
        calls = self.bwd_calls
        if self is g[-1]:
            self.tarr = (L.TinyConvDesc * len(g))(*[o.tdesc for o in g])
            self.fwd_calls.append(Call('myolo_tiny_conv_fwd', (self.tarr, len(g)), keep=[(o.bn, o.weight) for o in g]))
            for o in g:
                for tv, a, b in o.zero_first:
                    z = TV(plan, tv.n, tv.h, tv.w, b - a)
                    z.place(tv.buf, a)
                    zd = z.desc(grad=True)
                    calls.append(Call('myolo_fill_zero', (C.byref(zd),), keep=zd))
            calls.append(Call('myolo_tiny_conv_bwd', (self.tarr, len(g))))
        wd = L.WgradDesc()
        wd.x, wd.dy = self.x.desc(), self.dy_desc
        wd.dw, wd.db = plan.pgrad(self.weight).data_ptr(), None
        wd.ntaps, wd.stride, wd.up_shift, wd.ksplit, wd.cout, wd.cin = 1, 1, 0, 0, self.cout, self.cin
        ws = plan.wgrad_workspace()
        wd.ws, wd.ws_bytes = ws.data_ptr(), ws.numel() * 4
        fill_taps(wd, *taps_fwd(1, 1, 0)[:2])
        self.wds, self.wd = [wd], wd
        calls.append(Call('myolo_conv_wgrad', (C.byref(wd),), side=True))

    def build(self, plan):
        super().build(plan)
        if self.group:
            return self._build_tiny(plan)
        dt = plan.dtype
        seg, kc = SEG[dt], KC[dt]
        dev = plan.device
        training = plan.training
        has_bn = self.bn is not None
        two_pass = training and (has_bn or self.act != L.ACT_NONE)
        ntaps = self.k * self.k
        cin_pad, cout_pad = conv_pad(self.x.c, kc, dt), conv_pad(self.cout, 32, dt)
        self.wpack = torch.zeros(cout_pad, ntaps, cin_pad, dtype=dt, device=dev)
        w = self.weight
        w2 = self.weight2
        if training:      # repacked from the live fp32 master every forward (= autocast's cast): batched into one launch
            plan.add_pack_job(w, self.wpack, self.c1out, self.cin, ntaps, cout_pad, cin_pad, 0, w2)
        elif w2 is None:
            self.prep_calls.append(Call('myolo_pack_weight', (L.ptr(w), L.DT[w.dtype], self.cout, self.cin, self.k, self.k,
                                                              L.ptr(self.wpack), L.DT[dt], cout_pad, cin_pad, 0, None), keep=w))
        else:             # rows [0, c1out) from the first weight, the rest (+ zero padding) from the second
            self.prep_calls.append(Call('myolo_pack_weight', (L.ptr(w), L.DT[w.dtype], self.c1out, self.cin, self.k, self.k,
                                                              L.ptr(self.wpack), L.DT[dt], self.c1out, cin_pad, 0, None), keep=w))
            self.prep_calls.append(Call('myolo_pack_weight', (L.ptr(w2), L.DT[w2.dtype], self.cout - self.c1out, self.cin, self.k, self.k,
                                                              L.ptr(self.wpack[self.c1out:]), L.DT[dt], cout_pad - self.c1out, cin_pad,
                                                              0, None), keep=w2))
        d = L.ConvDesc()
        d.x = self.x.desc()
        d.w = self.wpack.data_ptr()
        d.cin_pad, d.cout_pad, d.wtaps, d.ntaps, d.stride, d.up_shift = cin_pad, cout_pad, ntaps, ntaps, self.s, 0
        fill_taps(d, *taps_fwd(self.k, self.d, self.pad))
        d.res = null_tensor()
        if two_pass:
            self.yraw = Buf(self.out.n, self.out.h, self.out.w, self.cout, dt)
            self.yraw.alloc(dev, False)
            yv = TV(plan, *self.yraw.t.shape)
            yv.place(self.yraw, 0)
            self.yv = yv
            d.y = yv.desc()
            d.act = L.ACT_NONE
            if has_bn:
                self.stats = plan.f32_fwd_zero(L.STAT_COPIES * 2 * self.cout)
                self.saved = torch.zeros(2 * self.cout, dtype=torch.float32, device=dev)
                d.stats = self.stats.data_ptr()
            self.yd, self.od = yv.desc(), self.out.desc()
            self.rd = self.res.desc() if self.res is not None else null_tensor()
            bn = self.bn
            # round 6 (north_star's "fused Conv+BN+SiLU", training mode): conv + batch statistics + device-wide barrier + BatchNorm +
            # activation (+ shortcut) in ONE launch where the library says every tile of the layer is resident (csrc/conv_mid.hip FUS)
            self.fwd_fused = bool(CONV_BN_ACT and barrier_launches_ok() and has_bn and dt == torch.float16 and self.sync_world == 1 and
                                  L.lib().myolo_conv_bn_act_ok(C.byref(d)))
            if self.fwd_fused:
                ff = self.ffuse = L.BnFwdFuse()
                ff.gamma, ff.beta, ff.running_mean, ff.running_var = bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                ff.nbt, ff.saved, ff.eps, ff.momentum, ff.act = bn.num_batches_tracked.data_ptr(), self.saved.data_ptr(), bn.eps, bn.momentum, self.act
                ff.res, ff.out, ff.barrier = self.rd, self.od, plan.grid_barrier().data_ptr()
                if self.bn2 is not None:
                    b2 = self.bn2
                    assert (b2.eps, b2.momentum) == (bn.eps, bn.momentum)
                    sp = self.split = L.BnSplit()
                    sp.c_split, sp.count_scale = self.c1out, 1
                    sp.gamma2, sp.beta2, sp.running_mean2 = b2.weight.data_ptr(), b2.bias.data_ptr(), b2.running_mean.data_ptr()
                    sp.running_var2, sp.nbt2 = b2.running_var.data_ptr(), b2.num_batches_tracked.data_ptr()
                    ff.split = C.pointer(sp)
                self.fwd_calls.append(Call('myolo_conv_bn_act', (C.byref(d), C.byref(ff)), keep=(bn, self.bn2)))
            else:
                self.fwd_calls.append(Call('myolo_conv', (C.byref(d),)))
            if self.fwd_fused:
                pass
            elif has_bn and (self.bn2 is not None or self.sync_world > 1):
                b2 = self.bn2
                sp = self.split = L.BnSplit()
                sp.c_split, sp.count_scale = self.cout, self.sync_world          # (c_split == c: no second parameter set)
                if b2 is not None:
                    assert (b2.eps, b2.momentum) == (bn.eps, bn.momentum)
                    sp.c_split = self.c1out
                    sp.gamma2, sp.beta2, sp.running_mean2 = b2.weight.data_ptr(), b2.bias.data_ptr(), b2.running_mean.data_ptr()
                    sp.running_var2, sp.nbt2 = b2.running_var.data_ptr(), b2.num_batches_tracked.data_ptr()
                if self.sync_world > 1:
                    self.fwd_calls.append(SyncPoint(self.stats, self.sync_group))
                self.fwd_calls.append(Call('myolo_bn_act_fwd_split', (
                    C.byref(self.yd), L.ptr(self.stats), L.ptr(bn.weight), L.ptr(bn.bias), L.ptr(bn.running_mean),
                    L.ptr(bn.running_var), L.ptr(bn.num_batches_tracked), L.ptr(self.saved), C.c_float(bn.eps),
                    C.c_float(bn.momentum), self.act, C.byref(self.rd), C.byref(self.od), C.byref(sp)), keep=(bn, b2)))
            elif has_bn:
                self.fwd_calls.append(Call('myolo_bn_act_fwd', (
                    C.byref(self.yd), L.ptr(self.stats), L.ptr(bn.weight), L.ptr(bn.bias), L.ptr(bn.running_mean),
                    L.ptr(bn.running_var), L.ptr(bn.num_batches_tracked), L.ptr(self.saved), C.c_float(bn.eps),
                    C.c_float(bn.momentum), self.act, C.byref(self.rd), C.byref(self.od)), keep=bn))
            else:
                self.fwd_calls.append(Call('myolo_bn_act_fwd', (
                    C.byref(self.yd), None, None, None, None, None, None, None, C.c_float(0), C.c_float(0), self.act,
                    C.byref(self.rd), C.byref(self.od))))
        else:
            d.y = self.out.desc()
            d.act = self.act
            if self.det:   # Detect: the forward result goes to a dense [N,na,ny,nx,no] tensor (yolo.py:214)
                o = self.out
                self.det_out = torch.zeros(o.n, self.det[0], o.h, o.w, self.det[1], dtype=dt, device=dev)
                d.y = CT(self.det_out.data_ptr(), o.n, o.h, o.w, self.cout, 0, 0, 0, L.DT[dt], 0)
                d.det_no = self.det[1]
            if self.res is not None:
                d.res = self.res.desc()
            if has_bn:       # eval: y = act(conv*scale + shift), scale/shift folded from running stats at prepare time
                self.scale = torch.empty(self.cout, dtype=torch.float32, device=dev)
                self.shift = torch.empty(self.cout, dtype=torch.float32, device=dev)
                d.scale, d.shift = self.scale.data_ptr(), self.shift.data_ptr()
            elif self.bias is not None:
                if self.bias.dtype == torch.float32 and self.bias2 is None:
                    d.shift = self.bias.data_ptr()        # read the live parameter: no stale copy under training
                else:
                    self.shift = torch.empty(self.cout, dtype=torch.float32, device=dev)
                    d.shift = self.shift.data_ptr()
            self.fwd_calls.append(Call('myolo_conv', (C.byref(d),)))
        self.fdesc = d
        # eval: a 1x1 layer whose output only this 3x3 layer reads (Bottleneck, models/common.py) runs inside this launch (csrc/conv_pair.hip)
        first = getattr(self, 'pair_first', None)
        if first is not None and CONV_PAIR and not training and len(self.fwd_calls) == 1 and len(first.fwd_calls) == 1 and \
                plan.sole_reader(first.out, first, self):      # (the intermediate is never written by the fused launch)
            first.fwd_calls.clear()
            self.fwd_calls[0] = Call('myolo_conv_pair', (C.byref(first.fdesc), C.byref(d)))
        if training:
            self._build_bwd(plan, two_pass, has_bn)

    def prepare(self):
        """(re)derive eval-mode epilogue constants from the parameters (host-side, once per weight version)."""
        with torch.no_grad():
            if hasattr(self, 'scale'):
                c0 = 0
                for bn in (self.bn, self.bn2):
                    if bn is None:
                        continue
                    sc = bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps)
                    c1 = c0 + sc.numel()
                    self.scale[c0:c1].copy_(sc)
                    self.shift[c0:c1].copy_(bn.bias.float() - bn.running_mean.float() * sc)
                    c0 = c1
            elif hasattr(self, 'shift'):
                self.shift[:self.c1out].copy_(self.bias.float())
                if self.bias2 is not None:
                    self.shift[self.c1out:].copy_(self.bias2.float())

    def _build_bwd(self, plan, two_pass, has_bn):
        dt, dev = plan.dtype, plan.device
        seg, kc = SEG[dt], KC[dt]
        calls = self.bwd_calls
        for tv, a, b in self.zero_first:
            z = TV(plan, tv.n, tv.h, tv.w, b - a)
            z.place(tv.buf, a)
            zd = z.desc(grad=True)
            calls.append(Call('myolo_fill_zero', (C.byref(zd),), keep=zd))
        if two_pass:
            self.dy = Buf(self.out.n, self.out.h, self.out.w, self.cout, dt)
            self.dy.alloc(dev, False)
            dyv = TV(plan, *self.dy.t.shape)
            dyv.place(self.dy, 0)
            self.god = self.out.desc(grad=True)
            self.dyd = dyv.desc()
            grd = self.res.desc(grad=True) if (self.res is not None and self.res.requires_grad) else null_tensor()
            self.grd = grd
            if has_bn:
                bn = self.bn
                self.dsum = plan.f32_bwd_zero(L.STAT_COPIES * 2 * self.cout)
                sync = [SyncPoint(self.dsum, self.sync_group)] if self.sync_world > 1 else []
                if self.bn2 is not None:
                    sp = self.split
                    sp.dgamma2, sp.dbeta2 = plan.pgrad(self.bn2.weight).data_ptr(), plan.pgrad(self.bn2.bias).data_ptr()
                fold_ok = (BN_APPLY_FOLD and dt == torch.float16 and self.k == 1 and self.s == 1 and self.bn2 is None and self.sync_world == 1 and
                           not grd.ptr and self.x.requires_grad and self.cout % 64 == 0 and self.cout <= 512 and self.out.c == self.cout)
                # one launch for both passes (round 6): the layer's sums are not produced by a dgrad epilogue, no collective sits between the
                # passes, every operand is a dense channel slice and the library says the tensor fits the resident grid
                own_reduce = self.bn2 is not None or self.reduce_by is None
                dense = all(t.sh == t.w * t.sw and t.sn == t.h * t.sh for t in (self.god, self.yd, self.dyd) + ((grd,) if grd.ptr else ()))
                self.bwd_fused = bool(BN_BWD_FUSED and barrier_launches_ok() and own_reduce and self.sync_world == 1 and dense and (BN_BWD_FUSED >= 2 or not fold_ok) and
                                      L.lib().myolo_bn_act_bwd_fused_ok(L.DT[dt], self.out.n * self.out.h * self.out.w, self.cout))
                if self.bwd_fused:
                    bar = plan.grid_barrier()
                    calls.append(Call('myolo_bn_act_bwd_fused', (
                        C.byref(self.god), C.byref(self.yd), L.ptr(self.saved), L.ptr(bn.weight), L.ptr(bn.bias), self.act, L.ptr(self.dsum),
                        L.ptr(plan.pgrad(bn.weight)), L.ptr(plan.pgrad(bn.bias)), C.byref(self.dyd), C.byref(grd), self.res_acc,
                        C.byref(self.split) if self.bn2 is not None else None, L.ptr(bar))))
                elif self.bn2 is not None:
                    calls.append(Call('myolo_bn_act_bwd_reduce_split', (C.byref(self.god), C.byref(self.yd), L.ptr(self.saved),
                                                                        L.ptr(bn.weight), L.ptr(bn.bias), self.act, L.ptr(self.dsum),
                                                                        C.byref(sp))))
                elif self.reduce_by is None:     # (else the dgrad that wrote the last piece of `gout` already left the sums in dsum)
                    calls.append(Call('myolo_bn_act_bwd_reduce', (C.byref(self.god), C.byref(self.yd), L.ptr(self.saved),
                                                                  L.ptr(bn.weight), L.ptr(bn.bias), self.act, L.ptr(self.dsum))))
                calls += sync                    # SyncBatchNorm: the sums of all ranks, before the apply pass reads them
                self.apply_fold = None
                if self.bwd_fused:
                    pass
                elif fold_ok:
                    f = L.BnApplyFold()
                    f.y, f.dy = self.yd, self.dyd
                    f.saved, f.gamma, f.beta, f.dsum = self.saved.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), self.dsum.data_ptr()
                    f.dgamma, f.dbeta, f.act = plan.pgrad(bn.weight).data_ptr(), plan.pgrad(bn.bias).data_ptr(), self.act
                    self.apply_fold = f                  # (the dgrad call below carries it: no apply launch of its own)
                elif self.bn2 is not None or self.sync_world > 1:
                    calls.append(Call('myolo_bn_act_bwd_apply_split', (
                        C.byref(self.god), C.byref(self.yd), L.ptr(self.saved), L.ptr(bn.weight), L.ptr(bn.bias), self.act,
                        L.ptr(self.dsum), L.ptr(plan.pgrad(bn.weight)), L.ptr(plan.pgrad(bn.bias)), C.byref(self.dyd),
                        C.byref(grd), self.res_acc, C.byref(self.split))))
                else:
                    calls.append(Call('myolo_bn_act_bwd_apply', (
                        C.byref(self.god), C.byref(self.yd), L.ptr(self.saved), L.ptr(bn.weight), L.ptr(bn.bias), self.act,
                        L.ptr(self.dsum), L.ptr(plan.pgrad(bn.weight)), L.ptr(plan.pgrad(bn.bias)), C.byref(self.dyd),
                        C.byref(grd), self.res_acc)))
            else:
                calls.append(Call('myolo_bn_act_bwd_apply', (
                    C.byref(self.god), C.byref(self.yd), None, None, None, self.act, None, None, None,
                    C.byref(self.dyd), C.byref(grd), self.res_acc)))
            dy_desc = dyv.desc()
        else:
            dy_desc = self.out.desc(grad=True, pad_to=seg)       # no BN, no act: dy is the output gradient itself
            if self.det:
                # the Detect output gradient arrives permuted [N,na,ny,nx,no]; un-permute into the NHWC twin first
                self.gdet = torch.zeros(self.out.n, self.det[0], self.out.h, self.out.w, self.det[1], dtype=dt, device=dev)
                plan.det_grads.append(self.gdet)
                self.gdd = self.out.desc(grad=True)
                calls.append(Call('myolo_detect_unpermute', (L.ptr(self.gdet), L.DT[dt], self.det[0], self.det[1],
                                                             C.byref(self.gdd))))
        self.dy_desc = dy_desc
        # dgrad
        if self.x.requires_grad:
            cin_pad_t, cout_pad_t = conv_pad(dy_desc.c, kc, dt), conv_pad(self.x.c, 32, dt)
            ntaps = self.k * self.k
            self.wpack_t = torch.zeros(cout_pad_t, ntaps, cin_pad_t, dtype=dt, device=dev)
            w = self.weight
            plan.add_pack_job(w, self.wpack_t, self.c1out, self.cin, ntaps, cout_pad_t, cin_pad_t, 1, self.weight2)
            self.dg = []
            par = []                                     # stride 2: the parity sub-convolutions, fused into one launch when all four exist
            s = self.s
            bnb = None
            if self.bnb_targets:                         # BatchNorm layers whose gout this dgrad completes: their reduce pass rides along
                bnb = (L.BnBwdSeg * len(self.bnb_targets))()
                for i, (lay, c0, c1) in enumerate(self.bnb_targets):
                    bnb[i].c0, bnb[i].c1, bnb[i].y = c0, c1, lay.yv.desc()
                    bnb[i].saved, bnb[i].gamma, bnb[i].beta = lay.saved.data_ptr(), lay.bn.weight.data_ptr(), lay.bn.bias.data_ptr()
                    bnb[i].dsum, bnb[i].act = lay.dsum.data_ptr(), lay.act
                self.bnb_arr = bnb
            gx_full = self.x.desc(grad=True)
            es = 2 if dt == torch.float16 else 4
            # (round 5, measured and removed: a stride-2 layer with a DENSE input gradient as two stride-1 convolutions over pixel pairs -- zero-slotted
            #  weight tensors, 2*cin contiguous output channels -- through conv_midx instead of the round-2 parity kernel: 7.682 vs 7.674 ms per step;
            #  those launches are HBM-bound either way (dy read twice: 268 MB for the stem's layer), profiles/r5d_threshold_sweep.txt)
            for py in range(s):
                for px in range(s):
                    tdy, tdx, tw = taps_dgrad(self.k, self.d, self.pad, s, py, px)
                    hh, ww = (self.x.h - py + s - 1) // s, (self.x.w - px + s - 1) // s
                    if hh <= 0 or ww <= 0:
                        continue
                    sub = CT(gx_full.ptr + (py * gx_full.sh + px * gx_full.sw) * es, gx_full.n, hh, ww, gx_full.c,
                             gx_full.sn, gx_full.sh * s, gx_full.sw * s, gx_full.dtype, 0)
                    if not tdy:
                        if not self.acc_x:
                            self.dg.append(sub)
                            calls.append(Call('myolo_fill_zero', (C.byref(sub),)))
                        continue
                    g = L.ConvDesc()
                    g.x, g.y, g.w = dy_desc, sub, self.wpack_t.data_ptr()
                    g.cin_pad, g.cout_pad, g.wtaps, g.ntaps, g.stride, g.up_shift = cin_pad_t, cout_pad_t, ntaps, len(tdy), 1, 0
                    fill_taps(g, tdy, tdx, tw)
                    g.act, g.accumulate, g.res = L.ACT_NONE, self.acc_x, null_tensor()
                    if bnb is not None and (s == 1 or not par):
                        g.nbnb, g.bnb = len(bnb), C.cast(bnb, C.POINTER(L.BnBwdSeg))
                    self.dg.append(g)
                    if s == 2:
                        par.append(g)
                    elif getattr(self, 'apply_fold', None) is not None:
                        g.x = self.god                   # the gradient w.r.t. the activation output: dy is formed inside the launch
                        calls.append(Call('myolo_conv_dgrad_bn', (C.byref(g), C.byref(self.apply_fold))))
                    else:
                        calls.append(Call('myolo_conv', (C.byref(g),)))
            if par:
                self.dg_arr = (C.POINTER(L.ConvDesc) * len(par))(*[C.pointer(g) for g in par])
                calls.append(Call('myolo_conv_dgrad_s2', (self.dg_arr, len(par))))
        # wgrad (one launch per weight tensor: the gradients live in separate slots of the flat gradient buffer)
        es = 2 if dt == torch.float16 else 4
        self.wds = []
        c0 = 0
        for w, b in ((self.weight, self.bias), (self.weight2, self.bias2)):
            if w is None:
                continue
            co = w.shape[0]
            wd = L.WgradDesc()
            wd.x = self.x.desc()
            wd.dy = dy_desc if self.weight2 is None else CT(dy_desc.ptr + c0 * es, dy_desc.n, dy_desc.h, dy_desc.w, co, dy_desc.sn,
                                                            dy_desc.sh, dy_desc.sw, dy_desc.dtype, 0)
            wd.dw = plan.pgrad(w).data_ptr()
            wd.db = plan.pgrad(b).data_ptr() if b is not None else None
            wd.ntaps, wd.stride, wd.up_shift, wd.ksplit, wd.cout, wd.cin = self.k * self.k, self.s, 0, 0, co, self.cin
            # weight gradients run beside the dgrad / BatchNorm chain: few long-lived workgroups (less CU / LDS stolen from the chain)
            # while plenty of the backward is still to come, many for the first layers of the network (= the END of the backward:
            # exposed tail)
            pos = getattr(plan, '_building', 0) / max(len(plan.ops), 1)
            wd.wg_hint = WGRAD_WG_TAIL if pos < WGRAD_TAIL_FRAC else WGRAD_WG
            ws = plan.wgrad_workspace()
            wd.ws, wd.ws_bytes = ws.data_ptr(), ws.numel() * 4
            tdy, tdx, _ = taps_fwd(self.k, self.d, self.pad)
            fill_taps(wd, tdy, tdx)
            self.wds.append(wd)
            calls.append(Call('myolo_conv_wgrad', (C.byref(wd),), side=True))
            c0 += co
        self.wd = self.wds[0]
        self._fuse_stem(plan, calls, two_pass, has_bn)

    def _fuse_stem(self, plan, calls, two_pass, has_bn):
        """round 6: a Conv + BatchNorm + SiLU layer WITHOUT an input gradient (the network's first layer) ends the backward with reduce -> apply ->
        weight gradient over the step's largest tensors; dy has one reader and is linear in sums one pass can form, so the three launches
        become `myolo_bn_wgrad_stem` (csrc/stem_wgrad.hip) on the MAIN stream (it is the critical tail; the side stream still holds the layer
        above's weight gradient)."""
        self.stem_fused = False
        if not (STEM_WGRAD and two_pass and has_bn and not self.x.requires_grad and self.bn2 is None and self.weight2 is None and
                self.sync_world == 1 and self.bias is None and self.act == L.ACT_SILU and plan.dtype == torch.float16 and
                self.reduce_by is None and not getattr(self, 'bwd_fused', False) and not self.grd.ptr and len(self.wds) == 1):
            return
        names = [getattr(c, 'name', None) for c in calls]
        want = ['myolo_bn_act_bwd_reduce', 'myolo_bn_act_bwd_apply', 'myolo_conv_wgrad']
        if names[-3:] != want or not L.lib().myolo_bn_wgrad_stem_ok(C.byref(self.wd), C.byref(self.god)):
            return
        bn = self.bn
        nbytes = int(L.lib().myolo_bn_wgrad_stem_ws_bytes())
        self.stem_ws = torch.empty(nbytes // 4, dtype=torch.float32, device=plan.device)      # (its own scratch: the side stream's weight gradients own the plan's)
        del calls[-3:]
        calls.append(Call('myolo_bn_wgrad_stem', (C.byref(self.wd), C.byref(self.god), C.byref(self.yd), L.ptr(self.saved), L.ptr(bn.weight), L.ptr(bn.bias),
                                                   self.act, L.ptr(plan.pgrad(bn.weight)), L.ptr(plan.pgrad(bn.bias)), L.ptr(self.stem_ws),
                                                   C.c_int64(nbytes))))
        self.stem_fused = True


class SimpleOp(Op):
    """Ops with one fwd launch and one bwd launch that (accumulate-)writes the gradient of `src`."""

    def __init__(self, plan, src, dst):
        self.src, self.dst = src, dst
        self.acc, self.zero_first = 0, []

    def grad_io(self):
        return [self.dst], ([self.src] if self.src.requires_grad else [])

    def plan_bwd(self, plan):
        if self.src.requires_grad:
            self.acc, z = claim(self.src)
            self.zero_first = [(self.src, a, b) for a, b in z]

    def _zero_calls(self, plan):
        out = []
        for tv, a, b in self.zero_first:
            z = TV(plan, tv.n, tv.h, tv.w, b - a)
            z.place(tv.buf, a)
            zd = z.desc(grad=True)
            out.append(Call('myolo_fill_zero', (C.byref(zd),), keep=zd))
        return out

    def build(self, plan):
        super().build(plan)
        self.sd, self.dd = self.src.desc(), self.dst.desc()
        self.emit_fwd(plan)
        if plan.training and self.src.requires_grad:
            self.bwd_calls += self._zero_calls(plan)
            self.gsd, self.gdd = self.src.desc(grad=True), self.dst.desc(grad=True)
            self.emit_bwd(plan)


class CopyUpOp(SimpleOp):
    def __init__(self, plan, src, dst, scale):
        super().__init__(plan, src, dst)
        self.scale = scale

    def emit_fwd(self, plan):
        self.fwd_calls.append(Call('myolo_copy_up_fwd', (C.byref(self.sd), C.byref(self.dd), self.scale)))

    def emit_bwd(self, plan):
        self.bwd_calls.append(Call('myolo_copy_up_bwd', (C.byref(self.gdd), C.byref(self.gsd), self.scale, self.acc)))


class BilinearOp(SimpleOp):
    """bilinear align_corners resize.  `group` (PyramidPooling's four upsamples, forward order): when their outputs are adjacent
    equal-width channel slices of one buffer, the four backward passes run as one pass over that buffer (myolo_pyramid_upsample_bwd)
    at the position of the group's LAST member (= the first of them in the backward; every consumer of the concat ran before)."""

    def __init__(self, plan, src, dst, group=None):
        super().__init__(plan, src, dst)
        self.group = group
        if group is not None:
            group.append(self)

    def _group_fused(self, plan, training=True):
        g = self.group
        if g is None or len(g) < 2 or len(g) > 4 :
            return False
        d0 = g[0].dst
        seg = SEG[plan.dtype]
        for i, o in enumerate(g):
            d, s_ = o.dst, o.src
            if d.buf is not d0.buf or d.c != d0.c or d.coff != d0.coff + i * d0.c or (d.h, d.w) != (d0.h, d0.w) or d.c % seg:
                return False
            if s_.h != s_.w or s_.h > 6 or s_.c != d.c or (training and not s_.requires_grad):
                return False
        return len(g) * d0.c // seg <= 16

    def emit_fwd(self, plan):
        if self._group_fused(plan, training=False):
            g = self.group
            if self is g[-1]:                     # the last source is ready: one launch writes the four adjacent slices
                d0 = g[0].dst
                allv = TV(plan, d0.n, d0.h, d0.w, d0.c * len(g))
                allv.place(d0.buf, d0.coff)
                self.fall = allv.desc()
                self.fxs = (CT * len(g))(*[o.src.desc() for o in g])
                self.fwd_calls.append(Call('myolo_pyramid_upsample_fwd', (self.fxs, len(g), C.byref(self.fall))))
            return
        self.fwd_calls.append(Call('myolo_bilinear_fwd', (C.byref(self.sd), C.byref(self.dd))))

    def emit_bwd(self, plan):
        if self._group_fused(plan):
            g = self.group
            if self is g[-1]:
                zero = []
                for o in g:
                    zero += o._zero_calls(plan)
                self.bwd_calls = zero
                d0 = g[0].dst
                allv = TV(plan, d0.n, d0.h, d0.w, d0.c * len(g))
                allv.place(d0.buf, d0.coff)
                self.gall = allv.desc(grad=True)
                self.gxs = (CT * len(g))(*[o.src.desc(grad=True) for o in g])
                self.gacc = (C.c_int32 * len(g))(*[int(o.acc) for o in g])
                scratch = plan.f32_bwd_zero(sum(o.src.n * o.src.h * o.src.w * o.src.c for o in g))
                self.bwd_calls.append(Call('myolo_pyramid_upsample_bwd', (C.byref(self.gall), self.gxs, len(g), self.gacc, L.ptr(scratch)),
                                           keep=scratch))
            else:
                self.bwd_calls = []
            return
        s_, d_ = self.src, self.dst
        scratch = None
        if s_.h <= 6 and s_.w <= 6 and (d_.h // max(s_.h, 1)) * (d_.w // max(s_.w, 1)) >= 64:      # PyramidPooling footprints
            scratch = plan.f32_bwd_zero(s_.n * s_.h * s_.w * s_.c)
        self.bwd_calls.append(Call('myolo_bilinear_bwd', (C.byref(self.gdd), C.byref(self.gsd), self.acc, L.ptr(scratch)),
                                   keep=scratch))


class AvgPoolOp(SimpleOp):
    """nn.AdaptiveAvgPool2d.  `group` (a list shared by the pools of one input, in forward order -- PyramidPooling): their backward
    passes run as ONE launch at the position of the group's first member (= the last of them in the backward), the input gradient is
    read-modified-written once."""

    def __init__(self, plan, src, dst, group=None):
        super().__init__(plan, src, dst)
        self.group = group
        if group is not None:
            group.append(self)

    def plan_bwd(self, plan):
        if self.group is None or len(self.group) < 2:
            return super().plan_bwd(plan)
        if self is self.group[-1] and self.src.requires_grad:        # first claimer in the backward: its flags stand for the group
            self.acc, z = claim(self.src)
            self.zero_first = [(self.src, a, b) for a, b in z]

    def _fwd_multi(self, plan):
        """PyramidPooling: the pools of one map as ONE pass over it (myolo_adaptive_avgpool_fwd_multi) -- the conditions of that entry point"""
        g = self.group if self.group is not None else [self]            # (a lone pool -- FFM's global average -- takes the same kernel)
        if len(g) > 4:
            return False
        s0, seg = g[0].src, SEG[plan.dtype]
        G = s0.c // seg
        if s0.c % seg or G < 1 or G > 256 or 256 % G:
            return False
        PL, kmax = 256 // G, max(o.dst.h for o in g)
        if any(o.src is not s0 or o.dst.h != o.dst.w or o.dst.c != s0.c for o in g):
            return False
        if (s0.w + PL - 1) // PL + 2 > s0.w // kmax or s0.w < PL:
            return False
        return sum(o.dst.h * o.dst.w for o in g) * s0.c * 4 <= 60 * 1024

    def emit_fwd(self, plan):
        d = self.dst
        if self._fwd_multi(plan):
            g = self.group if self.group is not None else [self]
            if self is g[0]:
                nb = sum(o.dst.h * o.dst.w for o in g)
                self.scratch = plan.f32_fwd_zero(8 * d.n * nb * self.src.c)
                self.fouts = (CT * len(g))(*[o.dst.desc() for o in g])
                self.fwd_calls.append(Call('myolo_adaptive_avgpool_fwd_multi', (C.byref(self.sd), self.fouts, len(g), L.ptr(self.scratch))))
            return
        self.scratch = plan.f32_fwd_zero(d.n * d.h * d.w * d.buf.c)
        self.fwd_calls.append(Call('myolo_adaptive_avgpool_fwd', (C.byref(self.sd), C.byref(self.dd), L.ptr(self.scratch))))

    def emit_bwd(self, plan):
        g = self.group
        if g is None or len(g) < 2:
            self.bwd_calls.append(Call('myolo_adaptive_avgpool_bwd', (C.byref(self.gdd), C.byref(self.gsd), self.acc)))
        elif self is g[0]:                                            # runs last in the backward: every pooled gradient is complete
            self.bwd_calls = [c for c in self.bwd_calls if c.name != 'myolo_fill_zero']
            self.bwd_calls += g[-1]._zero_calls(plan)
            self.garr = (CT * len(g))(*[o.dst.desc(grad=True) for o in g])
            self.bwd_calls.append(Call('myolo_adaptive_avgpool_bwd_multi', (self.garr, len(g), C.byref(self.gsd), g[-1].acc)))
        elif self is g[-1]:
            self.bwd_calls = []                                       # (its zero fills are issued by the group's launch)


class DropoutOp(SimpleOp):
    """train-mode nn.Dropout(p) (yolo.py:65,140)."""

    def __init__(self, plan, src, dst, p):
        super().__init__(plan, src, dst)
        self.p = p

    def emit_fwd(self, plan):
        s = self.src
        self.mask = torch.zeros(s.n * s.h * s.w * s.c, dtype=torch.uint8, device=plan.device)
        self.fwd_calls.append(Call('myolo_dropout_fwd', (C.byref(self.sd), C.byref(self.dd), L.ptr(self.mask), C.c_float(self.p),
                                                         L.ptr(plan.rng_counter()))))

    def emit_bwd(self, plan):
        self.bwd_calls.append(Call('myolo_dropout_bwd', (C.byref(self.gdd), L.ptr(self.mask), C.byref(self.gsd), C.c_float(self.p),
                                                         self.acc)))


class AddOp(Op):
    """out = a + b (BiSe `m16 + feat3`, yolo.py:84)."""

    def __init__(self, plan, a, b, out):
        self.a, self.b, self.out = a, b, out
        self.acc = [0, 0]
        self.zero_first = []

    def grad_io(self):
        return [self.out], [tv for tv in (self.a, self.b) if tv.requires_grad]

    def plan_bwd(self, plan):
        for i, tv in enumerate((self.a, self.b)):
            if tv.requires_grad:
                self.acc[i], z = claim(tv)
                self.zero_first += [(tv, x, y) for x, y in z]

    def build(self, plan):
        super().build(plan)
        self.ad, self.bd, self.od = self.a.desc(), self.b.desc(), self.out.desc()
        self.fwd_calls.append(Call('myolo_add', (C.byref(self.ad), C.byref(self.od), 0)))
        self.fwd_calls.append(Call('myolo_add', (C.byref(self.bd), C.byref(self.od), 1)))
        if plan.training:
            for tv, x, y in self.zero_first:
                z = TV(plan, tv.n, tv.h, tv.w, y - x)
                z.place(tv.buf, x)
                zd = z.desc(grad=True)
                self.bwd_calls.append(Call('myolo_fill_zero', (C.byref(zd),), keep=zd))
            self.god = self.out.desc(grad=True)
            self.gds = []
            for i, tv in enumerate((self.a, self.b)):
                if tv.requires_grad:
                    gd = tv.desc(grad=True)
                    self.gds.append(gd)
                    self.bwd_calls.append(Call('myolo_add', (C.byref(self.god), C.byref(gd), self.acc[i])))


class SppPoolOp(Op):
    """three stride-1 max pools of SPP (common.py:170); outputs are slices of the concat buffer."""

    def __init__(self, plan, x, outs):
        self.x, self.outs = x, outs
        self.acc, self.zero_first = 0, []

    def grad_io(self):
        return list(self.outs), ([self.x] if self.x.requires_grad else [])

    def plan_bwd(self, plan):
        if self.x.requires_grad:
            self.acc, z = claim(self.x)
            self.zero_first = [(self.x, a, b) for a, b in z]

    def build(self, plan):
        super().build(plan)
        self.xd = self.x.desc()
        self.ods = [o.desc() for o in self.outs]
        idx = None
        if plan.training:
            self.idx = torch.zeros(3 * self.x.n * self.x.h * self.x.w * self.x.c, dtype=torch.uint8, device=plan.device)
            idx = self.idx
        self.fwd_calls.append(Call('myolo_spp_pool_fwd', (C.byref(self.xd), C.byref(self.ods[0]), C.byref(self.ods[1]),
                                                          C.byref(self.ods[2]), L.ptr(idx))))
        if plan.training and self.x.requires_grad:
            for tv, a, b in self.zero_first:
                z = TV(plan, tv.n, tv.h, tv.w, b - a)
                z.place(tv.buf, a)
                zd = z.desc(grad=True)
                self.bwd_calls.append(Call('myolo_fill_zero', (C.byref(zd),), keep=zd))
            self.gds = [o.desc(grad=True) for o in self.outs]
            self.gxd = self.x.desc(grad=True)
            self.bwd_calls.append(Call('myolo_spp_pool_bwd', (C.byref(self.gds[0]), C.byref(self.gds[1]), C.byref(self.gds[2]),
                                                              L.ptr(self.idx), C.byref(self.gxd), self.acc)))


class GateOp(Op):
    """FFM: out = feat*att + feat (common.py:228-229); residual=False: ARM / Attention out = feat*att (common.py:192,207)."""

    def __init__(self, plan, feat, att, out, residual=True):
        self.feat, self.att, self.out = feat, att, out
        self.fn = 'myolo_gate' if residual else 'myolo_gate_mul'
        self.acc, self.zero_first, self.acc_att = 0, [], 0

    def grad_io(self):
        return [self.out], [self.feat, self.att]

    def plan_bwd(self, plan):
        self.acc, z = claim(self.feat)
        self.zero_first = [(self.feat, a, b) for a, b in z]
        self.acc_att, z2 = claim(self.att)
        assert not z2 and not self.acc_att, 'attention vector has a single consumer'

    def build(self, plan):
        super().build(plan)
        self.fd, self.ad, self.od = self.feat.desc(), self.att.desc(), self.out.desc()
        self.fwd_calls.append(Call(self.fn + '_fwd', (C.byref(self.fd), C.byref(self.ad), C.byref(self.od))))
        if plan.training:
            for tv, a, b in self.zero_first:
                z = TV(plan, tv.n, tv.h, tv.w, b - a)
                z.place(tv.buf, a)
                zd = z.desc(grad=True)
                self.bwd_calls.append(Call('myolo_fill_zero', (C.byref(zd),), keep=zd))
            self.gatt = plan.f32_bwd_zero(self.att.n * self.att.c)
            self.god, self.gfd, self.gad = self.out.desc(grad=True), self.feat.desc(grad=True), self.att.desc(grad=True)
            self.bwd_calls.append(Call(self.fn + '_bwd', (C.byref(self.god), C.byref(self.fd), C.byref(self.ad),
                                                          C.byref(self.gfd), self.acc, L.ptr(self.gatt))))
            self.bwd_calls.append(Call('myolo_cast_from_f32', (L.ptr(self.gatt), C.byref(self.gad))))


class SegOutOp(Op):
    """final bilinear x`scale` of the class logits into a [N,C,H,W]-logical output tensor (yolo.py:163 etc.)."""

    def __init__(self, plan, low, scale, slot):
        self.low, self.scale, self.slot = low, scale, slot
        self.acc = 0

    def grad_io(self):
        return [], [self.low]

    def plan_bwd(self, plan):
        self.acc, z = claim(self.low)
        assert not z

    def build(self, plan):
        super().build(plan)
        lw = self.low
        H, W = lw.h * self.scale, lw.w * self.scale
        store = torch.empty(lw.n, H, W, lw.c, dtype=plan.dtype, device=plan.device)
        out = store.permute(0, 3, 1, 2)                 # [N,C,H,W] logical, NHWC memory
        plan.outputs[self.slot] = out
        out._myolo_low = lw.torch_view()                # low-res class logits: utils.general.seg_argmax fuses resize+argmax on them
        self.ld = lw.desc()
        sn, sc, sh, sw = out.stride()
        up = Call('myolo_seg_upsample_fwd', (C.byref(self.ld), L.ptr(out), L.DT[out.dtype], H, W, sn, sc, sh, sw), keep=out)
        # training: the full-resolution logits (318 MB at 16x19x512x1024) are consumed by the fused loss from the LOW-resolution map
        # (K15) and nothing reads them -- the upsample is deferred until something other than that loss touches the tensor
        # (runtime.LazySegLogits).  Captured training graphs keep the eager launch.
        # eval (detect.py:191-193): utils.general.seg_argmax takes resize + argmax from the low-resolution logits, the 80 MB a 1024x2048
        # frame's upsample writes are read by nobody (25 us of a 1.19 ms forward); any other consumer triggers the launch as in training
        lazy = LAZY_SEG and ((plan.training and not GRAPH_TRAIN) or (not plan.training and LAZY_SEG_EVAL))
        self.lazy_call = up if lazy else None
        if self.lazy_call is None:
            self.fwd_calls.append(up)
        else:
            out._myolo_lazy = self
        if plan.training:
            g = torch.zeros_like(store).permute(0, 3, 1, 2)
            plan.output_grads[self.slot] = g
            out._myolo_grad_buf = g                     # utils.loss writes d(loss)/d(logits) straight into the plan's buffer
            # device scalar the incoming gradient is multiplied by (1 unless the fused CE left an unnormalised gradient in g;
            # 'fresh' = set by this step's loss backward, 'dirty' = still holds an old factor) -- see runtime.PlanFn.backward
            self.gscale = torch.ones(1, dtype=torch.float32, device=plan.device)
            self.gstate = {'fresh': False, 'dirty': False}
            out._myolo_grad_scale = (self.gscale, self.gstate)
            plan.output_scales[self.slot] = (self.gscale, self.gstate)
            self.gld = lw.desc(grad=True)
            full = Call('myolo_seg_upsample_bwd', (L.ptr(g), L.DT[g.dtype], H, W, *g.stride(), C.byref(self.gld),
                                                   self.acc, L.ptr(self.gscale)), keep=(g, self.gscale))
            # K15: utils.loss can run upsample + CE + both backwards in one pass over the low-res logits (myolo_seg_upce_fwd_grad); it
            # leaves the classifier's unnormalised fp32 gradient in glow32 and sets gstate['low'] -- the backward then only rescales it
            self.glow32 = torch.zeros(lw.n, lw.h, lw.w, lw.c, dtype=torch.float32, device=plan.device)
            if not GRAPH_TRAIN:          # a captured backward bakes in ONE branch of the switch below (the full-resolution one, chosen
                out._myolo_low_grad = self.glow32    # at capture time): the fused low-resolution loss is not offered under MYOLO_GRAPH_TRAIN
            self.gstate['low'] = False
            low = Call('myolo_seg_lowgrad_apply', (L.ptr(self.glow32), C.byref(self.gld), self.acc, L.ptr(self.gscale)),
                       keep=(self.glow32, self.gscale))
            self.bwd_calls.append(SwitchCall(full, low, self.gstate, 'low'))


class ExportOp(Op):
    """NHWC TV -> NCHW-logical output tensor (module-level boundary)."""

    def __init__(self, plan, src, slot):
        self.src, self.slot, self.acc = src, slot, 0

    def grad_io(self):
        return [], [self.src]

    def plan_bwd(self, plan):
        self.acc, z = claim(self.src)
        assert not z

    def build(self, plan):
        super().build(plan)
        s = self.src
        store = torch.empty(s.n, s.h, s.w, s.c, dtype=plan.dtype, device=plan.device)
        out = store.permute(0, 3, 1, 2)
        plan.outputs[self.slot] = out
        self.sd = s.desc()
        self.fwd_calls.append(Call('myolo_seg_upsample_fwd', (C.byref(self.sd), L.ptr(out), L.DT[out.dtype], s.h, s.w, *out.stride()),
                                   keep=out))
        if plan.training:
            g = torch.zeros_like(store).permute(0, 3, 1, 2)
            plan.output_grads[self.slot] = g
            self.gsd = s.desc(grad=True)
            self.bwd_calls.append(Call('myolo_seg_upsample_bwd', (L.ptr(g), L.DT[g.dtype], s.h, s.w, *g.stride(), C.byref(self.gsd),
                                                                  self.acc, None), keep=g))



def _slot_value(arg, argtype):
    """one ctypes call argument as the 8-byte slot of a myolo_prog_op (include/myolo.h)"""
    if arg is None:
        return 0
    if argtype is C.c_float:
        v = arg.value if isinstance(arg, C.c_float) else float(arg)
        return struct.unpack('<I', struct.pack('<f', v))[0]
    if argtype in (C.c_int32, C.c_int64, C.c_uint64, C.c_int, C.c_long):
        return int(arg.value if hasattr(arg, 'value') else arg) & 0xFFFFFFFFFFFFFFFF
    if isinstance(arg, C.c_void_p):
        return arg.value or 0
    if hasattr(arg, '_obj'):                         # C.byref(x)
        return C.addressof(arg._obj)
    if isinstance(arg, (C.Array, C.Structure)):
        return C.addressof(arg)
    if isinstance(arg, C._Pointer):
        return C.cast(arg, C.c_void_p).value or 0
    if isinstance(arg, int):
        return arg & 0xFFFFFFFFFFFFFFFF
    raise TypeError(f'cannot serialise launch argument {arg!r}')


class NativeProg:
    """a launch list serialised for csrc/plan_exec.hip.  `items`: Call | SwitchCall | ('memset', tensor, nbytes) | ('join',) | ('mark',
    key) -- a mark records the op index reached (range boundaries for the caller)."""

    def __init__(self, items, mutable_cells=()):
        lib = L.lib()
        recs, self.names, self.marks, self.switches, self.fixups = [], [], {}, [], []
        self.keep = []
        self.syncs = []                   # [(program index, SyncPoint)]: host-side collectives between two records
        mut = {id(c): c for c in mutable_cells}

        def add_call(c, kind, cond=None, cond_val=0):
            fn = lib.myolo_prog_fn_id(c.name.encode())
            if fn < 0:
                raise KeyError(c.name)
            argtypes = L._PROTOS[c.name][1][:-1]
            if len(argtypes) != len(c.args) or len(c.args) > L.PROG_MAX_ARGS:
                raise TypeError(f'{c.name}: {len(c.args)} arguments for {len(argtypes)} parameters')
            r = L.ProgOp()
            r.kind, r.fn, r.nargs = kind, fn, len(c.args)
            if cond is not None:
                r.cond, r.cond_val = C.addressof(cond), cond_val
            for i, (a, t) in enumerate(zip(c.args, argtypes)):
                r.a[i] = _slot_value(a, t)
                if id(a) in mut:
                    self.fixups.append((len(recs), i, a, t))
            recs.append(r)
            self.names.append(c.name)
            self.keep.append(c)

        for it in items:
            if isinstance(it, SwitchCall):
                self.switches.append(it)
                add_call(it.a, L.OP_CALL, it.cell, 0)
                add_call(it.b, L.OP_CALL, it.cell, 1)
            elif isinstance(it, Call):
                add_call(it, L.OP_CALL_SIDE if it.side else L.OP_CALL)
            elif isinstance(it, SyncPoint):
                self.syncs.append((len(recs), it))
            elif it[0] == 'memset':
                r = L.ProgOp()
                r.kind = L.OP_MEMSET
                r.a[0], r.a[1] = it[1].data_ptr(), int(it[2])
                recs.append(r)
                self.names.append('memset')
                self.keep.append(it[1])
            elif it[0] == 'join':
                r = L.ProgOp()
                r.kind = L.OP_JOIN
                recs.append(r)
                self.names.append('join')
            elif it[0] == 'mark':
                self.marks[it[1]] = len(recs)
            else:
                raise TypeError(it)
        self.n = len(recs)
        arr = (L.ProgOp * max(self.n, 1))(*recs)
        self.handle = lib.myolo_prog_create(arr, self.n)
        if not self.handle:
            raise L.MyoloError('myolo_prog_create rejected the launch program')
        self.slots = [(lib.myolo_prog_slot(self.handle, op, i), cell, t) for op, i, cell, t in self.fixups]
        self._lib = lib

    def run(self, first=0, last=None, side=None):
        for sw in self.switches:
            sw.cell.value = 1 if sw.state[sw.key] else 0
        for slot, cell, t in self.slots:                          # caller tensors re-bound for this run
            slot[0] = _slot_value(cell, t)
        last = self.n if last is None else last
        st = torch.cuda.current_stream().cuda_stream
        pos = first
        for idx, sp in self.syncs:        # a sync point at index i runs before record i; one that sits exactly at `last` belongs to
            if idx < first or idx > last or (idx == last and last != self.n):      # the next range (callers walk contiguous ranges)
                continue
            if idx > pos:
                self._run(pos, idx, st, side)
            sp(None)
            pos = idx
        if last > pos or not self.syncs:
            self._run(pos, last, st, side)

    def _run(self, first, last, st, side):
        e = self._lib.myolo_prog_run(self.handle, first, last, st, side)
        if e:
            i = self._lib.myolo_prog_last_op(self.handle)
            L.check(e, self.names[i] if 0 <= i < self.n else 'myolo_prog_run')

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self._lib.myolo_prog_destroy(self.handle)
        except Exception:   # noqa: BLE001 -- interpreter shutdown
            pass


_RKEY_SERIAL = [0]


def _rkey(reducer):
    """cache key of a gradient-cut object (parallel.GradReducer / runtime.StageCuts): a serial number stored ON the object plus its layout
    parameters -- an id() can be reused by a rebuilt reducer with a different bucket layout"""
    if reducer is None:
        return None
    d = reducer.__dict__
    if '_myolo_serial' not in d:
        _RKEY_SERIAL[0] += 1
        d['_myolo_serial'] = _RKEY_SERIAL[0]
    return (d['_myolo_serial'], getattr(reducer, 'nbuckets', None), getattr(reducer, 'world', None))


class Plan:
    """A built forward (+backward) launch plan for one (module, input shapes, dtype, mode)."""

    def __init__(self, device, dtype, training):
        self.device, self.dtype, self.training = device, dtype, training
        self.ops, self.bufs, self.tvs = [], [], []
        self.in_meta, self.in_ptr = [], []          # per input slot: {'shape','stride','dtype'} / ctypes pointer cell
        self.input_grads, self.outputs, self.output_grads = {}, {}, {}
        self.output_scales = {}                     # output slot -> (device scalar, state) of SegOutOp's gradient factor
        self.det_grads = []
        self.params, self._pgrad = [], {}
        self._pack_jobs, self._pack_call = [], None
        self.use_side_stream = True
        self.built = False

    # ---- graph construction -------------------------------------------------------------------------
    def new(self, n, h, w, c, requires_grad=True):
        tv = TV(self, n, h, w, c, requires_grad)
        self.tvs.append(tv)
        return tv

    def new_buf(self, n, h, w, c):
        b = Buf(n, h, w, c, self.dtype)
        self.bufs.append(b)
        return b

    def add(self, op):
        op.branch = getattr(self, 'cur_branch', None)       # set by the model while it emits an independent branch (models/yolo.py)
        self.ops.append(op)
        return op

    def add_pack_job(self, w, dst, cout, cin, ntaps, rows_pad, cols_pad, transpose, w2=None):
        """w2: a second weight tensor stacked behind `w` along cout (ConvOp weight2)"""
        assert w2 is None or w2.dtype == w.dtype
        self._pack_jobs.append((w, dst, cout, cin, ntaps, rows_pad, cols_pad, transpose, w2))

    def _build_pack_table(self):
        CH = 8192
        rows, chunks = [], []
        for j, (w, dst, cout, cin, ntaps, rp, cp, tr, w2) in enumerate(self._pack_jobs):
            rows.append((w.data_ptr(), dst.data_ptr(), cout, cin, ntaps, rp, cp, tr, L.DT[w.dtype], L.DT[dst.dtype],
                         w2.data_ptr() if w2 is not None else 0, w2.shape[0] if w2 is not None else 0))
            if PACK_TILED:              # LDS-tiled pack: coalesced source reads (the packed tensors are zero-initialised: padding stays)
                ca = cout + (w2.shape[0] if w2 is not None else 0)
                tco, tci = (64, 64) if ntaps == 1 else ((32, 16) if tr else (16, 32))
                for t_ in range(((ca + tco - 1) // tco) * ((cin + tci - 1) // tci)):
                    chunks.append((j, t_))
            else:
                for s0 in range(0, rp * ntaps * cp, CH):
                    chunks.append((j, s0))
        self._pack_key = tuple((r[0], r[10]) for r in rows)
        if not rows:
            self._pack_call = None
            return
        dev = self.device
        self._pack_tab = torch.tensor(rows, dtype=torch.int64).reshape(-1, 12).to(dev)
        self._pack_chunks = torch.tensor(chunks, dtype=torch.int32).reshape(-1, 2).to(dev)
        self._pack_call = Call('myolo_pack_weights_mt', (L.ptr(self._pack_tab), L.ptr(self._pack_chunks), len(chunks), -max(r[4] for r in rows) if PACK_TILED else CH))

    def add_input(self, t):
        """declare an input tensor slot (shape/stride/dtype are part of the plan; the pointer is rebound per run)."""
        self.in_meta.append({'shape': tuple(t.shape), 'stride': tuple(t.stride()), 'dtype': t.dtype})
        self.in_ptr.append(C.c_void_p(0))
        return len(self.in_meta) - 1

    def cat(self, tvs):
        """channel concat without a copy when the producer has not been placed yet (common.py:589)."""
        n, h, w = tvs[0].n, tvs[0].h, tvs[0].w
        ctot = sum(t.c for t in tvs)
        b = self.new_buf(n, h, w, ctot)
        off = 0
        for t in tvs:
            assert (t.n, t.h, t.w) == (n, h, w), 'concat of mismatched maps'
            if t.buf is None:
                t.place(b, off)
            else:
                dst = self.new(n, h, w, t.c, t.requires_grad)
                dst.place(b, off)
                self.add(CopyUpOp(self, t, dst, 1))
            off += t.c
        out = self.new(n, h, w, ctot, any(t.requires_grad for t in tvs))
        out.place(b, 0)
        return out

    def _carve(self, which, n):
        arena, used = self._arena[which], self._used[which]
        n = rup(n, 64)
        assert used + n <= arena.numel(), 'fp32 arena exhausted'
        self._used[which] = used + n
        return arena[used:used + n]

    def f32_fwd_zero(self, n):
        """fp32 scratch zeroed at the start of every forward (BN sum / sum-of-squares accumulators)."""
        return self._carve(0, n)

    def f32_bwd_zero(self, n):
        """fp32 scratch zeroed at the start of every backward (BN backward sums, gate partials)."""
        return self._carve(1, n)

    def sole_reader(self, tv, producer, reader):
        """True when `reader` is the only op of the plan besides `producer` that holds a view overlapping `tv` (any attribute that is a TV, or a
        list / tuple of them: inputs, residuals, concat members, exported outputs).  Fusions that stop writing an intermediate tensor
        (myolo_conv_pair) ask this at build time instead of trusting the emitting module's intent."""
        def views(o):
            for v in vars(o).values():
                if isinstance(v, TV):
                    yield v
                elif isinstance(v, (list, tuple)):
                    for u in v:
                        if isinstance(u, TV):
                            yield u
        lo, hi = tv.coff, tv.coff + tv.c
        for o in self.ops:
            if o is producer or o is reader:
                continue
            for v in views(o):
                if v.buf is tv.buf and v.coff < hi and v.coff + v.c > lo:
                    return False
        return True

    def grid_barrier(self):
        """state of the in-launch device-wide barrier (include/myolo.h MYOLO_GRID_BARRIER_BYTES; zeroed once, self-resetting; shared by the
        launches of the plan's MAIN stream: they are stream-ordered)"""
        if not hasattr(self, '_grid_bar'):
            self._grid_bar = torch.zeros(19 * 32, dtype=torch.int32, device=self.device)
        return self._grid_bar

    def wgrad_workspace(self):
        """split-K partial tiles of myolo_conv_wgrad (shared by all convs of the plan: launches are stream-ordered)."""
        if not hasattr(self, '_wg_ws'):
            self._wg_ws = torch.empty(48 << 20 >> 2, dtype=torch.float32, device=self.device)
        return self._wg_ws

    def rng_counter(self):
        if not hasattr(self, '_rng'):
            self._rng = torch.randint(0, 2 ** 40, (1,), dtype=torch.int64, device=self.device)
        return self._rng

    def pgrad(self, p):
        """fp32 gradient slot of a parameter inside the plan's flat gradient buffer."""
        if p is None:
            return None
        cur = getattr(self, '_building', None)
        if cur is not None:
            self._pg_first_op[id(p)] = min(self._pg_first_op.get(id(p), cur), cur)
        return self._pgrad[id(p)]

    # ---- build --------------------------------------------------------------------------------------
    def register_params(self, params):
        self.params = list(params)

    def build(self):
        dev = self.device
        for tv in self.tvs:
            if tv.buf is None:
                b = self.new_buf(tv.n, tv.h, tv.w, rup(tv.c, SEG[self.dtype]))
                tv.place(b, 0)
        for b in self.bufs:
            b.alloc(dev, self.training)
        self._arena = [torch.zeros(1 << 22, dtype=torch.float32, device=dev) for _ in range(2)]
        self._used = [0, 0]
        if self.training:
            tot = sum(p.numel() for p in self.params)
            self.flat_grad = torch.zeros(tot, dtype=torch.float32, device=dev)
            off = 0
            for p in self.params:
                self._pgrad[id(p)] = self.flat_grad[off:off + p.numel()].view(p.shape)
                off += p.numel()
            for b in self.bufs:
                b.gwritten = []
            for b in self.bufs:
                b.gwriters = []
            self._claim_log = []
            for op in reversed(self.ops):
                self._claiming = op
                op.plan_bwd(self)
            self._claiming = None
            self._check_group_claims()
            self._plan_tiny_groups()
            self._plan_bn_stats()
        self._pg_first_op = {}
        for i, op in enumerate(self.ops):
            self._building = i
            op.build(self)
        self._building = None
        # eval: the launches of a tagged branch go to the side stream (CALL_SIDE: behind everything issued on the main stream so far);
        # the main stream joins before the first op added after the module's own (the output ops read the branch's results)
        self._fwd_side = False
        if not self.training:
            for op in self.ops:
                if getattr(op, 'branch', None):
                    for c in op.fwd_calls:
                        if isinstance(c, Call):
                            c.side = True
                            self._fwd_side = True
        self._build_pack_table()
        self.built = True
        if torch.device(self.device).type == 'cuda':     # a CPU-device plan is a dry build (shape/launch-list checks only)
            self.prepare()

    def _plan_tiny_groups(self):
        """1x1 Conv+BatchNorm+activation layers on tiny maps (ConvOp.tiny_ok) leave the conv / BatchNorm launch chain: one workgroup per
        layer (myolo_tiny_conv_fwd / _bwd).  Up to four of them share a launch when nothing that runs between the first and the last
        depends on one of them: PyramidPooling's pool -> conv -> upsample triples (common.py:521-537) qualify when the four upsamples
        are one fused launch at the last of them (BilinearOp._group_fused) -- then every op between two member convs is a pool, or an
        upsample that launches nothing at its own position."""
        if not TINY_CONV:
            return
        ops = self.ops
        cand = [i for i, op in enumerate(ops) if isinstance(op, ConvOp) and op.tiny_ok(self)]
        k = 0
        while k < len(cand):
            grp = [cand[k]]
            while len(grp) < L.TINY_MAX_GROUP and k + len(grp) < len(cand):
                j = cand[k + len(grp)]
                between = ops[grp[-1] + 1:j]
                free = all(isinstance(o, AvgPoolOp) or (isinstance(o, BilinearOp) and o.group is not None and o is not o.group[-1] and
                                                        o._group_fused(self) and o._group_fused(self, training=False)) for o in between)
                prior = [ops[i].out for i in grp]
                prior_x = [ops[i].x for i in grp]

                def touches(tv, others=prior):      # does tv alias the output (the input) of an earlier member?
                    return any(tv.buf is o.buf and tv.coff < o.coff + o.c and o.coff < tv.coff + tv.c for o in others)
                # (two members reading the SAME tensor would add their input gradients into one range from two workgroups at once)
                if not free or touches(ops[j].x) or touches(ops[j].x, prior_x) or touches(ops[j].out) or any(isinstance(o, AvgPoolOp) and touches(o.src) for o in between):
                    break
                grp.append(j)
            members = [ops[i] for i in grp]
            for o in members:
                o.group = members
            k += len(grp)

    def _plan_bn_stats(self):
        """fold `bn_act_bwd_reduce` of a Conv+BatchNorm layer into the dgrad launch that writes the LAST contribution to its output
        gradient (the values are final in that launch's epilogue): possible when that last writer is a convolution's dgrad covering
        the layer's whole channel range, over the same pixels.  engine.BN_STATS_IN_DGRAD = False keeps the separate reduce launches."""
        if not BN_STATS_IN_DGRAD:      # default on: 9.40 -> 9.29 ms on the final build of round 2
            return
        max_elems = BN_STATS_MAX_ELEMS
        for op in self.ops:
            if not isinstance(op, ConvOp) or op.bn is None or op.det or op.bn2 is not None or op.group:
                continue
            o = op.out
            if o.n * o.h * o.w * o.c > max_elems:      # big maps: the separate reduce pass streams at 3-4 TB/s, cheaper than 8-byte epilogue loads
                continue
            lo, hi = o.coff, o.coff + o.c
            ws = [(a, z, w) for a, z, w in o.buf.gwriters if a < hi and z > lo]
            if not ws:
                continue
            a, z, w = ws[-1]
            if not isinstance(w, ConvOp) or w is op or a > lo or z < hi or len(w.bnb_targets) >= 4 or w.group:
                continue
            if (w.x.n, w.x.h, w.x.w) != (o.n, o.h, o.w) or w.x.buf is not o.buf:
                continue
            if w.s not in (1, 2) or (w.s == 2 and w.k != 3):
                continue
            op.reduce_by = w
            w.bnb_targets.append((op, lo - w.x.coff, hi - w.x.coff))

    def _check_group_claims(self):
        """a grouped AdaptiveAvgPool backward claims its source gradient at the LAST member's position (first in the backward) and
        writes it -- zero fills included -- in the FIRST member's launch: nothing in between may touch that range"""
        pos = {id(op): i for i, op in enumerate(self.ops)}
        for op, buf, lo, hi, _, _ in self._claim_log:
            g = getattr(op, 'group', None)
            if not isinstance(op, AvgPoolOp) or g is None or len(g) < 2:
                continue
            first, last = pos[id(g[0])], pos[id(g[-1])]
            for o2, b2, lo2, hi2, _, _ in self._claim_log:
                if o2 is not op and b2 is buf and lo2 < hi and hi2 > lo and first <= pos[id(o2)] < last:
                    raise L.MyoloError('grouped AdaptiveAvgPool backward: another op writes the source gradient between the group\'s '
                                       'claim and its launch')

    # ---- pruned backward ----------------------------------------------------------------------------
    # train.py:364-392 runs a detection pass and a segmentation pass per iteration: each backward starts from ONE of the two outputs
    # and the reference's autograd never visits the other head.  The static backward list would run that head (and every layer only it
    # depends on) over zero gradients.  bwd_schedule() derives, per set of outputs that DID receive a gradient, the sub-list that can
    # contribute: an op runs iff its output gradient can be non-zero; an op that does not run but was the first writer of a gradient range
    # a running producer reads is replaced by a zero fill of that range; a running BatchNorm layer whose reduce pass rode in the
    # epilogue of a pruned dgrad gets its own reduce launch back.  Parameter gradients of pruned ops stay at the zeros of the flat
    # buffer's memset (= what the full list computes from a zero output gradient).
    def _out_slot(self, op):
        if isinstance(op, (SegOutOp, ExportOp)):
            return op.slot
        if isinstance(op, ConvOp) and op.det:
            return self.__dict__.get('det_slot', {}).get(id(op))
        return None

    def bwd_liveness(self, live_slots):
        """[does op i's backward run] when only the output slots in `live_slots` carry a gradient; None: not analysable"""
        n = len(self.ops)
        live, ranges = [False] * n, {}
        for i in range(n - 1, -1, -1):
            op = self.ops[i]
            outs, ins = op.grad_io()
            slot = self._out_slot(op)
            if slot is not None:
                live[i] = slot in live_slots
            elif isinstance(op, ConvOp) and op.det:
                return None                                   # a Detect level that is not a registered output
            elif isinstance(op, ImportOp):
                live[i] = True                                # exports the gradient of a module input: always (zeros included)
            else:
                live[i] = any(a < tv.coff + tv.c and z > tv.coff for tv in outs for a, z in ranges.get(id(tv.buf), ()))
            if live[i]:
                for tv in ins:
                    ranges.setdefault(id(tv.buf), []).append((tv.coff, tv.coff + tv.c))
        for op in self.ops:                                   # grouped launches live on one member: the group runs or is pruned as a whole
            g = getattr(op, 'group', None)
            if g and len({live[self.ops.index(o)] for o in g}) > 1:
                return None
        return live

    def bwd_schedule(self, live_slots):
        """the backward in execution order as [('op', i) | ('reduce', i) | ('fill', i, buf, c0, c1)] for the outputs in `live_slots`;
        (a fill stands at the position i of the pruned op it replaces) 'full' when nothing can be pruned, None when the plan cannot be analysed (the caller runs the full list over zeros)"""
        if not self.training or not hasattr(self, '_claim_log'):
            return None
        live = self.bwd_liveness(frozenset(live_slots))
        if live is None:
            return None
        if all(lv for lv, op in zip(live, self.ops) if op.bwd_calls):
            return 'full'
        pos = {id(op): i for i, op in enumerate(self.ops)}
        read = {}                                             # gradient ranges a running op reads (its own output gradient)
        for i, op in enumerate(self.ops):
            if live[i]:
                for tv in op.grad_io()[0]:
                    read.setdefault(id(tv.buf), []).append((tv.coff, tv.coff + tv.c))
        claims = {}
        for op, buf, lo, hi, acc, gaps in self._claim_log:
            claims.setdefault(id(op), []).append((buf, lo, hi, acc, gaps))
        sched = []
        for i in range(len(self.ops) - 1, -1, -1):
            op = self.ops[i]
            if live[i]:
                rb = getattr(op, 'reduce_by', None)
                if rb is not None and not live[pos[id(rb)]] and op.bwd_calls:
                    sched.append(('reduce', i))
                sched.append(('op', i))
                continue
            for buf, lo, hi, acc, gaps in claims.get(id(op), ()):
                for a, z in (gaps if acc else [(lo, hi)]):
                    if any(ra < z and rz > a for ra, rz in read.get(id(buf), ())):
                        sched.append(('fill', i, buf, a, z))
        return sched

    def check_bwd_schedule(self, sched):
        """structural check of a schedule (tests): every gradient range a scheduled launch reads, or accumulates into, was written
        earlier in the same schedule.  Returns the list of violations."""
        done, bad = {}, []
        claims = {}
        for op, buf, lo, hi, acc, gaps in self._claim_log:
            claims.setdefault(id(op), []).append((buf, lo, hi, acc, gaps))

        def missing(buf, lo, hi):
            cur = lo
            for a, z in sorted(done.get(id(buf), ())):
                if a > cur:
                    break
                cur = max(cur, z)
            return cur < hi

        for e in sched:
            if e[0] == 'fill':
                done.setdefault(id(e[2]), []).append((e[3], e[4]))
                continue
            op = self.ops[e[1]]
            reads_out = e[0] == 'reduce' or (op.bwd_calls and self._out_slot(op) is None)
            if reads_out:
                for tv in op.grad_io()[0]:
                    if tv.requires_grad and missing(tv.buf, tv.coff, tv.coff + tv.c):
                        bad.append((e, 'reads an unwritten output gradient', tv.coff, tv.coff + tv.c))
            if e[0] == 'reduce':
                continue
            for buf, lo, hi, acc, gaps in claims.get(id(op), ()):
                if acc:
                    cur = lo
                    for a, z in sorted(gaps) + [(hi, hi)]:
                        if a > cur and missing(buf, cur, a):
                            bad.append((e, 'accumulates into an unwritten range', cur, a))
                        cur = max(cur, z)
                done.setdefault(id(buf), []).append((lo, hi))
        return bad

    def prepare(self):
        st = L.stream_ptr()
        for op in self.ops:
            if hasattr(op, 'prepare'):
                op.prepare()
            for c in op.prep_calls:
                c(st)

    # ---- run ----------------------------------------------------------------------------------------
    # A training plan is ~330 forward + ~340 backward C-ABI launches: issued one ctypes call at a time that is ~5.6 ms of host time per
    # step.  After two eager runs the launch lists are captured ONCE into hipGraphs (torch.cuda.CUDAGraph == hipGraph on ROCm) and a
    # run is a handful of graph launches.  Only plan-owned static buffers are baked into the graphs: the ops that read caller tensors
    # (FocusPackOp / ImportOp) stay outside and run eagerly first.  The backward is cut into BWD_SEGMENTS pieces; each piece is one
    # graph for the main stream (BatchNorm backward, dgrad, pooling ...) and one for the weight-gradient side stream, chained by
    # events BETWEEN graph launches -- no cross-stream dependency is captured (a multi-stream capture crashed HIP in round 1), and the
    # RCCL slices of a parallel.GradReducer are issued eagerly between the pieces.
    def _zero_fwd(self):
        if self._used[0]:
            self._arena[0][:self._used[0]].zero_()

    def _check_pack_table(self):
        if self._pack_call is not None and self._pack_key != tuple((j[0].data_ptr(), j[8].data_ptr() if j[8] is not None else 0) for j in self._pack_jobs):
            self._build_pack_table()                          # a parameter was re-allocated (.to(), load): new table, new graphs
            self.__dict__.pop('_graphs', None)
            for k in ('_nprog_fwd', '_nprog_bwd', '_split_progs'):        # every native program carries the old pointers
                self.__dict__.pop(k, None)

    def _fwd_lists(self):
        """(ops reading caller tensors -> eager, every other op -> graph)"""
        eager = [op for op in self.ops if isinstance(op, (FocusPackOp, ImportOp))]
        rest = [op for op in self.ops if not isinstance(op, (FocusPackOp, ImportOp))]
        return eager, rest

    def _fwd_body(self, st, ops):
        self._zero_fwd()
        if self._pack_call is not None:
            self._pack_call(st)
        for op in ops:
            for c in op.fwd_calls:
                c(st)

    def has_sync(self):
        """the launch lists contain host-side collectives (SyncBatchNorm with more than one rank): no hipGraph capture, one chain"""
        hs = self.__dict__.get('_has_sync')
        if hs is None:
            hs = self._has_sync = any(isinstance(c, SyncPoint) for op in self.ops for c in list(op.fwd_calls) + list(op.bwd_calls))
        return hs

    def graphable(self):
        return GRAPH_TRAIN and self.training and torch.device(self.device).type == 'cuda' and not self.has_sync()

    def native_ok(self):
        return NATIVE_EXEC and torch.device(self.device).type == 'cuda' and os.environ.get('MYOLO_DBG_SKIP_WGRAD', '0') != '1'

    def _native_fwd(self):
        """the forward launch list as a native program (None: an entry point outside the executor's table -> the Python loop)"""
        np_ = self.__dict__.get('_nprog_fwd')
        if np_ is None:
            items = []
            if self._used[0]:
                items.append(('memset', self._arena[0], self._used[0] * 4))
            if self._pack_call is not None:
                items.append(self._pack_call)
            joined = not self._fwd_side
            for i, op in enumerate(self.ops):
                if not joined and i >= getattr(self, 'n_emit_ops', len(self.ops)):
                    items.append(('join',))
                    joined = True
                items += list(op.fwd_calls)
            if not joined:
                items.append(('join',))
            try:
                np_ = NativeProg(items, list(self.in_ptr) + list(getattr(self, 'mutable_cells', ())))
            except KeyError:
                np_ = False
            self._nprog_fwd = np_
        return np_ or None

    def eval_split_progs(self):
        """eval plan with a side-stream branch (models/yolo.py: the segmentation head): (main program, index of the fork in it, branch
        program) so that the caller can run the branch WITHOUT joining it back -- detect.py's NMS then runs on the main stream while the
        head is still busy on the side stream.  None when the plan has no branch, or when something after the module's own ops reads the
        branch (only Detect's decode launches may follow; the deferred logit upsample has no launch)."""
        if self.training or not getattr(self, '_fwd_side', False) or not self.native_ok() or not self.use_side_stream:
            return None
        n_emit = getattr(self, 'n_emit_ops', len(self.ops))
        if any(op.fwd_calls and not isinstance(op, DecodeOp) for op in self.ops[n_emit:]):
            return None
        cached = self.__dict__.get('_split_progs')
        if cached is None:
            main, side = [], []
            if self._used[0]:
                main.append(('memset', self._arena[0], self._used[0] * 4))
            if self._pack_call is not None:
                main.append(self._pack_call)
            forked = False
            for op in self.ops:
                if getattr(op, 'branch', None) and op.fwd_calls:
                    if not forked:
                        main.append(('mark', 'fork'))
                        forked = True
                    side += list(op.fwd_calls)          # (their `side` flag falls back to the launch stream: run() gets no side stream)
                else:
                    main += list(op.fwd_calls)
            try:
                cells = list(self.in_ptr) + list(getattr(self, 'mutable_cells', ()))
                pm, ps = NativeProg(main, cells), NativeProg(side, cells)
                cached = (pm, pm.marks['fork'], ps) if forked else False
            except KeyError:
                cached = False
            self._split_progs = cached
        return cached or None

    def run_fwd(self):
        st = L.stream_ptr()
        self._check_pack_table()
        if not self.graphable():
            np_ = self._native_fwd() if self.native_ok() else None
            if np_ is not None:
                np_.run(side=self._side_stream().cuda_stream if (self._fwd_side and self.use_side_stream) else None)
            else:
                self._fwd_body(st, self.ops)
            return
        g = self.__dict__.setdefault('_graphs', {'warm': 0})
        eager, rest = self._fwd_lists()
        for op in eager:
            for c in op.fwd_calls:
                c(st)
        if g['warm'] < 2 or g.get('failed'):
            g['warm'] += 1
            self._fwd_body(st, rest)
            return
        if 'fwd' not in g:
            try:
                g['fwd'] = self._capture(lambda s: self._fwd_body(s, rest))
            except Exception as e:  # noqa: BLE001 -- capture is an optimisation only: fall back to the eager launch list, loudly once
                import warnings
                warnings.warn(f'multiyolov5_amd: hipGraph capture of the training forward failed ({e!r}); running eagerly')
                g['failed'] = True
                self._fwd_body(st, rest)
                return
        g['fwd'].replay()

    def _capture(self, body, stream=None):
        """capture body(stream_ptr) into a hipGraph (nothing executes during capture)"""
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            body(L.stream_ptr())
        return graph

    def flat_grad_is_cuda(self):
        return torch.device(self.device).type == 'cuda'

    def grad_buckets(self, reducer):
        """bucket layout of flat_grad for a parallel.GradReducer (cached per reducer)."""
        key = _rkey(reducer)
        if getattr(self, '_bucket_key', None) != key:
            sizes = [p.numel() for p in self.params]
            first = [self._pg_first_op.get(id(p), 0) for p in self.params]
            self._buckets = reducer.layout(sizes, first)
            self._bucket_key = key
        return self._buckets

    def _zero_bwd(self):
        if self._used[1]:
            self._arena[1][:self._used[1]].zero_()
        if self.training:
            self.flat_grad.zero_()

    def _side_stream(self):
        if getattr(self, '_side', None) is None:
            # (round 5: WHICH hardware queue the second stream lands on -- 1-3 streams taken out of torch's pool first, the high-priority
            #  pool, GPU_MAX_HW_QUEUES=8 -- measured neutral: 961-969 FPS / 7.75-7.81 ms for every placement)
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def _bwd_segments(self, reducer):
        """cut points of the backward op list (descending op index): [(hi, lo, [buckets ready after op lo])] -- a bucket boundary
        always ends a segment, the rest is balanced by launch count"""
        cache = self.__dict__.setdefault('_seg_cache', {})
        if _rkey(reducer) in cache:
            return cache[_rkey(reducer)]
        pending = list(self.grad_buckets(reducer)) if reducer is not None else []
        n = len(self.ops)
        cuts = {0}
        for _, _, ready in pending:
            cuts.add(max(0, min(n, ready)))
        ncalls = [len(op.bwd_calls) for op in self.ops]
        tot = sum(ncalls)
        nseg = max(1, BWD_SEGMENTS)
        acc, k = 0, 1
        for i in range(n - 1, 0, -1):
            acc += ncalls[i]
            if acc >= tot * k / nseg:
                cuts.add(i)
                k += 1
        cuts = sorted(cuts, reverse=True)
        segs, hi = [], n
        for lo in cuts:
            if lo == hi:
                continue
            ready = [(a, b) for a, b, r in pending if lo <= r < hi] if lo > 0 else [(a, b) for a, b, r in pending if r < hi]
            segs.append((hi, lo, ready))
            hi = lo
        cache[_rkey(reducer)] = segs
        return segs

    def _seg_main(self, st, hi, lo, with_side):
        for i in range(hi - 1, lo - 1, -1):
            for c in self.ops[i].bwd_calls:
                if not (c.side and with_side):
                    c(st)

    def _seg_side(self, st, hi, lo):
        for i in range(hi - 1, lo - 1, -1):
            for c in self.ops[i].bwd_calls:
                if c.side:
                    c(st)

    def capture_bwd(self, reducer=None):
        """capture the backward pieces (called from the forward, on the caller's thread, once the forward graph exists: nothing
        executes during a capture, so no gradients are needed yet).  Returns True when the backward can be replayed."""
        g = self.__dict__.get('_graphs')
        if not self.graphable() or g is None or 'fwd' not in g or g.get('failed'):
            return False
        use_side = self.use_side_stream
        if g.get('bwd_key') == (_rkey(reducer), use_side):
            return True
        try:
            caps = []
            if GRAPH_BWD == 'fork' and reducer is None:
                g['bwd'], g['bwd_key'] = [(self._capture(lambda s: self._bwd_eager(None)), 'fork')], (_rkey(reducer), use_side)
                return True
            for hi, lo, _ in self._bwd_segments(reducer):
                gm = self._capture(lambda s, hi=hi, lo=lo: self._seg_main(s, hi, lo, use_side))
                gs = None
                if use_side and any(c.side for i in range(lo, hi) for c in self.ops[i].bwd_calls):
                    gs = self._capture(lambda s, hi=hi, lo=lo: self._seg_side(s, hi, lo))
                caps.append((gm, gs))
            g['bwd'], g['bwd_key'] = caps, (_rkey(reducer), use_side)
            return True
        except Exception as e:  # noqa: BLE001 -- capture is an optimisation only
            import warnings
            warnings.warn(f'multiyolov5_amd: hipGraph capture of the training backward failed ({e!r}); running eagerly')
            g['failed'] = True
            return False

    def _sched_calls(self, sched):
        """{op index: launches} of a pruned schedule (Plan.bwd_schedule)"""
        per = {}
        for e in sched:
            calls = per.setdefault(e[1], [])
            if e[0] == 'op':
                calls += list(self.ops[e[1]].bwd_calls)
            elif e[0] == 'reduce':
                calls.append(self.ops[e[1]].bn_reduce_call())
            else:
                _, _, buf, a, z = e
                tv = TV(self, buf.n, buf.h, buf.w, z - a)
                tv.place(buf, a)
                zd = tv.desc(grad=True)
                calls.append(Call('myolo_fill_zero', (C.byref(zd),), keep=zd))
        return per

    def _bwd_items(self, reducer, live=None):
        """the launch items of the backward program for `reducer` (see NativeProg); `live`: frozenset of the output slots that received
        a gradient -> the pruned list, or 'full' when there is nothing to prune / the plan cannot be analysed"""
        per = None
        if live is not None:
            sched = self.bwd_schedule(live)
            if sched is None or sched == 'full':
                return 'full'
            per = self._sched_calls(sched)
        items = []
        if self._used[1]:
            items.append(('memset', self._arena[1], self._used[1] * 4))
        items.append(('memset', self.flat_grad, self.flat_grad.numel() * 4))
        for si, (hi, lo, ready) in enumerate(self._bwd_segments(reducer)):
            for i in range(hi - 1, lo - 1, -1):
                items += list(self.ops[i].bwd_calls) if per is None else per.get(i, [])
            if ready:
                items.append(('join',))
            items.append(('mark', si))
        items.append(('join',))
        return items

    def _native_bwd(self, reducer, live=None):
        """the backward launch list as a native program, cut at the gradient-slice boundaries of `reducer`.  `live`: see _bwd_items
        (the SAME program object as live=None when nothing can be pruned)"""
        cache = self.__dict__.setdefault('_nprog_bwd', {})
        key = (_rkey(reducer), live)
        if key not in cache:
            items = self._bwd_items(reducer, live)
            if items == 'full':
                cache[key] = self._native_bwd(reducer) or False
            else:
                try:
                    cache[key] = NativeProg(items)
                except KeyError:
                    cache[key] = False
        return cache[key] or None

    def pruned_bwd(self, reducer, live):
        """the native backward program for a backward that starts from the output slots in `live` only, or None when the full list
        runs (every output has a gradient, nothing to prune, no native executor, captured training graphs)"""
        if not PRUNE_BWD or not live or not self.training or not self.native_ok() or self.graphable() or not self.flat_grad.is_cuda:
            return None
        full = self._native_bwd(reducer)
        np_ = self._native_bwd(reducer, frozenset(live))
        return np_ if (np_ is not None and np_ is not full) else None

    def _bwd_native(self, np_, reducer):
        side = self._side_stream().cuda_stream if self.use_side_stream else None
        first = 0
        for si, (hi, lo, ready) in enumerate(self._bwd_segments(reducer)):
            if ready:                                         # every kernel writing into these slices has been enqueued (and joined)
                np_.run(first, np_.marks[si], side)
                first = np_.marks[si]
                for a, b in ready:
                    reducer.reduce_slice(self.flat_grad, a, b)
        np_.run(first, np_.n, side)
        if reducer is not None:
            reducer.finish(self.flat_grad)

    def _bwd_eager(self, reducer, prog=None):
        """the backward launch list call by call on the CURRENT stream (also the body of the 'fork' capture): weight-gradient launches
        are forked to the side stream behind an event each, gradient slices go to the reducer as soon as they are final.  `prog`: a
        pruned native program of this plan and reducer (Plan.pruned_bwd)"""
        if self.native_ok() and self.flat_grad.is_cuda and not torch.cuda.is_current_stream_capturing():
            np_ = prog if prog is not None else self._native_bwd(reducer)
            if np_ is not None:
                self._bwd_native(np_, reducer)
                return
        cuda = self.flat_grad.is_cuda
        main = torch.cuda.current_stream() if cuda else None
        side = self._side_stream() if (cuda and self.use_side_stream) else None
        self._zero_bwd()
        if side is not None:
            side.wait_stream(main)                            # the zero fills above
        side_ptr = C.c_void_p(side.cuda_stream) if side is not None else None
        st = L.stream_ptr()
        skip_side = os.environ.get('MYOLO_DBG_SKIP_WGRAD', '0') == '1'       # profiling only: how long is the step WITHOUT the weight gradients?
        evs = self.__dict__.setdefault('_side_events', [])                   # one event per fork point, created once (78 per step:
        k = 0                                                                # creating them anew was ~1 ms of host time per step)
        for hi, lo, ready in self._bwd_segments(reducer):
            for i in range(hi - 1, lo - 1, -1):
                for c in self.ops[i].bwd_calls:
                    if c.side and skip_side:
                        continue
                    if c.side and side is not None:
                        if k == len(evs):
                            evs.append(torch.cuda.Event())
                        ev = evs[k]
                        k += 1
                        ev.record(main)
                        side.wait_event(ev)
                        c(side_ptr)
                    else:
                        c(st)
            if ready:                                         # every kernel writing into these slices has been enqueued
                if side is not None:
                    main.wait_stream(side)
                for a, b in ready:
                    reducer.reduce_slice(self.flat_grad, a, b)
        if side is not None:
            main.wait_stream(side)
        if reducer is not None:
            reducer.finish(self.flat_grad)

    def run_bwd(self, reducer=None, prog=None):
        """backward launch list.  Weight-gradient kernels go to a side HIP stream: they only feed the optimizer, so they
        overlap with the latency-bound dgrad / BatchNorm chain on the main stream (many of those launches fill < 1 CU wave)."""
        if prog is not None:                                  # pruned list (only handed out when the native executor runs the backward)
            self._bwd_eager(reducer, prog)
            return
        g = self.__dict__.get('_graphs')
        graphed = self.graphable() and g is not None and 'fwd' in g and not g.get('failed')
        if graphed and 'bwd' not in g:
            graphed = self.capture_bwd(reducer)
        use_side = self.flat_grad.is_cuda and self.use_side_stream
        if graphed and g.get('bwd_key') != (_rkey(reducer), use_side):
            graphed = False                                   # reducer attached / detached after the capture: eager this time
        if not graphed:
            self._bwd_eager(reducer)
            return
        if g['bwd'][0][1] == 'fork':
            g['bwd'][0][0].replay()
            return
        main = torch.cuda.current_stream()
        side = self._side_stream() if use_side else None
        self._zero_bwd()
        if side is not None:
            side.wait_stream(main)
        for (hi, lo, ready), (gm, gs) in zip(self._bwd_segments(reducer), g['bwd']):
            gm.replay()
            if gs is not None:
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    gs.replay()
            if ready:
                if side is not None:
                    main.wait_stream(side)
                for a, b in ready:
                    reducer.reduce_slice(self.flat_grad, a, b)
        if side is not None:
            main.wait_stream(side)
        if reducer is not None:
            reducer.finish(self.flat_grad)

    def nbytes(self):
        tot = 0
        for b in self.bufs:
            tot += b.t.numel() * b.t.element_size() * (2 if b.g is not None else 1)
        return tot


# ---------------------------------------------------------------------------------------------------------
# output handles returned by module emit() functions besides plain TVs
class ImageInput:
    """the raw NCHW image fed to Focus (3 channels; consumed by FocusPackOp)."""

    def __init__(self, slot, shape):
        self.slot, self.shape = slot, shape


class DetHandle:
    """one Detect level in training layout [N,na,ny,nx,no] (yolo.py:214)."""

    def __init__(self, conv_op):
        self.op = conv_op


class SegHandle:
    """class logits (low-res TV) + the final bilinear scale (yolo.py:163): materialised by SegOutOp."""

    def __init__(self, low, scale):
        self.low, self.scale = low, scale


class DecodeHandle:
    """eval-mode Detect: (cat of decoded levels, [raw levels]) (yolo.py:225)."""

    def __init__(self, det_handles, module):
        self.dets, self.module = det_handles, module     # the live Detect module: anchors / strides are re-read by prepare()


class DecodeOp(Op):
    def __init__(self, plan, handle, slot):
        self.h, self.slot = handle, slot

    def build(self, plan):
        super().build(plan)
        convs = [d.op for d in self.h.dets]
        n = convs[0].out.n
        na, no = convs[0].det
        rows = [na * c.out.h * c.out.w for c in convs]
        a_total = sum(rows)
        self.z = torch.zeros(n, a_total, no, dtype=plan.dtype, device=plan.device)
        plan.outputs[self.slot] = self.z
        row0 = 0
        self.anch, self.strd = [], []
        for i, c in enumerate(convs):
            wh = (C.c_float * (na * 2))()
            sd = C.c_float(0.0)
            self.anch.append(wh)
            self.strd.append(sd)
            plan.__dict__.setdefault('mutable_cells', []).append(sd)   # (a launch-time HOST value: re-read by the native executor per run)
            self.fwd_calls.append(Call('myolo_detect_decode', (L.ptr(c.det_out), L.DT[plan.dtype], n, na, c.out.h, c.out.w, no,
                                                               sd, wh, L.ptr(self.z), a_total, row0)))
            row0 += rows[i]
        self.prepare()

    def prepare(self):
        """anchor_grid / stride are launch-time HOST constants of myolo_detect_decode: re-read them from the live module whenever
        a buffer version changed (load_state_dict into a live model, autoanchor's `m.anchor_grid[:] = ...`)"""
        m = self.h.module
        ag = m.anchor_grid.detach().float().cpu()
        for i, wh in enumerate(self.anch):
            v = ag[i].reshape(-1).tolist()
            for j in range(len(wh)):
                wh[j] = float(v[j])
            self.strd[i].value = float(m.stride[i])


# ---------------------------------------------------------------------------------------------------------
# algorithmic work of one myolo_conv launch (bench.py roofline): input + weights + output moved once, 2*MAC flops
def _conv_desc_of(call):
    return call.args[0]._obj


def _s2_descs(call):
    return [call.args[0][i].contents for i in range(call.args[1])]


def conv_call_bytes(call):
    if call.name == 'myolo_conv_dgrad_s2':       # the four parity sub-convolutions of a stride-2 dgrad: dy and the weights once, gx once
        ds = _s2_descs(call)
        es = 2 if ds[0].x.dtype == L.F16 else 4
        d0 = ds[0]
        return (d0.x.n * d0.x.h * d0.x.w * d0.x.c + sum(d.y.n * d.y.h * d.y.w * d.y.c for d in ds) + d0.y.c * d0.wtaps * d0.x.c) * es
    tot = 0
    for a in (call.args[:2] if call.name == 'myolo_conv_pair' else call.args[:1]):   # (a fused pair counts both layers' SURVEY 8(d) bytes: no fusion credit)
        d = a._obj
        es = 2 if d.x.dtype == L.F16 else 4
        tot += (d.x.n * d.x.h * d.x.w * d.x.c + d.y.n * d.y.h * d.y.w * d.y.c + d.y.c * d.ntaps * d.x.c) * es
    return tot


def apply_fold_bytes(call):
    """algorithmic bytes of the BatchNorm-backward apply pass a myolo_conv_dgrad_bn launch carries in its operand path: the layer's raw
    conv output read + dy written (gout is the launch's own input operand, counted by conv_call_bytes) -- the unfused pass' byte count"""
    if call.name != 'myolo_conv_dgrad_bn':
        return 0
    f = call.args[1]._obj
    return _tensor_bytes(f.y) + _tensor_bytes(f.dy)


def bnb_call_bytes(call):
    """algorithmic bytes of the BatchNorm-backward reduce passes a conv launch carries in its epilogue (myolo_conv_desc.bnb): the
    unfused pass reads the output gradient and the raw conv output of every segment once"""
    descs = _s2_descs(call)[:1] if call.name == 'myolo_conv_dgrad_s2' else [_conv_desc_of(call)]
    tot = 0
    for d in descs:
        for i in range(d.nbnb if d.bnb else 0):
            tot += _tensor_bytes(d.bnb[i].y) * 2
    return tot


def conv_call_flops(call):
    if call.name == 'myolo_conv_dgrad_s2':
        return sum(2.0 * d.y.n * d.y.h * d.y.w * d.y.c * d.ntaps * d.x.c for d in _s2_descs(call))
    d = _conv_desc_of(call)
    return 2.0 * d.y.n * d.y.h * d.y.w * d.y.c * d.ntaps * d.x.c


def _tensor_bytes(t, c=None):
    es = 2 if t.dtype == L.F16 else 4
    return t.n * t.h * t.w * (t.c if c is None else c) * es


def call_algorithmic_bytes(call):
    """algorithmic HBM bytes of one C-ABI launch of the training step (bench.py whole-step roofline): every operand read once, every
    result written once -- conv / dgrad: input + weights + output; wgrad: x + dy + fp32 gradient; BatchNorm forward: raw + out
    (+ residual); backward reduce: gout + raw; backward apply: gout + raw + dy (+ residual gradient).  None for other launches."""
    n = call.name
    if n in ('myolo_conv', 'myolo_conv_dgrad_s2', 'myolo_conv_dgrad_bn', 'myolo_conv_bn_act'):
        return conv_call_bytes(call)
    if n == 'myolo_conv_wgrad':
        d = call.args[0]._obj
        return _tensor_bytes(d.x) + _tensor_bytes(d.dy) + d.ntaps * max(d.cout, 1) * max(d.cin, 1) * 4
    a = [x._obj if hasattr(x, '_obj') else x for x in call.args]
    if n in ('myolo_bn_act_fwd', 'myolo_bn_act_fwd_split'):
        y, res = a[0], a[11]
        return _tensor_bytes(y) * 2 + (_tensor_bytes(res) if res.ptr else 0)
    if n in ('myolo_bn_act_bwd_reduce', 'myolo_bn_act_bwd_reduce_split'):
        return _tensor_bytes(a[0]) * 2
    if n in ('myolo_bn_act_bwd_apply', 'myolo_bn_act_bwd_apply_split'):
        g, gres = a[0], a[10]
        return _tensor_bytes(g) * 3 + (_tensor_bytes(gres) if gres.ptr else 0)
    if n == 'myolo_bn_wgrad_stem':          # (the operand count of the three launches it replaces: reduce 2, apply 3, weight gradient x + dy + dw)
        d, g = a[0], a[1]
        return _tensor_bytes(g) * 5
    if n == 'myolo_bn_act_bwd_fused':       # (the operand count of the two launches it replaces: the family totals stay comparable across builds)
        g, gres = a[0], a[10]
        return _tensor_bytes(g) * 5 + (_tensor_bytes(gres) if gres.ptr else 0)
    return None


def plan_algorithmic_bytes(plan):
    """{family: bytes} over the forward + backward launch lists of a training plan"""
    out = {'conv': 0, 'wgrad': 0, 'batchnorm': 0}
    fam = {'myolo_conv': 'conv', 'myolo_conv_dgrad_s2': 'conv', 'myolo_conv_dgrad_bn': 'conv', 'myolo_conv_bn_act': 'conv', 'myolo_conv_wgrad': 'wgrad', 'myolo_bn_act_fwd': 'batchnorm', 'myolo_bn_act_bwd_reduce': 'batchnorm',
           'myolo_bn_act_bwd_apply': 'batchnorm', 'myolo_bn_act_fwd_split': 'batchnorm', 'myolo_bn_act_bwd_reduce_split': 'batchnorm',
           'myolo_bn_act_bwd_apply_split': 'batchnorm', 'myolo_bn_act_bwd_fused': 'batchnorm', 'myolo_bn_wgrad_stem': 'batchnorm'}
    for op in plan.ops:
        for c in list(op.fwd_calls) + list(op.bwd_calls):
            b = call_algorithmic_bytes(c)
            if b is not None:
                out[fam[c.name]] += b
            if c.name == 'myolo_conv_bn_act':              # (the BatchNorm forward pass the launch carries keeps its unfused byte count: raw + out (+ residual))
                f = c.args[1]._obj
                out['batchnorm'] += _tensor_bytes(f.out) * 2 + (_tensor_bytes(f.res, f.out.c) if f.res.ptr else 0)
            if c.name in ('myolo_conv', 'myolo_conv_dgrad_s2', 'myolo_conv_dgrad_bn'):
                out['batchnorm'] += bnb_call_bytes(c) + apply_fold_bytes(c)    # (reduce / apply passes folded into dgrad launches keep their unfused byte count)
    return out
