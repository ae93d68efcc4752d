"""Module <-> plan glue: `PlannedModule.forward` builds (once per input signature) and runs a static launch plan.

The torch side is plumbing only: tensors own device memory, `torch.autograd.Function` hands the plan's backward to
`loss.backward()` (reference train.py:371,392) and to DDP's reducer hooks.  All arithmetic is in libmyolo.
"""
import torch
import torch.nn as nn

from . import _lib as L
from . import engine as E


def compute_dtype(tensors):
    """fp16 under autocast (train.py:364,380) or for .half() models/inputs (detect.py:103,136); fp32 otherwise."""
    if torch.is_autocast_enabled():
        return torch.float16
    for t in tensors:
        if t.dtype == torch.float16:
            return torch.float16
    return torch.float32


def _flatten(x, out):
    if torch.is_tensor(x):
        out.append(x)
        return ('t', len(out) - 1)
    if isinstance(x, (list, tuple)):
        return ('l', [_flatten(v, out) for v in x])
    raise TypeError(f'unsupported module input {type(x)}')


def _unflatten(spec, vals):
    kind, v = spec
    if kind == 't':
        return vals[v]
    return [_unflatten(s, vals) for s in v]


class _OutSpec:
    """walks the structure returned by emit(), registering plan outputs; rebuilds it from tensors at run time."""

    def __init__(self, plan):
        self.plan = plan
        self.nslots = 0
        self.det_slots = {}      # slot -> ConvOp (Detect training outputs)

    def register(self, y):
        plan = self.plan
        if isinstance(y, E.TV):
            s = self._slot()
            plan.add(E.ExportOp(plan, y, s))
            return ('o', s)
        if isinstance(y, E.SegHandle):
            s = self._slot()
            plan.add(E.SegOutOp(plan, y.low, y.scale, s))
            return ('o', s)
        if isinstance(y, E.DetHandle):
            s = self._slot()
            self.det_slots[s] = y.op
            plan.__dict__.setdefault('det_slot', {})[id(y.op)] = s       # (engine.Plan.bwd_liveness: which output a Detect level is)
            return ('o', s)
        if isinstance(y, E.DecodeHandle):
            s = self._slot()
            plan.add(E.DecodeOp(plan, y, s))
            return ('tuple', [('o', s), ('l', [self.register(d) for d in y.dets])])
        if isinstance(y, (list, tuple)):
            return ('l', [self.register(v) for v in y])
        raise TypeError(f'unsupported module output {type(y)}')

    def _slot(self):
        self.nslots += 1
        return self.nslots - 1

    @staticmethod
    def rebuild(spec, vals):
        kind, v = spec
        if kind == 'o':
            return vals[v]
        if kind == 'tuple':
            return tuple(_OutSpec.rebuild(s, vals) for s in v)
        return [_OutSpec.rebuild(s, vals) for s in v]


class LazySegLogits(torch.Tensor):
    """The training-mode segmentation output `pred[1]` (models/yolo.py:163: the x8 bilinear upsample of the class logits).  The fused
    loss reads the LOW-resolution logits (utils/loss._SegCE) and never needs these 38 bytes per pixel; everything else --
    any torch function or method other than pure metadata -- first runs the deferred upsample launch into the tensor's storage
    (`materialize`), so the values any other consumer sees are the reference's.  A tensor of an OLDER forward cannot be materialised
    once the plan's activations were overwritten: that raises instead of returning stale data."""
    _META = None

    @staticmethod
    def _meta_funcs():
        T = torch.Tensor
        names = ('shape', 'dtype', 'device', 'requires_grad', 'grad_fn', 'is_cuda', 'ndim', 'is_leaf', 'layout', 'grad', 'names',
                 'is_sparse', 'is_quantized', 'is_meta', 'output_nr', '_version', 'is_mkldnn', 'is_xpu', 'is_cpu')
        fs = {getattr(T, n).__get__ for n in names if hasattr(T, n)}
        fs |= {T.stride, T.size, T.dim, T.numel, T.is_contiguous, T.element_size, T.is_floating_point, T.is_complex, T.storage_offset,
               T.nelement, T.ndimension, T.get_device, T.__len__, T.register_hook, T.retain_grad, T.backward, T.requires_grad_,
               T.__hash__}
        fs.add(getattr(T, 'grad').__set__)
        return fs

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if cls._META is None:
            cls._META = cls._meta_funcs()
        if func not in cls._META:
            if func is torch.nn.functional.interpolate and LAZY_RESIZE:
                r = LazyResized.from_interpolate(args, kwargs or {})
                if r is not None:
                    return r                              # detect.py:191 / test.py:38: nothing launched until the result is used
            for a in tuple(args) + tuple((kwargs or {}).values()):
                for t in (a if isinstance(a, (list, tuple)) else (a,)):
                    if isinstance(t, LazySegLogits):
                        materialize(t)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))


LAZY_RESIZE = True          # tests flip it to compare against the ATen route


class _LazyTensor(torch.Tensor):
    """A tensor with the reference's metadata and NO storage whose values come from `_thunk()` on first real use.  Every torch function
    or method other than pure metadata either matches a pattern of the subclass (`_pattern`) or runs on the forced tensor; nothing ever
    reaches the dispatcher with the wrapper itself (that raises)."""

    @staticmethod
    def __new__(cls, shape, dtype, device):
        return torch.Tensor._make_wrapper_subclass(cls, tuple(shape), dtype=dtype, device=device)

    def _force(self):
        v = self.__dict__.get('_forced')
        if v is None:
            v = self.__dict__['_forced'] = self.__dict__['_thunk']()
        return v

    def _pattern(self, func, args, kwargs):
        return NotImplemented

    @classmethod
    def _swap(cls, a):
        if isinstance(a, _LazyTensor):
            return a._force()
        if isinstance(a, (list, tuple)):
            return type(a)(cls._swap(v) for v in a)
        return a

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if LazySegLogits._META is None:
            LazySegLogits._META = LazySegLogits._meta_funcs()
        if func in LazySegLogits._META:
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        if args and isinstance(args[0], _LazyTensor):
            r = args[0]._pattern(func, args, kwargs)
            if r is not NotImplemented:
                return r
        args = tuple(cls._swap(a) for a in args)
        kwargs = {k: cls._swap(v) for k, v in kwargs.items()}
        return func(*args, **kwargs)                       # (a forced LazySegLogits among them keeps its own protocol)

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        raise L.MyoloError(f'lazy tensor reached the dispatcher through {func}: it must be forced in __torch_function__')


class LazyResized(_LazyTensor):
    """`F.interpolate(seg, (h0, w0), mode='bilinear', align_corners=True)` of the model's segmentation output (detect.py:191, test.py:38)
    as a deferred view.  What the reference's callers do with it is `[0]`, `.data`, and a maximum over the class axis of which they keep
    the INDICES (`seg.max(axis=0)[1]`, detect.py:193; `torch.max(output, 1)`, utils/metrics.py:240,259): those run the fused resize +
    arg-max kernel (`myolo_seg_argmax`, csrc/seg_out.hip) on the head's logits and never form the h0 x w0 x 19 tensor; the label map is
    cached, so the two `torch.max` of test.py's metrics launch once.  The maximum VALUES are themselves deferred (nobody reads them);
    any other use computes the real interpolation (ATen) on the materialised logits, as before round 6."""

    @staticmethod
    def from_interpolate(args, kw):
        x = args[0] if args else kw.get('input')
        size = args[1] if len(args) > 1 else kw.get('size')
        if not isinstance(x, LazySegLogits) or x.dim() != 4 or size is None or kw.get('scale_factor') is not None:
            return None
        if kw.get('mode') != 'bilinear' or kw.get('align_corners') is not True or kw.get('antialias', False):
            return None
        if not isinstance(size, (tuple, list, torch.Size)) or len(size) != 2 or x.dtype not in (torch.float16, torch.float32) \
                or x.shape[1] > 32 or x.requires_grad:
            return None
        h0, w0 = int(size[0]), int(size[1])
        r = LazyResized((x.shape[0], x.shape[1], h0, w0), x.dtype, x.device)
        r.__dict__.update(_src=x, _hw=(h0, w0), _sel=None, _labels={})

        def real():
            materialize(x)
            with torch._C.DisableTorchFunctionSubclass():
                return torch.nn.functional.interpolate(x, (h0, w0), mode='bilinear', align_corners=True).as_subclass(torch.Tensor)
        r.__dict__['_thunk'] = real
        return r

    def _labels_all(self):
        lab = self._labels.get('all')
        if lab is None:
            from .utils.general import seg_argmax
            src = self._src
            st = src.__dict__.get('_myolo_lazy_state')
            if st is not None and not st['done'] and st['holder'].generation != st['generation']:
                materialize(src)                           # raises: the logits of an older forward are gone
            lab = self._labels['all'] = seg_argmax(src, *self._hw)
        return lab

    def _class_dim(self, d):
        if isinstance(d, bool) or not isinstance(d, int):
            return False
        return (d + self.dim() if d < 0 else d) == self.dim() - 3

    def _pattern(self, func, args, kw):
        T = torch.Tensor
        if func is T.__getitem__ and self._sel is None and isinstance(args[1], int) and not isinstance(args[1], bool):
            i = args[1] + (self.shape[0] if args[1] < 0 else 0)
            if not 0 <= i < self.shape[0]:
                raise IndexError(f'index {args[1]} is out of bounds for dimension 0 with size {self.shape[0]}')
            r = LazyResized(tuple(self.shape[1:]), self.dtype, self.device)
            r.__dict__.update(_src=self._src, _hw=self._hw, _sel=i, _labels=self._labels, _thunk=lambda: self._force()[i])
            return r
        if func in (T.detach, getattr(T, 'data').__get__):
            return self
        if func in (T.max, torch.max, T.argmax, torch.argmax):
            rest = dict(kw)
            d = args[1] if len(args) > 1 else rest.pop('dim', rest.pop('axis', None))
            keep = args[2] if len(args) > 2 else rest.pop('keepdim', False)
            if rest or len(args) > 3 or keep or not self._class_dim(d):
                return NotImplemented
            lab = self._labels_all()
            if self._sel is not None:
                lab = lab[self._sel]
            if func in (T.argmax, torch.argmax):
                return lab
            vals = _LazyTensor(tuple(lab.shape), self.dtype, self.device)
            vals.__dict__['_thunk'] = lambda: torch.max(self._force(), d)[0]
            return torch.return_types.max((vals, lab))
        return NotImplemented


def materialize(t):
    """run the deferred upsample of a LazySegLogits (no-op for anything else / when already done)"""
    st = t.__dict__.get('_myolo_lazy_state')
    if st is None or st['done']:
        return t
    holder, op = st['holder'], st['op']
    holder.wait_branch()
    if holder.generation != st['generation']:
        raise L.MyoloError('segmentation logits of an earlier forward were read after a newer forward of the same module overwrote '
                           "the plan's activations (they are materialised on first use; read them before the next forward, or set "
                           'MYOLO_LAZY_SEG=0)')
    op.lazy_call(L.stream_ptr())
    st['done'] = True
    return t


class PlanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, holder, *tensors):
        plan = holder.plan
        ctx.holder = holder
        ctx.set_materialize_grads(False)
        holder.bind_inputs(tensors[:holder.n_in])
        plan.run_fwd()
        plan.capture_bwd(holder.module.__dict__.get('_grad_reducer'))      # no-op until the forward graph exists / once captured
        holder.generation += 1                       # activations / BN statistics / dropout masks of this forward
        outs = PlanFn._wrap_outputs(holder)
        for sc in plan.output_scales.values():        # (a fused low-resolution CE of an earlier forward that never ran its backward)
            sc[1]['low'] = False
        ctx.generation = holder.generation
        holder.pending_bwd = True
        return outs

    @staticmethod
    def _wrap_outputs(holder):
        outs = []
        for o in holder.output_tensors():
            d = o.detach()
            lazy = o.__dict__.get('_myolo_lazy')
            if lazy is not None:
                d = d.as_subclass(LazySegLogits)
                d._myolo_lazy_state = {'holder': holder, 'op': lazy, 'generation': holder.generation, 'done': False}
            for attr in ('_myolo_low', '_myolo_grad_buf', '_myolo_grad_scale', '_myolo_low_grad'):   # side channels of the fused loss / argmax kernels
                if hasattr(o, attr):
                    setattr(d, attr, getattr(o, attr))
            outs.append(d)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        holder = ctx.holder
        plan = holder.plan
        if not holder.pending_bwd or ctx.generation != holder.generation:
            raise L.MyoloError('backward through a plan whose activations were overwritten by a newer forward '
                               '(one outstanding forward per module/shape; call backward before the next forward)')
        red = holder.module.__dict__.get('_grad_reducer')
        prog = PlanFn._take_output_grads(holder, plan, grads, red)
        plan.run_bwd(red, prog)
        return PlanFn._finish_backward(holder, plan)

    @staticmethod
    def _take_output_grads(holder, plan, grads, cuts=None):
        """the incoming output gradients -> the plan's gradient buffers (the fused losses already wrote theirs in place).  Returns the
        pruned backward program when only some outputs received a gradient (train.py:371 / 392: the detection pass, the segmentation
        pass -- engine.Plan.bwd_schedule), else None = the full launch list; `cuts`: the reducer / stage cuts the program is built for"""
        live = [s for s, g in enumerate(grads) if g is not None]
        prog = plan.pruned_bwd(cuts, live) if len(live) < len(grads) else None
        for s, g in enumerate(grads):
            dst = holder.output_grad_tensor(s)
            sc = plan.output_scales.get(s)
            if sc is not None and sc[1]['fresh'] and (g is None or g.data_ptr() != dst.data_ptr()):
                sc[1]['fresh'] = False
                raise L.MyoloError('the segmentation logits were consumed by the fused cross-entropy AND another differentiable op: '
                                   'their gradients cannot be combined (set MYOLO_FUSED_CE=0 for that use)')
            if g is None:
                if prog is None:                          # (the pruned list never reads this output's gradient buffer)
                    dst.zero_()
                if sc is not None:
                    sc[1]['low'] = False                  # this output's loss is not part of the backward: no low-resolution gradient either
            elif g.data_ptr() != dst.data_ptr() or g.stride() != dst.stride():
                dst.copy_(g)                              # (the fused losses already wrote into dst: nothing to move)
            if sc is not None:                            # gradient factor published by the fused CE (utils.loss._SegCE)
                scale, state = sc
                if state['fresh']:
                    state['fresh'], state['dirty'] = False, True
                elif state['dirty']:
                    scale.fill_(1.0)
                    state['dirty'] = False
        return prog

    @staticmethod
    def _finish_backward(holder, plan):
        for sc in plan.output_scales.values():
            sc[1]['low'] = False
        holder.pending_bwd = False
        in_grads = [plan.input_grads.get(i) if holder.in_requires_grad[i] else None for i in range(holder.n_in)]
        if _accumulate_in_place(holder, plan):
            # gradient accumulation (train.py:364-401 runs a detection and a segmentation backward per iteration and steps the optimizer
            # every `accumulate` iterations): every `p.grad` is still the view this function handed out of ONE flat buffer, so the
            # ~230 per-parameter `grad += new` kernels autograd would launch are one add over the flat buffer
            holder._accum_buf.add_(plan.flat_grad)
            return (None, *in_grads, *([None] * len(plan.params)))
        # the buffer handed out as `.grad` views is REUSED once nothing references the previous hand-out any more (zero_grad() dropped
        # the views): the gradients then keep their addresses from step to step, so pointer tables built over them (FusedSGD, ModelEMA:
        # a rebuild is a blocking host-to-device copy, i.e. a device sync + ~2 ms of host work per step) stay valid
        flat = holder.__dict__.get('_accum_buf')
        if flat is not None and torch._C._storage_Use_Count(flat.untyped_storage()._cdata) <= 2:
            flat.copy_(plan.flat_grad)
        else:
            flat = plan.flat_grad.clone()
        pg, off = [], 0
        for p in plan.params:
            n = p.numel()
            pg.append(flat[off:off + n].view(p.shape) if p.requires_grad else None)
            off += n
        holder._accum_buf = flat
        return (None, *in_grads, *pg)



# ---- the backward as a CHAIN of autograd nodes -------------------------------------------------------------------------------
# train.py:243-245 wraps the model in stock DistributedDataParallel: its reducer learns that a gradient is final from the
# AccumulateGrad hook of the parameter.  With ONE autograd node for the whole plan every hook fires after the last backward launch
# and the all-reduce of 31 MB starts when nothing is left to overlap it with.  Here the plan's backward launch list is cut
# at the same bucket boundaries parallel.GradReducer uses (parameters in backward-completion order, 3 slices of the flat gradient
# buffer): stage 0 runs the forward and, in backward, the launches that complete slice 0 (Detect / segmentation head / neck), hands
# those parameter gradients to autograd -- whose engine runs their AccumulateGrad nodes (DDP's hooks: bucket ready -> all-reduce on
# DDP's stream) BEFORE the next node -- then stage 1 runs the next piece of the launch list, and so on.  The stages are chained through
# a dummy token tensor; the model inputs hang on the LAST stage (their gradient is complete when the whole list has run).
# Used when a process group exists (stock DDP, or parallel.GradReducer with world > 1) or MYOLO_STAGED_BWD=force: every stage boundary makes
# the main stream wait for the weight-gradient stream (the slice must be final before it is handed out) -- 2 x ~0.2 ms of main-queue idle
# per step in the r3d trace, which only pays off when an exchange overlaps with the rest of the backward.
import os as _os_staged
STAGED_BWD = _os_staged.environ.get('MYOLO_STAGED_BWD', '1')
STAGES = 3


def _staged_wanted(holder):
    if STAGED_BWD in ('0', False):
        return False
    if STAGED_BWD in ('force', True):
        return True
    red = holder.module.__dict__.get('_grad_reducer')
    if red is not None:
        return red.world > 1
    return torch.distributed.is_available() and torch.distributed.is_initialized()


class StageCuts:
    """bucket layout of the flat gradient buffer without an exchange of its own (stock DDP / single GPU): parallel.GradReducer's rule"""
    world = 1

    def __init__(self, nbuckets=STAGES):
        self.nbuckets = nbuckets

    def layout(self, sizes, first_op):
        from .parallel import GradReducer
        return GradReducer.layout(self, sizes, first_op)

    def reduce_slice(self, flat, lo, hi):
        return

    def finish(self, flat):
        return


def _stage_plan(holder):
    """[(param index range, [flat slices], program op range)] per stage for this holder's plan, or None when the staged form does not
    apply (no native executor, a single bucket)"""
    plan = holder.plan
    if not (_staged_wanted(holder) and plan.training and plan.native_ok() and not plan.graphable()):
        return None
    red = holder.module.__dict__.get('_grad_reducer')
    cuts = red if red is not None else holder.__dict__.setdefault('_stage_cuts', StageCuts())
    # the cut object itself is held (an id() can be reused by a rebuilt reducer with another layout) together with the parameters'
    # requires_grad flags: a stage whose parameters are all frozen (train.py `freeze`) would never be visited by autograd
    flags = tuple(bool(p.requires_grad) for p in plan.params)
    key = (cuts, flags)
    st = holder.__dict__.get('_stages')
    if st is not None and st[0][0] is cuts and st[0][1] == flags and \
            (st[1] is None or holder.__dict__.get('_stage_prog') is plan._native_bwd(cuts)):   # (a re-packed plan has new programs)
        return st[1]
    stages = None
    if plan.flat_grad.is_cuda:
        np_ = plan._native_bwd(cuts)
        if np_ is not None:
            offs, off = [], 0
            for p in plan.params:
                offs.append(off)
                off += p.numel()
            # a stage is the program range between two segment marks ('mk': mark keys, None = program start / end): the same keys
            # address the pruned programs of this plan (engine.Plan.pruned_bwd), whose op indices differ
            stages, first = [], None
            segs = plan._bwd_segments(cuts)
            pend = []
            for si, (hi, lo, ready) in enumerate(segs):
                pend += ready
                last = si == len(segs) - 1
                if ready and not last:
                    stages.append({'mk': (first, si), 'slices': list(pend)})
                    first, pend = si, []
                elif last:
                    if pend or not stages:
                        stages.append({'mk': (first, None), 'slices': list(pend)})
                    else:                                  # nothing but the final join is left: it belongs to the last real stage (a stage
                        stages[-1]['mk'] = (stages[-1]['mk'][0], None)       # without parameters would be pruned by autograd)
            for sg in stages:
                sg['params'] = [i for i, o in enumerate(offs) if any(a <= o < b for a, b in sg['slices'])]
            covered = sorted(i for sg in stages for i in sg['params'])
            if len(stages) < 2 or covered != [i for i in range(len(plan.params)) if plan.params[i].numel() > 0]:
                stages = None
            elif not all(any(flags[i] for i in sg['params']) for sg in stages):
                # autograd only runs a stage's backward when something behind it needs a gradient; a stage of frozen
                # parameters (and inputs that need none) would be skipped together with the final join / flat accumulation: one node
                stages = None
            else:
                holder._stage_prog, holder._stage_cut_obj = np_, cuts
    holder._stages = (key, stages)
    return stages


class PlanStageFn(torch.autograd.Function):
    """stage k of a plan's backward chain (k = 0: also the forward).  forward(holder, k, nstage, token | model inputs..., params of the stage)"""

    @staticmethod
    def forward(ctx, holder, k, nstage, n_lead, *args):
        ctx.holder, ctx.k, ctx.nstage, ctx.n_lead = holder, k, nstage, n_lead
        ctx.set_materialize_grads(False)
        plan = holder.plan
        if k == nstage - 1:                           # the deepest stage sees the model inputs
            holder.bind_inputs(args[:n_lead])
        if k > 0:
            return torch.zeros((), device=plan.device)                      # the token
        plan.run_fwd()
        holder.generation += 1
        outs = PlanFn._wrap_outputs(holder)
        for sc in plan.output_scales.values():
            sc[1]['low'] = False
        ctx.generation = holder.generation
        holder.pending_bwd = True
        return outs

    @staticmethod
    def backward(ctx, *grads):
        holder, k, nstage = ctx.holder, ctx.k, ctx.nstage
        plan = holder.plan
        stages = holder._stages[1]
        sg = stages[k]
        np_, cuts = holder._stage_prog, holder._stage_cut_obj
        red = cuts if hasattr(cuts, 'stream') else None                      # a parallel.GradReducer (else StageCuts: nothing to exchange)
        if k == 0:
            if not holder.pending_bwd or ctx.generation != holder.generation:
                raise L.MyoloError('backward through a plan whose activations were overwritten by a newer forward '
                                   '(one outstanding forward per module/shape; call backward before the next forward)')
            holder._stage_prog_cur = PlanFn._take_output_grads(holder, plan, grads, cuts) or np_
            holder._bwd_accumulate = _accumulate_in_place(holder, plan)
            if not holder._bwd_accumulate:
                flat = holder.__dict__.get('_accum_buf')
                if flat is None or torch._C._storage_Use_Count(flat.untyped_storage()._cdata) > 2:
                    flat = torch.empty_like(plan.flat_grad)
                holder._accum_buf = flat
        side = plan._side_stream().cuda_stream if plan.use_side_stream else None
        np_ = holder._stage_prog_cur
        m0, m1 = sg['mk']
        np_.run(0 if m0 is None else np_.marks[m0], np_.n if m1 is None else np_.marks[m1], side)
        last = k == nstage - 1
        acc = holder._bwd_accumulate
        flat = holder._accum_buf
        for a, b in sg['slices']:
            if acc:
                if red is not None:
                    red.reduce_slice(plan.flat_grad, a, b)
            else:
                flat[a:b].copy_(plan.flat_grad[a:b])
                if red is not None:
                    red.reduce_slice(flat, a, b)
        if last:
            if red is not None:
                red.finish(plan.flat_grad if acc else flat)
            if acc:
                flat.add_(plan.flat_grad)
            for sc in plan.output_scales.values():
                sc[1]['low'] = False
            holder.pending_bwd = False
        lead = [None] * ctx.n_lead
        if last:
            lead = [plan.input_grads.get(i) if holder.in_requires_grad[i] else None for i in range(holder.n_in)]
        elif ctx.n_lead:
            lead = [torch.zeros((), device=plan.device)]                     # the next stage's token: runs that node
        if acc:
            pg = [None] * len(sg['params'])
        else:
            pg, offs = [], holder._param_offsets()
            for i in sg['params']:
                p = plan.params[i]
                pg.append(flat[offs[i]:offs[i] + p.numel()].view(p.shape) if p.requires_grad else None)
        return (None, None, None, None, *lead, *pg)


FLAT_ACCUMULATE = True


def _accumulate_in_place(holder, plan):
    """True when autograd would add this backward's parameter gradients, one kernel per parameter, into `.grad` tensors that are all
    views of the flat buffer of this holder's previous backward -- and nothing hangs on the per-parameter accumulation: no tensor
    hooks on the parameters, no stock DistributedDataParallel (its reducer hooks the gradient accumulators; with torch.distributed
    initialised the shortcut is only taken under this package's own GradReducer)."""
    buf = holder.__dict__.get('_accum_buf')
    if buf is None or not FLAT_ACCUMULATE:
        return False
    if torch.distributed.is_available() and torch.distributed.is_initialized() and holder.module.__dict__.get('_grad_reducer') is None:
        return False
    base, off = buf.data_ptr(), 0
    first = last = None
    for p in plan.params:
        n = p.numel()
        if p.requires_grad:
            g = p.grad
            if g is None or g.data_ptr() != base + off * 4 or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape:
                return False
            if p._backward_hooks or getattr(p, '_post_accumulate_grad_hooks', None):
                return False
            last = p
            first = first if first is not None else p
        off += n
    # `torch.autograd.grad(loss, params)` / `loss.backward(inputs=...)` RETURN gradients or accumulate into a subset: adding into
    # `.grad` would be a side effect there.  Inside a backward pass, ask the engine whether it will run the parameters'
    # AccumulateGrad nodes (first and last parameter; under autograd.grad() the query itself raises for a leaf)
    if first is not None and torch._C._current_graph_task_id() != -1:
        try:
            for q in (first, last):
                if not torch._C._will_engine_execute_node(torch.autograd.graph.get_gradient_edge(q).node):
                    return False
        except RuntimeError:
            return False
    return True


# eval-mode forwards (detect.py path) replay the launch list as ONE hipGraph (torch.cuda.graph == hipGraph on ROCm): the
# reference turns on cudnn.benchmark for streams (detect.py:115-124); here the static launch plan is captured instead.
# MYOLO_GRAPH=0 disables it; any capture failure falls back to the eager launch list.
import os as _os
GRAPH_EVAL = _os.environ.get('MYOLO_GRAPH', '1') != '0'
SPLIT_EVAL = True            # eval graphs: the side-stream branch is not joined inside the forward
# how the three pieces of a split eval forward are issued (r3k trace: the main queue sat idle ~140 us between graph A's last kernel and
# graph B's first one although B had been enqueued long before): 'g' = hipGraph replay, 'e' = the native launch program, eagerly
EVAL_TAIL = 'g'              # (round 5 sweep: every other placement measured slower) graph B (neck tail + Detect, main stream)
EVAL_HEAD = 'g'              # graph C (segmentation head, side stream)
EVAL_ORDER = 'cb'            # host order of the two launches behind graph A
# how the segmentation head's stream learns that the neck is done.  'event': hipEventRecord + hipStreamWaitEvent between graph A and graphs B / C --
# whichever launch came second on the main queue then started 90-170 us late (profiles/r6_infer_fork_*.txt; scripts/ubench/two_queue_gap.py shows
# the same for plain torch kernels, and no HIP / ROCr switch that changes it).  'sem' (round 6): ONE main-stream graph (chain up to the fork, a
# myolo_queue_post launch, neck tail + Detect) and the head's graph behind a myolo_queue_wait launch: a device-memory semaphore, no event
EVAL_FORK = 'sem'
EVAL_SEM_TIMEOUT_MS = 2000
EAGER_INPUT_OPS = True       # the un-joined eval graphs start BEHIND the launches that read the caller's tensors (no static input copy)
EVAL_SEM_CHECK_EVERY = 64    # frames between two reads of the semaphore's timeout word (a host sync)


class PlanHolder:
    """a built plan + its input/output binding for one (module, signature)."""

    def __init__(self, module, tensors, spec, dtype, training):
        dev = tensors[0].device
        self.plan = plan = E.Plan(dev, dtype, training)
        self.module = module
        self.n_in = len(tensors)
        self.in_requires_grad = [bool(t.requires_grad) for t in tensors]
        handles = []
        for t in tensors:
            slot = plan.add_input(t)
            if t.dim() != 4:
                raise L.MyoloError('module inputs are NCHW tensors')
            n, c, h, w = t.shape
            if c % E.SEG[dtype] != 0:
                if not t.is_contiguous():
                    raise L.MyoloError('image input must be contiguous NCHW')
                handles.append(E.ImageInput(slot, tuple(t.shape)))
            else:
                tv = plan.new(n, h, w, c, requires_grad=training and t.requires_grad)
                plan.add(E.ImportOp(plan, slot, tv))
                handles.append(tv)
        y = module.emit(plan, _unflatten(spec, handles))
        plan.n_emit_ops = len(plan.ops)              # ops from here on export the outputs (they may read a side-stream branch)
        self.ospec = _OutSpec(plan)
        self.out_spec = self.ospec.register(y)
        plan.register_params([p for p in module.parameters()])
        plan.build()
        self.pending_bwd = False
        self.generation = 0
        self.sig = None

    @staticmethod
    def param_sig(module):
        return tuple((p.data_ptr(), p.dtype) for p in module.parameters()) + \
            tuple((b.data_ptr(), b.dtype) for b in module.buffers())

    def run_graphed(self, tensors):
        """eval forward through a captured hipGraph: inputs are copied into static buffers, outputs are the plan's own."""
        st = self.__dict__
        if st.get('_graph_failed'):
            self.bind_inputs(tensors)
            self.plan.run_fwd()
            return self.output_tensors()
        if st.get('_graph') is None:
            n = st.get('_graph_warm', 0)
            if n < 2:                                            # two eager runs first (lazy allocations, prepare())
                st['_graph_warm'] = n + 1
                self.bind_inputs(tensors)
                self.plan.run_fwd()
                return self.output_tensors()
            try:
                self._static_in = [torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device) for t in tensors]
                for s_, t in zip(self._static_in, tensors):
                    s_.copy_(t)
                self.bind_inputs(self._static_in)
                torch.cuda.synchronize()
                split = self.plan.eval_split_progs() if SPLIT_EVAL else None
                g = torch.cuda.CUDAGraph()
                if split is None:
                    # one graph, ONE stream: a branch is captured in line (round 6: the two-stream capture of the same launch list -- fork and
                    # join as graph edges -- returned fp16 box coordinates up to 0.75 px away from the eager launch list in 34 of 40 frames,
                    # scripts/ubench/fork_stress.py FORK=joined; the shipped configurations never take this path, their head is un-joined)
                    side_was = self.plan.use_side_stream
                    self.plan.use_side_stream = False
                    try:
                        with torch.cuda.graph(g):
                            self.plan.run_fwd()
                    finally:
                        self.plan.use_side_stream = side_was
                else:
                    # three graphs: main chain up to the fork | rest of the main chain (neck tail, Detect) | the branch (segmentation head),
                    # replayed on the plan's side stream behind the fork and NOT joined: what the caller enqueues next on the main stream
                    # (detect.py: non_max_suppression) runs beside the head; the head's consumers wait for `_branch_done`
                    pm, fork, ps = split
                    side = self.plan._side_stream()
                    gb, gc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                    # the launches that read the CALLER's tensors (Focus' space-to-depth pack of the image, models/common.py:197-198) stay outside
                    # the graph and take the caller's pointer per frame: no copy into a static input buffer (9 us + a queue bubble per 1024x2048
                    # frame, and detect.py hands over a new tensor every frame).  They are a prefix of the program; anything else -> static inputs
                    ins = {id(c) for c in self.plan.in_ptr}
                    k0 = max([op + 1 for op, _i, cell, _t in pm.fixups if id(cell) in ins], default=0)
                    if not EAGER_INPUT_OPS or k0 > min(fork, 4):
                        k0 = 0
                    else:
                        self.bind_inputs(tensors)
                    st['_eager_head'] = k0
                    if EVAL_FORK == 'sem':
                        sem = st['_sem'] = torch.zeros(L.QUEUE_SEM_BYTES // 4, dtype=torch.int32, device=tensors[0].device)
                        torch.cuda.synchronize()
                        with torch.cuda.graph(g):
                            pm.run(k0, fork)
                            L.check(L.lib().myolo_queue_post(L.ptr(sem), L.stream_ptr()), 'myolo_queue_post')
                            pm.run(fork, pm.n)
                        with torch.cuda.graph(gc, stream=side):
                            L.check(L.lib().myolo_queue_wait(L.ptr(sem), EVAL_SEM_TIMEOUT_MS, L.stream_ptr()), 'myolo_queue_wait')
                            ps.run()
                        gb = None
                    else:
                        with torch.cuda.graph(g):
                            pm.run(k0, fork)
                        with torch.cuda.graph(gb):
                            pm.run(fork, pm.n)
                        with torch.cuda.graph(gc, stream=side):
                            ps.run()
                    st['_graph_b'], st['_graph_c'] = gb, gc
                    st['_split'] = split
                    st['_fork_ev'], st['_branch_done'] = torch.cuda.Event(), torch.cuda.Event()
                    st['_branch_pending'] = False
                st['_graph'] = g
            except Exception as e:                               # noqa: BLE001 -- capture is an optimisation only: eager launch list, loudly once
                import warnings
                warnings.warn(f'multiyolov5_amd: hipGraph capture of the eval forward failed ({e!r}); running the launch list eagerly '
                              '(bench.py reports graph_replayed: false)')
                st['_graph_failed'] = True
                self.bind_inputs(tensors)
                self.plan.run_fwd()
                return self.output_tensors()
        if st.get('_graph_c') is not None:
            main, side = torch.cuda.current_stream(), self.plan._side_stream()
            if st['_branch_pending'] or st.get('_branch_main_owes'):
                main.wait_event(st['_branch_done'])      # the previous frame's head may still be reading the neck features
                st['_branch_main_owes'] = False
            if st.get('_eager_head'):
                self.bind_inputs(tensors)
                st['_split'][0].run(0, st['_eager_head'])
            else:
                for s_, t in zip(self._static_in, tensors):
                    if s_.data_ptr() != t.data_ptr():
                        s_.copy_(t)
            if st.get('_sem') is not None:
                n = st['_sem_frames'] = st.get('_sem_frames', 0) + 1
                if n % EVAL_SEM_CHECK_EVERY == 0 and int(st['_sem'][32].item()):
                    st['_graph_failed'] = True
                    raise L.MyoloError('the segmentation head went on without the neck (myolo_queue_wait timed out): results of the last '
                                       f'{EVAL_SEM_CHECK_EVERY} frames are not valid; later forwards run the eager launch list')
                st['_graph'].replay()                    # post before wait in host order: correct even if both streams share one hardware queue
                with torch.cuda.stream(side):
                    st['_graph_c'].replay()
                    st['_branch_done'].record(side)
                st['_branch_pending'] = True
                st['_branch_main'] = main
                return self.output_tensors()
            st['_graph'].replay()
            st['_fork_ev'].record(main)
            pm, fork, ps = st['_split']
            for which in EVAL_ORDER:
                if which == 'c':
                    side.wait_event(st['_fork_ev'])
                    with torch.cuda.stream(side):
                        if EVAL_HEAD == 'e':
                            ps.run()
                        else:
                            st['_graph_c'].replay()
                        st['_branch_done'].record(side)
                    st['_branch_pending'] = True
                    st['_branch_main'] = main
                elif EVAL_TAIL == 'e':
                    pm.run(fork, pm.n)
                else:
                    st['_graph_b'].replay()
            return self.output_tensors()
        for s_, t in zip(self._static_in, tensors):
            if s_.data_ptr() != t.data_ptr():
                s_.copy_(t)
        st['_graph'].replay()
        return self.output_tensors()

    def wait_branch(self):
        """the main stream waits for the un-joined branch of the last eval forward (no-op without one)"""
        st = self.__dict__
        if st.get('_branch_pending'):
            cur = torch.cuda.current_stream()
            cur.wait_event(st['_branch_done'])
            st['_branch_pending'] = False
            # a consumer on ANOTHER stream waited, the launch stream of the next forward has not: it still owes the wait
            st['_branch_main_owes'] = cur != st.get('_branch_main', cur)

    def bind_inputs(self, tensors):
        for i, t in enumerate(tensors):
            self.plan.in_ptr[i].value = t.data_ptr()

    def output_tensors(self):
        outs = []
        for s in range(self.ospec.nslots):
            if s in self.ospec.det_slots:
                outs.append(self.ospec.det_slots[s].det_out)
            else:
                outs.append(self.plan.outputs[s])
        return outs

    def _param_offsets(self):
        offs = self.__dict__.get('_poffs')
        if offs is None:
            offs, off = [], 0
            for p in self.plan.params:
                offs.append(off)
                off += p.numel()
            self._poffs = offs
        return offs

    def output_grad_tensor(self, s):
        if s in self.ospec.det_slots:
            return self.ospec.det_slots[s].gdet
        return self.plan.output_grads[s]


class PlannedModule(nn.Module):
    """nn.Module whose forward is a libmyolo launch plan.  Subclasses implement emit(plan, x).

    Host cost matters (a training step is ~700 launches): the parameter / buffer tensors are listed once per module and a
    forward only re-reads their data pointers (re-allocation by .to()/.half()/load_state_dict is detected) and, in eval mode,
    their version counters (in-place weight edits re-run the one-off epilogue constant preparation).  Replacing a Parameter
    OBJECT after the first forward is not detected: call invalidate_plans() (Model.fuse() does).

    Outputs are views of plan-owned static buffers (that is what makes a forward allocation-free and hipGraph-replayable): the
    next forward of the same module with the same input signature overwrites them in place.  Callers that keep results across
    forwards (accumulated predictions, one module used twice inside an Ensemble) must clone them -- models.experimental.Ensemble
    does.  A backward through activations that a newer forward has overwritten raises (forward generation counter)."""

    _RUNTIME_STATE = ('_plans', '_tensor_list', '_prepared_version', '_grad_reducer', '_has_sync_bn')

    def __getstate__(self):
        """pickling / deepcopy (checkpoints, ModelEMA) carries the module, not its launch plans (device buffers, ctypes descriptors)"""
        d = self.__dict__.copy()
        for k in self._RUNTIME_STATE:
            d.pop(k, None)
        return d

    def _tensors(self):
        ts = self.__dict__.get('_tensor_list')
        if ts is None:
            ts = self.__dict__['_tensor_list'] = list(self.parameters()) + list(self.buffers())
        return ts

    def _sig(self):
        return tuple(t.data_ptr() for t in self._tensors())

    # Re-allocation of parameters / buffers happens through nn.Module._apply (.to / .half / .float / .cuda) or load_state_dict(assign=True):
    # every PlannedModule bumps ONE process-wide epoch there, and a forward re-reads the ~170-350 data pointers only when the epoch moved
    # since its plan last checked them (round 4: 16 us of host time per detect.py frame, profiles/r3k_infer_timeline.md).  A `.half()`
    # called directly on a LEAF nn.Conv2d / nn.BatchNorm2d inside the model is not seen (like replacing a Parameter object): invalidate_plans().
    _EPOCH = [0]

    def _apply(self, fn, *args, **kwargs):
        r = super()._apply(fn, *args, **kwargs)
        PlannedModule._EPOCH[0] += 1
        return r

    def load_state_dict(self, *args, **kwargs):
        r = super().load_state_dict(*args, **kwargs)
        PlannedModule._EPOCH[0] += 1
        return r

    def _holder(self, tensors, spec):
        for t in tensors:
            L.require_gpu(t)
        dtype = compute_dtype(tensors)
        grad = torch.is_grad_enabled() and self.training
        key = (tuple((tuple(t.shape), tuple(t.stride()), t.dtype, bool(t.requires_grad and grad)) for t in tensors),
               str(spec), dtype, bool(self.training), grad, self._sync_world())
        plans = self.__dict__.setdefault('_plans', {})
        h = plans.get(key)
        epoch = PlannedModule._EPOCH[0]
        if h is not None and h.__dict__.get('_sig_epoch') == epoch and self._sentinels_match(h.sig):
            sig = h.sig                                  # nothing re-allocated since this plan compared the pointers
        else:
            sig = self._sig()
            if h is not None:
                h.__dict__['_sig_epoch'] = epoch
        if h is None or h.sig != sig:
            # module.training decides BatchNorm batch statistics (and builds the backward launch list);
            # `grad` only decides whether autograd is wired through PlanFn
            self.__dict__.pop('_tensor_list', None)
            sig = self._sig()
            h = PlanHolder(self, tensors, spec, dtype, bool(self.training))
            h.sig = sig
            h.__dict__['_sig_epoch'] = epoch
            plans[key] = h
            self.__dict__['_prepared_version'] = self._param_version()
        elif not self.training:
            v = self._param_version()
            if self.__dict__.get('_prepared_version') != v:
                h.plan.prepare()
                h.__dict__['_graph'] = None             # host-side launch constants (Detect anchors) are baked into a capture
                h.__dict__['_graph_warm'] = 0
                self.__dict__['_prepared_version'] = v
        return h, grad

    def _sentinels_match(self, sig):
        """a re-allocation that bypasses _apply / load_state_dict (`.half()` on a leaf nn.Conv2d, `p.data = ...`) does not move the
        epoch; the fast path still compares a handful of data pointers spread over the list (first, last, every 16th: ~1 us) so that a
        whole-submodule conversion is caught and the plan rebuilt instead of launching on freed storage"""
        ts = self._tensors()
        n = len(ts)
        if n != len(sig):
            return False
        for i in range(0, n, 16):
            if ts[i].data_ptr() != sig[i]:
                return False
        return n == 0 or ts[-1].data_ptr() == sig[-1]

    def _param_version(self):
        return sum(t._version for t in self._tensors())

    def _sync_world(self):
        """ranks a training plan's nn.SyncBatchNorm layers exchange statistics with (part of the plan key: a plan built before
        init_process_group counts this GPU's samples only); 0 for models without SyncBatchNorm and for eval"""
        has = self.__dict__.get('_has_sync_bn')
        if has is None:
            has = self.__dict__['_has_sync_bn'] = any(isinstance(m, nn.SyncBatchNorm) for m in self.modules())
        if not (has and self.training and torch.distributed.is_available() and torch.distributed.is_initialized()):
            return 0
        g = next(m.process_group for m in self.modules() if isinstance(m, nn.SyncBatchNorm))
        return torch.distributed.get_world_size(g)

    def forward(self, x):
        tensors = []
        spec = _flatten(x, tensors)
        h, grad = self._holder(tensors, spec)
        if grad:
            stages = _stage_plan(h)
            if stages is None:
                outs = PlanFn.apply(h, *tensors, *h.plan.params)
            else:
                n, params = len(stages), h.plan.params
                tok = PlanStageFn.apply(h, n - 1, n, len(tensors), *tensors, *[params[i] for i in stages[n - 1]['params']])
                for k in range(n - 2, 0, -1):
                    tok = PlanStageFn.apply(h, k, n, 1, tok, *[params[i] for i in stages[k]['params']])
                outs = PlanStageFn.apply(h, 0, n, 1, tok, *[params[i] for i in stages[0]['params']])
        elif not self.training and GRAPH_EVAL:
            outs = self._eval_outputs(h, h.run_graphed(tensors))
        else:
            h.bind_inputs(tensors)
            h.plan.run_fwd()
            if not self.training:
                return _OutSpec.rebuild(h.out_spec, list(self._eval_outputs(h, h.output_tensors())))
            if self.training:
                # train mode without autograd (`with torch.no_grad(): model(x)`: BatchNorm recalibration, train-mode validation): the
                # deferred x8 upsample of the logits has no LazySegLogits wrapper to trigger it -- run it now
                st = L.stream_ptr()
                for op in h.plan.ops:
                    lazy = getattr(op, 'lazy_call', None)
                    if lazy is not None:
                        lazy(st)
                h.generation += 1
            outs = h.output_tensors()
        return _OutSpec.rebuild(h.out_spec, list(outs))

    @staticmethod
    def _eval_outputs(h, outs):
        """eval forward: an output whose last launch is deferred (the x8 upsample of the class logits, engine.SegOutOp) goes out as a
        LazySegLogits of THIS forward -- detect.py's resize + argmax (utils.general.seg_argmax) never needs the full-resolution values,
        anything else materialises them on first use"""
        if not any(o.__dict__.get('_myolo_lazy') is not None for o in outs):
            h.wait_branch()
            return outs
        h.generation += 1
        outs = PlanFn._wrap_outputs(h)
        for o in outs:
            if isinstance(o, LazySegLogits):
                o._myolo_holder = h                      # utils.general.seg_argmax: wait for the un-joined head before reading its logits
        return outs

    def invalidate_plans(self):
        self.__dict__.pop('_plans', None)
        self.__dict__.pop('_tensor_list', None)
        self.__dict__.pop('_has_sync_bn', None)        # (convert_sync_batchnorm after a first forward: call this)
