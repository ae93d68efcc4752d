"""CPU: host logic added in round 2 that needs no GPU -- the launch-list structure of a training plan (merged entry convs, grouped
PyramidPooling passes, the fused-loss switch), the lazily materialised training logits and the flat gradient accumulation rule."""
import os
from collections import Counter
from types import SimpleNamespace

import pytest
import torch

from tests.util import CFG, TAGS


def _plan(tag='s_psp', training=True, dt=torch.float16):
    from multiyolov5_amd import runtime as R
    from multiyolov5_amd.models.yolo import Model
    m = Model(os.path.join(CFG, TAGS[tag]))
    m.train(training)
    return R.PlanHolder(m, [torch.zeros(2, 3, 64, 128)], ('t', 0), dt, training).plan


def test_training_plan_launch_list_structure(monkeypatch):
    from multiyolov5_amd import engine as E
    # round 6: at this size every BatchNorm backward whose sums are not produced by a dgrad epilogue is ONE launch (reduce + barrier + apply)
    plan = _plan()
    bwd = Counter(c.name for op in plan.ops for c in op.bwd_calls)
    assert bwd['myolo_bn_act_bwd_fused'] >= 9 and bwd['myolo_bn_act_bwd_reduce_split'] == 0 and bwd['myolo_bn_act_bwd_apply_split'] == 0
    assert hasattr(plan, '_grid_bar') and plan._grid_bar.numel() == 19 * 32
    monkeypatch.setattr(E, 'BN_BWD_FUSED', 0)                       # the two-launch form (what the big maps of the benchmarked step still run)
    monkeypatch.setattr(E, 'CONV_BN_ACT', False)                    # ... and conv / BatchNorm forward as two launches
    plan = _plan()
    fwd = Counter(c.name for op in plan.ops for c in op.fwd_calls)
    bwd = Counter(c.name for op in plan.ops for c in op.bwd_calls)
    assert bwd['myolo_bn_act_bwd_fused'] == 0
    # 8 C3 blocks + RFB2: two same-input 1x1 Conv+BN+SiLU as one convolution with split BatchNorm parameters (common.py:137,500-501)
    assert fwd['myolo_bn_act_fwd_split'] == 9 and bwd['myolo_bn_act_bwd_apply_split'] == 9 and bwd['myolo_bn_act_bwd_reduce_split'] == 9
    merged = [op for op in plan.ops if isinstance(op, E.ConvOp) and op.weight2 is not None]
    assert len(merged) == 9 and all(op.cout == op.weight.shape[0] + op.weight2.shape[0] for op in merged)
    # PyramidPooling: four pools / four upsamples, one backward pass each, one forward launch for the upsamples (common.py:521-537)
    assert fwd['myolo_pyramid_upsample_fwd'] == 1 and bwd['myolo_pyramid_upsample_bwd'] == 1 and bwd['myolo_adaptive_avgpool_bwd_multi'] == 1
    # (8x16 maps are too narrow for the one-pass pyramid pools: four single launches; FFM's lone global pool takes the one-pass kernel)
    assert fwd['myolo_adaptive_avgpool_fwd'] + fwd['myolo_adaptive_avgpool_fwd_multi'] == 5 and bwd['myolo_adaptive_avgpool_bwd'] == 1
    # the x8 upsample of the logits is deferred; its backward is chosen per step (full-resolution gradient | low-resolution fused CE)
    seg = [op for op in plan.ops if isinstance(op, E.SegOutOp)]
    assert len(seg) == 1 and seg[0].lazy_call is not None and fwd['myolo_seg_upsample_fwd'] == 0
    sw = [c for c in seg[0].bwd_calls if isinstance(c, E.SwitchCall)]
    assert len(sw) == 1 and sw[0].a.name == 'myolo_seg_upsample_bwd' and sw[0].b.name == 'myolo_seg_lowgrad_apply'
    # weight gradients: one launch per weight tensor (two for a merged conv), all on the side stream
    wg = [c for op in plan.ops for c in op.bwd_calls if c.name == 'myolo_conv_wgrad']
    assert all(c.side for c in wg) and len(wg) == sum(2 if op.weight2 is not None else 1 for op in plan.ops if isinstance(op, E.ConvOp))
    # one weight repack per forward, tiled
    assert plan._pack_call is not None and plan._pack_call.args[3] == -9      # tiled mode: -(largest tap count), myolo.h


def test_eval_plan_defers_the_logit_upsample_and_merges_too(monkeypatch):
    from multiyolov5_amd import engine as E
    plan = _plan(training=False)
    fwd = Counter(c.name for op in plan.ops for c in op.fwd_calls)
    seg = [op for op in plan.ops if isinstance(op, E.SegOutOp)]
    # detect.py's resize + argmax reads the low-resolution logits (utils.general.seg_argmax): the x8 upsample is deferred in eval too
    assert fwd['myolo_seg_upsample_fwd'] == 0 and seg[0].lazy_call is not None and fwd['myolo_pyramid_upsample_fwd'] == 1
    monkeypatch.setattr(E, 'LAZY_SEG_EVAL', False)
    plan = _plan(training=False)
    fwd = Counter(c.name for op in plan.ops for c in op.fwd_calls)
    assert fwd['myolo_seg_upsample_fwd'] == 1 and all(op.lazy_call is None for op in plan.ops if isinstance(op, E.SegOutOp))
    assert sum(1 for op in plan.ops if isinstance(op, E.ConvOp) and op.weight2 is not None) == 9
    assert not any(op.bwd_calls for op in plan.ops)


def test_lazy_logits_materialise_on_first_real_use_only():
    from multiyolov5_amd import _lib as L
    from multiyolov5_amd import runtime as R
    launches, joins = [], []
    holder = SimpleNamespace(generation=3, wait_branch=lambda: joins.append(len(launches)))   # (eval: the un-joined head is waited for first)
    op = SimpleNamespace(lazy_call=lambda st: launches.append(st))
    base = torch.arange(24, dtype=torch.float32).reshape(1, 2, 3, 4)
    t = base.detach().as_subclass(R.LazySegLogits)
    t._myolo_lazy_state = {'holder': holder, 'op': op, 'generation': 3, 'done': False}
    orig = L.stream_ptr
    L.stream_ptr = lambda: 'stream'
    try:
        assert tuple(t.shape) == (1, 2, 3, 4) and t.dtype == torch.float32 and t.dim() == 4 and t.stride() == (24, 12, 4, 1)
        assert t.numel() == 24 and t.is_contiguous() and not t.requires_grad and t.device.type == 'cpu'
        assert launches == []                                         # metadata only: nothing launched
        s = t.sum()
        assert joins == [0]                                           # ... after the branch was joined, before the launch
        assert launches == ['stream'] and type(s) is torch.Tensor and float(s) == float(base.sum())
        (t * 2).float()
        assert launches == ['stream']                                 # once
        # a tensor of an older forward refuses to serve values
        u = base.detach().as_subclass(R.LazySegLogits)
        u._myolo_lazy_state = {'holder': holder, 'op': op, 'generation': 2, 'done': False}
        assert tuple(u.shape) == (1, 2, 3, 4)
        with pytest.raises(L.MyoloError):
            u + 1
    finally:
        L.stream_ptr = orig


def test_flat_accumulation_rule():
    from multiyolov5_amd import runtime as R
    ps = [torch.nn.Parameter(torch.zeros(2, 3)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(4), requires_grad=False)]
    plan = SimpleNamespace(params=ps)
    holder = SimpleNamespace(module=SimpleNamespace())
    assert R._accumulate_in_place(holder, plan) is False              # no earlier hand-out
    buf = torch.zeros(15)
    holder._accum_buf = buf
    assert R._accumulate_in_place(holder, plan) is False              # .grad is None (zero_grad(set_to_none=True)): autograd assigns
    ps[0].grad, ps[1].grad = buf[0:6].view(2, 3), buf[6:11].view(5)
    assert R._accumulate_in_place(holder, plan) is True
    ps[1].grad = torch.zeros(5)                                       # replaced by the caller
    assert R._accumulate_in_place(holder, plan) is False
    ps[1].grad = buf[6:11].view(5)
    h = ps[0].register_hook(lambda g: g)                              # a tensor hook must see its per-parameter gradient
    assert R._accumulate_in_place(holder, plan) is False
    h.remove()
    assert R._accumulate_in_place(holder, plan) is True
    # inside a backward pass: only when the engine will run the parameters' AccumulateGrad nodes -- not under autograd.grad(), which
    # RETURNS gradients, nor under backward(inputs=subset)
    seen = []

    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a, b, c):
            return a.sum() + b.sum() + c.sum()

        @staticmethod
        def backward(ctx, g):
            seen.append(R._accumulate_in_place(holder, plan))
            return torch.ones(2, 3) * g, torch.ones(5) * g, None
    y = F.apply(*ps)
    y.backward(retain_graph=True)
    ps[0].grad, ps[1].grad = buf[0:6].view(2, 3), buf[6:11].view(5)
    torch.autograd.grad(y, ps[:2], retain_graph=True)
    y.backward(inputs=[ps[0]], retain_graph=True)
    assert seen == [True, False, False]


def test_sync_batchnorm_without_a_process_group_is_plain_batchnorm():
    """train.py:190-193 --sync-bn converts the mirror's BatchNorm2d modules.  Like torch's SyncBatchNorm, without an initialised process
    group (world 1) the layer IS BatchNorm2d: same launch list, no collective (the 2-rank form: tests/test_parallel_cpu.py)"""
    from multiyolov5_amd import engine as E
    from multiyolov5_amd import runtime as R
    from multiyolov5_amd.models.yolo import Model
    m = Model(os.path.join(CFG, TAGS['s_psp']))
    ref = _plan()
    m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(m).train()
    assert any(isinstance(x, torch.nn.SyncBatchNorm) for x in m.modules())
    plan = R.PlanHolder(m, [torch.zeros(2, 3, 64, 128)], ('t', 0), torch.float16, True).plan
    assert not plan.has_sync()
    names = lambda pl, which: [c.name for op in pl.ops for c in getattr(op, which)]
    assert names(plan, 'fwd_calls') == names(ref, 'fwd_calls') and names(plan, 'bwd_calls') == names(ref, 'bwd_calls')


def test_native_program_walks_ranges_around_sync_points():
    """NativeProg.run cuts its record range at the host-side collectives: each runs exactly once, before the record it precedes, also
    when callers walk the program in contiguous pieces (staged backward, gradient slices)"""
    from multiyolov5_amd import engine as E
    log = []
    np_ = E.NativeProg.__new__(E.NativeProg)
    np_.switches, np_.slots, np_.n, np_.handle = [], [], 10, None
    np_._run = lambda first, last, st, side: log.append((first, last))
    mk = lambda k: (lambda st=None: log.append(k))
    np_.syncs = [(0, mk('a')), (4, mk('b')), (4, mk('c')), (7, mk('d')), (10, mk('e'))]
    orig = torch.cuda.current_stream
    torch.cuda.current_stream = lambda: SimpleNamespace(cuda_stream=0)
    try:
        np_.run()
        assert log == ['a', (0, 4), 'b', 'c', (4, 7), 'd', (7, 10), 'e']
        for cuts in ([0, 4, 10], [0, 3, 7, 10], [0, 5, 6, 10], [0, 7, 7, 10]):
            del log[:]
            for f, l in zip(cuts, cuts[1:]):
                np_.run(f, l)
            assert [x for x in log if isinstance(x, str)] == ['a', 'b', 'c', 'd', 'e'], (cuts, log)
            recs = [x for x in log if isinstance(x, tuple)]
            assert recs[0][0] == 0 and recs[-1][1] == 10 and all(a[1] == b[0] for a, b in zip(recs, recs[1:])), (cuts, log)
            for k, idx in (('b', 4), ('d', 7)):            # the collective sits between the records before and after its index
                i = log.index(k)
                assert all(x[1] <= idx for x in log[:i] if isinstance(x, tuple)) and all(x[0] >= idx for x in log[i:] if isinstance(x, tuple))
        np_.syncs = []
        del log[:]
        np_.run(2, 5)
        assert log == [(2, 5)]
    finally:
        torch.cuda.current_stream = orig
        np_.handle = None


@pytest.mark.parametrize('tag', ['s_psp', 's_base', 's_lab', 's_bise', 'm_lab'])
def test_pruned_backward_schedules_are_self_consistent(tag):
    """train.py:371 / 392 -- a backward from the detection outputs only, or from the segmentation output(s) only, runs a sub-list of the
    static backward (engine.Plan.bwd_schedule).  Dry plan: every gradient range a scheduled launch reads or accumulates into has an
    earlier writer in the same schedule; the checker itself is validated on the full list and on schedules with a link removed."""
    from multiyolov5_amd import runtime as R
    from multiyolov5_amd.models.yolo import Model
    m = Model(os.path.join(CFG, TAGS[tag])).train()
    h = R.PlanHolder(m, [torch.zeros(2, 3, 64, 128)], ('t', 0), torch.float16, True)
    plan, ns = h.plan, h.ospec.nslots
    det = frozenset(h.ospec.det_slots)
    seg = frozenset(range(ns)) - det
    assert len(det) == 3 and len(seg) >= 1
    full = [('op', i) for i in range(len(plan.ops) - 1, -1, -1)]
    assert plan.check_bwd_schedule(full) == []
    assert plan.bwd_schedule(det | seg) == 'full'
    nfull = sum(len(op.bwd_calls) for op in plan.ops)
    for live in (det, seg):
        sched = plan.bwd_schedule(live)
        assert isinstance(sched, list) and plan.check_bwd_schedule(sched) == []
        kept = [e[1] for e in sched if e[0] == 'op']
        n = sum(len(plan.ops[i].bwd_calls) for i in kept)
        assert kept == sorted(kept, reverse=True) and 0.5 * nfull < n < 0.95 * nfull      # a real sub-list, in backward order
        # the output ops of the other head do not run, those of this head do
        for i, op in enumerate(plan.ops):
            s_ = plan._out_slot(op)
            if s_ is not None:
                assert (i in kept) == (s_ in live)
        # every pruned launch list still writes each parameter gradient at most through launches of running ops: the weight-gradient
        # launches kept are exactly those of the running convolutions
        items = plan._bwd_items(None, live)
        wg = sum(1 for it in items if getattr(it, 'name', '') == 'myolo_conv_wgrad')
        assert wg == sum(1 for i in kept for c in plan.ops[i].bwd_calls if c.name == 'myolo_conv_wgrad')
        # mutations: drop a zero fill / a restored reduce / a running op that feeds a running producer -> the checker objects
        fills = [e for e in sched if e[0] == 'fill']
        assert fills, 'a pruned op was the first writer of a range a running producer reads'
        for e in fills:
            assert plan.check_bwd_schedule([x for x in sched if x is not e]), e
        stores = {id(c[0]) for c in plan._claim_log if c[4] == 0}           # ops that are the FIRST writer of a gradient range
        mid = [e for e in sched if e[0] == 'op' and id(plan.ops[e[1]]) in stores and plan._out_slot(plan.ops[e[1]]) is None]
        for e in mid[::7]:
            assert plan.check_bwd_schedule([x for x in sched if x is not e]), e
    # a BatchNorm layer whose reduce pass rides in the dgrad epilogue of a pruned convolution gets its own reduce launch back
    from multiyolov5_amd import engine as E
    pos = {id(op): i for i, op in enumerate(plan.ops)}
    for live in (det, seg):
        lv = plan.bwd_liveness(live)
        sched = plan.bwd_schedule(live)
        want = {i for i, op in enumerate(plan.ops) if lv[i] and getattr(op, 'reduce_by', None) is not None and not lv[pos[id(op.reduce_by)]]}
        assert {e[1] for e in sched if e[0] == 'reduce'} == want
        names = [it.name if hasattr(it, 'name') else it[0] for it in plan._bwd_items(None, live)]
        assert names.count('myolo_bn_act_bwd_reduce') == sum(1 for e in sched if e[0] == 'op' for c in plan.ops[e[1]].bwd_calls
                                                             if c.name == 'myolo_bn_act_bwd_reduce') + len(want)


def test_staged_backward_marks_address_pruned_programs():
    """the stock-DDP stage chain cuts the native program at segment marks: a pruned program has the same mark keys at other indices"""
    from multiyolov5_amd import runtime as R
    from multiyolov5_amd.models.yolo import Model
    m = Model(os.path.join(CFG, TAGS['s_psp'])).train()
    h = R.PlanHolder(m, [torch.zeros(2, 3, 64, 128)], ('t', 0), torch.float16, True)
    plan = h.plan
    cuts = R.StageCuts()

    def marks(items):
        out, n = {}, 0
        for it in items:
            if isinstance(it, tuple) and it[0] == 'mark':
                out[it[1]] = n
            else:
                n += 2 if type(it).__name__ == 'SwitchCall' else 1
        return out, n
    mf, nf = marks(plan._bwd_items(cuts))
    md, nd = marks(plan._bwd_items(cuts, frozenset(h.ospec.det_slots)))
    assert set(mf) == set(md) and nd < nf and all(md[k] <= mf[k] for k in mf)
    assert [md[k] for k in sorted(md)] == sorted(md[k] for k in md)


def test_tiny_conv_groups_leave_the_launch_chain(monkeypatch):
    """MYOLO_TINY_CONV=1: PyramidPooling's four branch Conv+BatchNorm+SiLU layers (common.py:521-537) become ONE forward and ONE backward
    launch (csrc/tiny_conv.hip) at the last branch; dependent tiny layers (FFM's two attention convs, common.py:218-224) stay separate
    launches; the pruned one-loss schedules remain valid; fp32 plans, whose four upsamples are not one launch, do not group"""
    from multiyolov5_amd import engine as E
    monkeypatch.setattr(E, 'TINY_CONV', True)
    monkeypatch.setattr(E, 'BN_BWD_FUSED', 0)        # (the launch counts below are those of the two-pass BatchNorm backward)
    monkeypatch.setattr(E, 'CONV_BN_ACT', False)     # (... and of the two-launch Conv + BatchNorm forward)
    from multiyolov5_amd import runtime as R
    from multiyolov5_amd.models.yolo import Model

    def _plan(dt=torch.float16):           # the bench batch (BASELINE configs[1]) at 256 x 512: the smallest backbone map is 16 x 8 x 16 = 2048
        # pixels, above the tiny kernels' 1024 like at 512 x 1024 (at 64 x 128 most of the network is "tiny"); a quarter of the host memory
        import gc
        gc.collect()
        m = Model(os.path.join(CFG, TAGS['s_psp'])).train()
        return R.PlanHolder(m, [torch.zeros(16, 3, 256, 512)], ('t', 0), dt, True)
    holder = _plan()
    plan = holder.plan
    groups, seen = [], set()
    for op in plan.ops:
        if isinstance(op, E.ConvOp) and op.group and id(op.group) not in seen:
            seen.add(id(op.group))
            groups.append(op.group)
    sizes = sorted(len(g) for g in groups)
    assert sizes == [1, 1, 4], sizes
    psp = next(g for g in groups if len(g) == 4)
    assert [o.x.h for o in psp] == [1, 2, 3, 6] and all(o.bn is not None and o.cout == 32 for o in psp)
    fwd = Counter(c.name for op in plan.ops for c in op.fwd_calls)
    bwd = Counter(c.name for op in plan.ops for c in op.bwd_calls)
    assert fwd['myolo_tiny_conv_fwd'] == 3 and bwd['myolo_tiny_conv_bwd'] == 3
    # only the last member launches; every member keeps its own weight-gradient launch; no member is a BatchNorm-sum carrier or target
    for g in groups:
        for o in g:
            names = [c.name for c in o.bwd_calls]
            assert names.count('myolo_conv_wgrad') == 1 and (('myolo_tiny_conv_bwd' in names) == (o is g[-1]))
            assert [c.name for c in o.fwd_calls] == (['myolo_tiny_conv_fwd'] if o is g[-1] else [])
            assert o.reduce_by is None and not o.bnb_targets
            assert not any(n.startswith('myolo_bn_act') or n == 'myolo_conv' for n in names)
    # the launch of a group sits behind every member's zero fills and the members are pruned / kept together
    pos = {id(op): i for i, op in enumerate(plan.ops)}
    assert all(pos[id(g[-1])] == max(pos[id(o)] for o in g) for g in groups)
    det = frozenset(holder.ospec.det_slots)
    for live in (det, frozenset(range(len(holder.ospec.det_slots) + 1)) - det):
        sched = plan.bwd_schedule(live)
        assert isinstance(sched, list) and plan.check_bwd_schedule(sched) == []
        kept = {e[1] for e in sched if e[0] == 'op'}
        for g in groups:
            assert len({pos[id(o)] in kept for o in g}) == 1
    # the same model with the flag off has 9 forward / 13 backward launches more (8 -> 1 and 12 -> 1 for the pyramid, 2 x (2 -> 1) each way for FFM)
    nf, nb = sum(len(o.fwd_calls) for o in plan.ops), sum(len(o.bwd_calls) for o in plan.ops)
    del holder, plan, groups, psp, sched, op, g, o, pos, seen
    monkeypatch.setattr(E, 'TINY_CONV', False)
    off = _plan().plan
    assert sum(len(o.fwd_calls) for o in off.ops) - nf == 9 and sum(len(o.bwd_calls) for o in off.ops) - nb == 13
    del off
    monkeypatch.setattr(E, 'TINY_CONV', True)
    p32 = _plan(dt=torch.float32).plan
    assert sorted(len(o.group) for o in p32.ops if isinstance(o, E.ConvOp) and o.group) == [1] * 6


@pytest.mark.parametrize('res', [(2, 64, 128), (16, 128, 256)], ids=['2x64x128', '16x128x256'])
@pytest.mark.parametrize('tag', list(TAGS))
def test_every_tiny_conv_launch_passes_the_librarys_host_checks(tag, res, monkeypatch):
    """engine.ConvOp.tiny_ok mirrors tiny_check (csrc/tiny_conv.hip): every descriptor array a plan builds is accepted by the entry
    points' host-side validation (no GPU: an accepted call fails later, at the launch, with a HIP error -- never MYOLO_EINVAL); no two
    members of a group read the same tensor (their input gradients would race) or depend on each other.  The tiny layers' shapes depend on
    the batch size and the channel counts only (pyramid maps are 1x1 .. 6x6 per image whatever the input resolution), so the batch-16 case
    runs at 128x256: a dry plan of 16x3x512x1024 is 17 GB of host memory (round 4: the one-process CPU suite was OOM-killed at 65 GB)"""
    import gc
    from multiyolov5_amd import engine as E, runtime as R, _lib as L
    from multiyolov5_amd.models.yolo import Model
    monkeypatch.setattr(E, 'TINY_CONV', True)
    lib = L.lib()
    m = Model(os.path.join(CFG, TAGS[tag])).train()
    for dt in (torch.float16, torch.float32):
        plan = R.PlanHolder(m, [torch.zeros(res[0], 3, res[1], res[2])], ('t', 0), dt, True).plan
        n = 0
        for op in plan.ops:
            for calls, fn in ((op.fwd_calls, 'myolo_tiny_conv_fwd'), (op.bwd_calls, 'myolo_tiny_conv_bwd')):
                for c in calls:
                    if getattr(c, 'name', '') == fn:
                        n += 1
                        assert getattr(lib, fn)(c.args[0], c.args[1], None) != L.EINVAL, (tag, dt, fn, op.cin, op.cout)
            g = op.group if isinstance(op, E.ConvOp) else None
            if g and op is g[-1]:
                def overlap(a, b):
                    return a.buf is b.buf and a.coff < b.coff + b.c and b.coff < a.coff + a.c
                for i, a in enumerate(g):
                    for b in g[i + 1:]:
                        assert not overlap(a.x, b.x) and not overlap(b.x, a.out) and not overlap(a.out, b.out)
        assert n % 2 == 0
        del plan, op, calls, c, g
        gc.collect()


class _AtenLog(torch.utils._python_dispatch.TorchDispatchMode):
    """every ATen op that reaches the dispatcher while the mode is active"""

    def __init__(self):
        super().__init__()
        self.ops = []

    muted = False            # (the stand-in for the library's kernel computes with ATen: not part of what is checked)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        if not _AtenLog.muted:
            self.ops.append(str(func))
        return func(*args, **(kwargs or {}))


def _lazy_logits(base, launches):
    from multiyolov5_amd import runtime as R
    holder = SimpleNamespace(generation=1, wait_branch=lambda: None)
    op = SimpleNamespace(lazy_call=lambda st: launches.append('upsample'))
    t = base.detach().as_subclass(R.LazySegLogits)
    t._myolo_lazy_state = {'holder': holder, 'op': op, 'generation': 1, 'done': False}
    return t


def test_unchanged_detect_and_test_py_statements_reach_the_fused_argmax(monkeypatch):
    """detect.py:191-193 and test.py:38 + utils/metrics.py:240,259 as the reference writes them: `F.interpolate(seg, size, mode='bilinear',
    align_corners=True)` of the model's lazy logits is a deferred view, `[0]`, `.data`, `.max(axis=0)[1]` / `torch.max(output, 1)` on it
    call the fused resize + arg-max ONCE and never run ATen's upsample / max; any other use computes the reference's values"""
    import torch.nn.functional as F
    from multiyolov5_amd import _lib as L
    from multiyolov5_amd import runtime as R
    from multiyolov5_amd.utils import general as G
    calls = []

    def fake_seg_argmax(seg, h0=None, w0=None, out_dtype=torch.int64):
        calls.append((h0, w0))
        _AtenLog.muted = True
        try:
            with torch._C.DisableTorchFunctionSubclass():
                return F.interpolate(seg.as_subclass(torch.Tensor), (h0, w0), mode='bilinear', align_corners=True).argmax(1)
        finally:
            _AtenLog.muted = False
    monkeypatch.setattr(G, 'seg_argmax', fake_seg_argmax)
    monkeypatch.setattr(L, 'stream_ptr', lambda: 'stream')
    torch.manual_seed(0)
    base = torch.randn(2, 19, 8, 16)
    ref = F.interpolate(base, (24, 40), mode='bilinear', align_corners=True)
    # --- detect.py:191-193
    launches = []
    seg = _lazy_logits(base, launches)
    with _AtenLog() as log:
        seg = F.interpolate(seg, (24, 40), mode='bilinear', align_corners=True)[0]
        assert isinstance(seg, R.LazyResized) and tuple(seg.shape) == (19, 24, 40) and seg.dtype == torch.float32
        lab = seg.max(axis=0)[1]
    assert type(lab) is torch.Tensor and torch.equal(lab, ref[0].argmax(0)) and calls == [(24, 40)] and launches == []
    assert not any('upsample' in o or 'max' in o for o in log.ops), log.ops
    # --- test.py:38 + metrics.py:240,259: two torch.max over the same resized prediction -> one launch
    calls.clear()
    pred = F.interpolate(_lazy_logits(base, launches), (24, 40), mode='bilinear', align_corners=True)
    with _AtenLog() as log:
        _, p1 = torch.max(pred.data, 1)
        _, p2 = torch.max(pred.data, 1)
    assert torch.equal(p1, ref.argmax(1)) and p2 is p1 and calls == [(24, 40)] and launches == []
    assert not any('upsample' in o or 'max' in o for o in log.ops), log.ops
    assert torch.equal(pred.argmax(1), p1) and torch.equal(pred[-1].argmax(dim=-3), p1[1])
    # the deferred maximum VALUES and every other use are the reference's numbers (materialise + ATen)
    v, i = pred.max(1)
    assert launches == [] and torch.equal(i, p1)
    torch.testing.assert_close(v + 0, ref.max(1)[0])
    assert launches == ['upsample']
    torch.testing.assert_close(pred * 1.0, ref)
    torch.testing.assert_close(pred[1].float(), ref[1])
    torch.testing.assert_close(pred.max(), ref.max())                          # (no dim: not the pattern)
    torch.testing.assert_close(pred.max(2)[0], ref.max(2)[0])                  # (not the class axis)
    # not the reference's call: nearest / align_corners=False / a training-mode output take the ATen route at once
    launches.clear()
    out = F.interpolate(_lazy_logits(base, launches), (24, 40), mode='bilinear', align_corners=False)
    assert type(out) is torch.Tensor and launches == ['upsample']
    # a view of an OLDER forward refuses to serve labels
    old = _lazy_logits(base, launches)
    r = F.interpolate(old, (24, 40), mode='bilinear', align_corners=True)
    old._myolo_lazy_state['holder'].generation = 2
    with pytest.raises(L.MyoloError):
        r.max(1)
    # switched off: the round-5 behaviour
    monkeypatch.setattr(R, 'LAZY_RESIZE', False)
    out = F.interpolate(_lazy_logits(base, launches), (24, 40), mode='bilinear', align_corners=True)
    assert type(out) is torch.Tensor


def test_benchmarked_plan_fuses_conv_bn_act_where_every_tile_is_resident(monkeypatch):
    """round 6: the dry-built plan of BASELINE configs[1] (16x3x512x1024, fp16).  Conv + batch statistics + BatchNorm + SiLU
    (+ shortcut) is -- with engine.CONV_BN_ACT on -- ONE launch (myolo_conv_bn_act: device-wide barrier inside) for the 1x1 and stride-2 Conv layers of the 32x64 and 16x32
    maps -- the layers whose tiles are all resident at one 8-wave workgroup per CU; the larger maps and the 3x3 stride-1 layers (conv_midx) keep
    conv + bn_act_fwd.  The BatchNorm backward is one launch
    (reduce + barrier + apply) for the tensors of at most 256 register-resident workgroups."""
    from multiyolov5_amd import engine as E, runtime as R
    from multiyolov5_amd.models.yolo import Model
    assert E.CONV_BN_ACT is False                      # opt-in (measured neutral in the step): MYOLO_CONV_BN_ACT=1
    monkeypatch.setattr(E, 'CONV_BN_ACT', True)
    m = Model(os.path.join(CFG, TAGS['s_psp']))
    m.train()
    plan = R.PlanHolder(m, [torch.zeros(16, 3, 512, 1024)], ('t', 0), torch.float16, True).plan
    fwd = Counter(c.name for op in plan.ops for c in op.fwd_calls)
    bwd = Counter(c.name for op in plan.ops for c in op.bwd_calls)
    fused = [op for op in plan.ops if getattr(op, 'fwd_fused', False)]
    assert fwd['myolo_conv_bn_act'] == len(fused) == 27
    assert all(op.out.h * op.out.w <= 32 * 64 for op in fused) and {(op.out.h, op.out.w) for op in fused} == {(32, 64), (16, 32)}
    assert fwd['myolo_bn_act_fwd'] + fwd['myolo_bn_act_fwd_split'] == 33          # (60 before round 6)
    assert all(op.k == 1 or op.s == 2 for op in fused)
    assert all(not any(c.name.startswith('myolo_bn_act_fwd') for c in op.fwd_calls) for op in fused)
    merged = [op for op in fused if op.weight2 is not None]
    assert len(merged) == 5 and all(op.ffuse.split.contents.c_split == op.c1out for op in merged)      # C3's cv1 | cv2 pairs: two parameter sets
    assert bwd['myolo_bn_act_bwd_fused'] == 7
    # the first layer (no input gradient): BatchNorm backward + weight gradient are ONE pass over (gout, y, x) on the main stream
    stem = [op for op in plan.ops if getattr(op, 'stem_fused', False)]
    assert bwd['myolo_bn_wgrad_stem'] == 1 and len(stem) == 1 and not stem[0].x.requires_grad and (stem[0].out.h, stem[0].out.w, stem[0].cout) == (256, 512, 32)
    assert [c.name for c in stem[0].bwd_calls] == ['myolo_bn_wgrad_stem'] and not stem[0].bwd_calls[0].side
    del plan, m
