"""sched.py (branch-parallel forward, MYOLO_PAR=1, off by default): the dependency analysis is derived from the C-ABI launches of a
plan; these tests check it on CPU dry builds of every head -- (1) against the ops' own input/output views (an independent, module
structure derived ground truth), (2) that the stream/event schedule orders every conflicting pair, incl. under adversarial
interleavings of a simulated multi-stream execution."""
import os
import random

import pytest
import torch

from tests.util import CFG, TAGS


def _plan(tag, training, dtype=torch.float16):
    from multiyolov5_amd import runtime as R
    from multiyolov5_amd.models.yolo import Model
    m = Model(os.path.join(CFG, TAGS[tag]))
    m.train(training)
    if not training:
        m.fuse()
    return R.PlanHolder(m, [torch.zeros(2, 3, 64, 128)], ('t', 0), dtype, training).plan


def _tv_region(tv, space='t'):
    return ('buf', id(tv.buf), space, tv.coff, tv.coff + tv.c)


def _declared(op):
    """(input TVs, output TVs) of an op from its own attributes"""
    n = type(op).__name__
    if n == 'ConvOp':
        return [op.x] + ([op.res] if op.res is not None else []), ([] if op.det else [op.out])
    if n in ('CopyUpOp', 'BilinearOp', 'AvgPoolOp', 'DropoutOp'):
        return [op.src], [op.dst]
    if n == 'AddOp':
        return [op.a, op.b], [op.out]
    if n == 'SppPoolOp':
        return [op.x], list(op.outs)
    if n == 'GateOp':
        return [op.feat, op.att], [op.out]
    if n == 'SegOutOp':
        return [op.low], []
    if n == 'ExportOp':
        return [op.src], []
    if n in ('FocusPackOp', 'ImportOp'):
        return [], [op.out]
    if n == 'DecodeOp':
        return [], []
    raise AssertionError(f'op class {n} unknown to the test: add its views')


@pytest.mark.parametrize('training', [True, False], ids=['train', 'eval'])
@pytest.mark.parametrize('tag', list(TAGS))
def test_call_derived_accesses_cover_the_ops_views(tag, training):
    from multiyolov5_amd import sched as S
    plan = _plan(tag, training)
    acc = S.op_accesses(plan, lambda op: op.fwd_calls)
    for op, (rd, wr) in zip(plan.ops, acc):
        ins, outs = _declared(op)
        for tv in ins:
            r = _tv_region(tv)
            assert any(S._conflict(r, a) and a[3] <= r[3] and a[4] >= r[4] for a in rd), f'{type(op).__name__}: input view not read by its launches'
        for tv in outs:
            r = _tv_region(tv)
            assert any(S._conflict(r, a) and a[3] <= r[3] and a[4] >= r[4] for a in wr), f'{type(op).__name__}: output view not written by its launches'
        # and nothing is written that the op does not own: writes into plan buffers stay inside its declared outputs
        own = [_tv_region(tv) for tv in outs]
        plan_bufs = {id(b) for b in plan.bufs}
        for a in wr:
            if a[0] == 'buf' and a[1] in plan_bufs:
                assert any(o[1] == a[1] and o[3] <= a[3] and a[4] <= o[4] for o in own), f'{type(op).__name__} writes outside its outputs: {a}'


@pytest.mark.parametrize('training', [True, False], ids=['train', 'eval'])
@pytest.mark.parametrize('tag', list(TAGS))
def test_schedule_orders_every_conflict(tag, training):
    from multiyolov5_amd import sched as S
    plan = _plan(tag, training)
    deps, sch, empty = S.forward_schedule(plan, 4)
    assert S.check_schedule(deps, sch, empty) == []
    used = {sch.stream[i] for i in range(len(deps)) if not empty[i]}
    assert 0 in used and len(used) >= 3                     # branches really leave the caller's stream
    on0 = sum(1 for i in range(len(deps)) if not empty[i] and sch.stream[i] == 0)
    assert on0 >= len(deps) // 3                            # ... and the trunk stays on it
    # concat members are independent of each other: two ops writing different channel slices of one buffer never depend
    acc = S.op_accesses(plan, lambda op: op.fwd_calls)
    for i, (rd_i, wr_i) in enumerate(acc):
        for j in deps[i]:
            rd_j, wr_j = acc[j]
            hit = any(S._conflict(a, b) for a in rd_i + wr_i for b in wr_j) or any(S._conflict(a, b) for a in wr_i for b in rd_j)
            assert hit
    # simulated multi-stream execution under adversarial interleavings: every op must observe the producers the serial order gives
    serial_seen = {}
    version = {}
    for i, (rd, wr) in enumerate(acc):
        serial_seen[i] = [max([v for k, v in version.items() if S._conflict(k, r)], default=-1) for r in rd]
        for w in wr:
            version[w] = i
    rng = random.Random(0)
    n = len(deps)
    for trial in range(8):
        done, pos = set(), {k: 0 for k in range(sch.nstreams)}
        queues = {k: [i for i in range(n) if not empty[i] and sch.stream[i] == k] for k in range(sch.nstreams)}
        version = {}
        while len(done) < sum(len(q) for q in queues.values()):
            ready = [k for k in queues if pos[k] < len(queues[k]) and all(j in done for j in sch.waits[queues[k][pos[k]]])]
            assert ready, 'deadlock in the simulated execution'
            k = rng.choice(ready) if trial else max(ready)          # trial 0: always the highest stream first
            i = queues[k][pos[k]]
            rd, wr = acc[i]
            seen = [max([v for kk, v in version.items() if S._conflict(kk, r)], default=-1) for r in rd]
            assert seen == serial_seen[i], f'op {i} ({type(plan.ops[i]).__name__}) read stale data in interleaving {trial}'
            for w in wr:
                version[w] = i
            done.add(i)
            pos[k] += 1


def test_unknown_arguments_serialise():
    """an argument or function the access table does not know counts as a write"""
    import ctypes as C
    from multiyolov5_amd import sched as S

    class FakeCall:
        def __init__(self, name, args):
            self.name, self.args = name, args
    res = S.Resolver([])
    p = C.c_void_p(4096)
    assert S.call_regions(res, FakeCall('myolo_brand_new', (p, 3))) == [(('ptr', 4096), 'w')]
    assert S.call_regions(res, FakeCall('myolo_add', (p, C.c_void_p(8192), 0, C.c_void_p(12288))))[-1] == (('ptr', 12288), 'w')
    # different channel slices of one buffer do not conflict, overlapping ones do, 't' and 'g' twins never
    a, b, c = ('buf', 1, 't', 0, 32), ('buf', 1, 't', 32, 64), ('buf', 1, 't', 16, 48)
    assert not S._conflict(a, b) and S._conflict(a, c) and S._conflict(b, c) and not S._conflict(a, ('buf', 1, 'g', 0, 32))
