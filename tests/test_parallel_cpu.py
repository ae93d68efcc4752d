"""world_size-2 gloo tests of the data-parallel gradient exchange (multiyolov5_amd/parallel.py) -- no GPU needed: the bucket
layout comes from a dry-built training plan, the reduction runs on CPU tensors through the same GradReducer code path."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from multiyolov5_amd import runtime as R
        from multiyolov5_amd.models.yolo import Model
        from multiyolov5_amd.parallel import GradReducer
        from tests.util import CFG, TAGS
        torch.manual_seed(0)
        m = Model(os.path.join(CFG, TAGS['s_psp'])).train()
        red = GradReducer(m, world, nbuckets=3)
        h = R.PlanHolder(m, [torch.zeros(2, 3, 64, 128)], ('t', 0), torch.float32, True)     # dry build (CPU device)
        plan = h.plan
        buckets = plan.grad_buckets(red)
        total = plan.flat_grad.numel()
        # round 6: plans built under a process group of more than one rank carry no launch with a device-wide barrier inside (RCCL's persistent
        # kernels share the CUs: engine.barrier_launches_ok)
        names = {c.name for op in plan.ops for c in list(op.fwd_calls) + list(op.bwd_calls) if hasattr(c, 'name')}
        assert 'myolo_conv_bn_act' not in names and 'myolo_bn_act_bwd_fused' not in names
        # buckets tile the flat buffer exactly, and arrive in backward completion order (descending op index)
        assert sorted((lo, hi) for lo, hi, _ in buckets)[0][0] == 0
        cover = sorted((lo, hi) for lo, hi, _ in buckets)
        assert all(cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1)) and cover[-1][1] == total
        assert [b[2] for b in buckets] == sorted([b[2] for b in buckets], reverse=True)
        # Detect/seg-head parameters (end of the flat buffer) are final first; the Focus conv (offset 0) last
        assert buckets[0][1] == total and buckets[-1][0] == 0
        # the exchange itself: rank r holds r+1 everywhere -> mean 1.5; every slice is reduced exactly once
        g = torch.Generator().manual_seed(100 + rank)
        local = torch.randn(total, generator=g)
        plan.flat_grad.copy_(local)
        for lo, hi, _ in buckets:
            red.reduce_slice(plan.flat_grad, lo, hi)
        red.finish(plan.flat_grad)
        ref = sum(torch.randn(total, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
        err = (plan.flat_grad - ref).abs().max().item()
        q.put((rank, 'ok', err, len(buckets)))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, 'fail', traceback.format_exc(), 0))
    finally:
        dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, status, err, nb in res:
        assert status == 'ok', err
        assert err < 1e-6 and nb == 3


def _sync_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from multiyolov5_amd import engine as E
        from multiyolov5_amd import runtime as R
        from multiyolov5_amd.models.yolo import Model
        from tests.util import CFG, TAGS
        m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(Model(os.path.join(CFG, TAGS['s_psp']))).train()
        plan = R.PlanHolder(m, [torch.zeros(2, 3, 64, 128)], ('t', 0), torch.float16, True).plan            # dry build (CPU device)
        assert plan.has_sync() and not plan.graphable()
        nbn = sum(1 for op in plan.ops if isinstance(op, E.ConvOp) and op.bn is not None)
        fwd = [c for op in plan.ops for c in op.fwd_calls]
        bwd = [c for op in reversed(plan.ops) for c in op.bwd_calls]
        for calls, before, after in ((fwd, ('myolo_conv',), ('myolo_bn_act_fwd_split',)),
                                     (bwd, None, ('myolo_bn_act_bwd_apply_split',))):
            idx = [i for i, c in enumerate(calls) if isinstance(c, E.SyncPoint)]
            assert len(idx) == nbn                                            # one exchange per BatchNorm layer and direction
            for i in idx:
                assert calls[i + 1].name in after and (before is None or calls[i - 1].name in before)
        # every BatchNorm pass of the plan counts the samples of both ranks
        sp = [op.split for op in plan.ops if isinstance(op, E.ConvOp) and op.bn is not None]
        assert all(s.count_scale == world for s in sp)
        assert not any((c.name == 'myolo_bn_act_fwd' and c.args[1] is not None) or (c.name == 'myolo_bn_act_bwd_apply' and c.args[2] is not None)
                       for c in fwd + bwd if isinstance(c, E.Call))               # (the plain entry points only serve activation-only layers)
        # the exchange itself (gloo, CPU tensors): the statistics arrays hold the sums over the ranks afterwards
        op = next(o for o in plan.ops if isinstance(o, E.ConvOp) and o.bn is not None)
        op.stats.fill_(float(rank + 1))
        next(c for c in op.fwd_calls if isinstance(c, E.SyncPoint))()
        assert float(op.stats.min()) == float(op.stats.max()) == 3.0
        # pruned one-loss backward lists keep the exchange of every layer that still runs, and the items serialise (marks aside)
        h_det = frozenset({0, 1, 2})
        items = plan._bwd_items(None, h_det)
        kept = sum(1 for it in items if isinstance(it, E.SyncPoint))
        assert 0 < kept < nbn
        q.put((rank, 'ok', ''))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, 'fail', traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_sync_batchnorm_plan_world2_gloo():
    """nn.SyncBatchNorm (train.py --sync-bn) under a 2-rank group: the launch lists carry one all-reduce of the per-channel sums per
    BatchNorm layer and direction, between the launch that produces them and the pass that consumes them"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, status, err in res:
        assert status == 'ok', err


def test_layout_single_process():
    from multiyolov5_amd.parallel import GradReducer

    class M:
        pass
    red = GradReducer(M(), world_size=4, nbuckets=2)
    b = red.layout([10, 10, 10, 10], [0, 3, 5, 9])
    assert b == [(20, 40, 5), (0, 20, 0)]
    assert GradReducer(M(), 1).layout([5], [2]) == [(0, 5, 2)]


def test_reducer_is_not_pickled_or_deepcopied_with_the_model(tmp_path):
    """train.py:481-499 deep-copies and torch.saves the model (and ModelEMA deep-copies it) with the gradient exchange attached:
    the GradReducer (process group, HIP stream, Work handles) and every other runtime key must stay out of the copy / the file"""
    import copy
    import io
    from multiyolov5_amd import runtime as R
    from multiyolov5_amd.models.common import Conv
    from multiyolov5_amd.parallel import GradReducer

    class Unpicklable:                       # stands in for the ProcessGroup / Stream members
        def __reduce__(self):
            raise TypeError('process groups do not pickle')
    m = Conv(8, 16, 3)
    red = GradReducer(m, world_size=2)
    red.group = Unpicklable()
    m.__dict__['_plans'] = {'k': Unpicklable()}
    m.__dict__['_tensor_list'] = [Unpicklable()]
    assert '_grad_reducer' in m.__dict__
    st = m.__getstate__()
    assert not any(k in st for k in R.PlannedModule._RUNTIME_STATE)
    c = copy.deepcopy(m)
    assert not any(k in c.__dict__ for k in R.PlannedModule._RUNTIME_STATE)
    buf = io.BytesIO()
    torch.save({'model': m}, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=False)['model']
    assert not any(k in back.__dict__ for k in R.PlannedModule._RUNTIME_STATE)
    assert torch.equal(back.conv.weight, m.conv.weight)
    assert m.__dict__['_grad_reducer'] is red         # the live model keeps its reducer
