"""Data-parallel gradient exchange for the joint det+seg step: one process per GPU, RCCL (`backend='nccl'`) over xGMI.

The reference wraps the model in torch DDP (train.py:243-245): 25 MB buckets discovered through ~230 AccumulateGrad hooks.
Here the whole backward is one launch plan that writes every parameter gradient into ONE flat fp32 buffer
(engine.Plan.flat_grad, 31 MB for yolov5s+PSP), so the exchange is a few large in-place all-reduces of contiguous slices,
issued on a side HIP stream the moment the backward launch list has finished the last kernel writing into a slice
(head and neck first, backbone last) and overlapped with the rest of the backward.  Gradients are averaged over ranks
exactly like DDP (the reference compensates for the detection loss only, train.py:366-367).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): 31 MB is latency-dominated on every algorithm RCCL can pick, so few
large buckets (default 3) beat DDP's 25 MB + remainder split.
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, model, world_size=None, nbuckets=3, group=None):
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.group = group
        self.nbuckets = max(1, int(nbuckets))
        self.stream = None
        self._works = []
        self.issued = []               # (lo, hi) of the slices handed over since the last finish(), in issue order
        self.last_issued = []          # ... of the step before (describe(): what a scaling run's bench line reports)
        self.exposed = None            # bench.py: [(event before, event after)] around the main stream's wait for the exchange
        model.__dict__['_grad_reducer'] = self         # consulted by runtime.PlanFn.backward

    # ---- bucket layout (called once per plan) ---------------------------------------------------------
    def layout(self, sizes, first_op):
        """sizes[i]: numel of parameter i (flat order); first_op[i]: lowest op index of the plan writing its gradient
        (the backward runs ops from high to low index, so the gradient is final once that op has run).
        Returns buckets [(lo, hi, ready_after_op)] over the flat buffer, in completion order."""
        total = sum(sizes)
        target = max(1, (total + self.nbuckets - 1) // self.nbuckets)
        buckets, lo, acc, ready = [], 0, 0, None
        off = 0
        for n, op in zip(sizes, first_op):
            acc += n
            off += n
            ready = op if ready is None else min(ready, op)
            if acc >= target:
                buckets.append((lo, off, ready))
                lo, acc, ready = off, 0, None
        if acc:
            buckets.append((lo, off, ready))
        # a bucket may only be sent once every parameter inside it is final; completion order = descending ready op
        return sorted(buckets, key=lambda b: -b[2])

    # ---- exchange ----------------------------------------------------------------------------------------
    def reduce_slice(self, flat, lo, hi):
        if self.world == 1:
            return
        self.issued.append((int(lo), int(hi)))
        view = flat[lo:hi]
        if flat.is_cuda:
            if self.stream is None:
                self.stream = torch.cuda.Stream(device=flat.device, priority=self._low_priority())
            ev = torch.cuda.Event()
            ev.record()
            self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                # DDP's recipe (pre-divide, then SUM): does not depend on the backend offering ReduceOp.AVG; /world is exact in
                # fp32 for the power-of-two worlds of an 8-GPU node
                view.mul_(1.0 / self.world)
                if dist.get_backend(self.group) == 'nccl':           # RCCL: asynchronous on its own stream
                    self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                else:                                                # gloo on device tensors (single-GPU smoke tests)
                    dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
        else:                                           # gloo (CPU tests)
            view.mul_(1.0 / self.world)
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)

    def finish(self, flat):
        """make the current stream wait for every outstanding slice."""
        self.last_issued, self.issued = self.issued, []
        if self.world == 1 or not flat.is_cuda or self.stream is None:
            self._works = []
            return
        with torch.cuda.stream(self.stream):
            for w in self._works:
                w.wait()
        self._works = []
        if self.exposed is not None:   # how long the main stream sits in this wait = the part of the exchange the backward did not hide
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            torch.cuda.current_stream().wait_stream(self.stream)
            e1.record()
            self.exposed.append((e0, e1))
            return
        torch.cuda.current_stream().wait_stream(self.stream)

    @staticmethod
    def _low_priority():
        """the communication stream only carries the pre-divide and the hand-over to RCCL (ProcessGroupNCCL runs the collective itself on
        its own stream, ordered behind this one).  It ASKS for the lowest priority the device offers (MYOLO_REDUCER_PRIO overrides); what
        it gets is in describe()['comm_stream_priority'] -- ADVICE r5: torch on ROCm may report a one-level priority range and clamp the
        request to the default pool, in which case this stream simply has the launch stream's priority (its two small kernels per slice are
        ~10 us of a 7.7 ms step either way; nothing relies on the priority for correctness).  Ordering against the
        weight-gradient stream: Plan._bwd_eager / the staged backward make the LAUNCH stream wait for the weight-gradient stream before a
        slice is handed over (the slice must be final), and the event recorded on the launch stream right here orders this stream behind
        both -- the exchange never runs beside a weight gradient that still writes into its slice."""
        import os
        if 'MYOLO_REDUCER_PRIO' in os.environ:
            return int(os.environ['MYOLO_REDUCER_PRIO'])
        try:
            lo, hi = torch.cuda.Stream.priority_range()          # (least, greatest): numerically larger = lower priority
            return int(lo)
        except Exception:                                        # noqa: BLE001 -- older torch: default priority
            return 0

    def describe(self):
        """one-line diagnosis data for bench.py's N > 1 line"""
        return {'buckets': self.nbuckets, 'world': self.world, 'pre_divide_then_sum': True,
                'comm_stream_priority': getattr(self.stream, 'priority', None) if self.stream is not None else None,
                'slices_issued_last_step': len(self.last_issued), 'slice_issue_order': list(self.last_issued)}

    def wait(self):
        """kept for call-site symmetry with DDP's implicit sync: PlanFn.backward already waited before handing out grads."""
        return
