"""Branch-parallel execution of a launch plan's FORWARD list (experimental, off by default: MYOLO_PAR=1).

`engine.Plan` runs its ops as one chain on the caller's stream.  Many of them are independent (C3's `cv1`/`cv2`, the three Detect
levels, the PyramidPooling branches, the heads) and, on small maps or at batch 1, too small to fill 256 CUs: the detect.py forward
is ~100 launches of ~20 us.  This module derives the dependencies of the launch list FROM THE LAUNCHES THEMSELVES -- every pointer
a C-ABI call receives is resolved to a memory region and classified read / write by the function's signature (ACCESS) -- and deals
the ops to a few HIP streams with event edges where a dependency crosses streams.  Nothing is inferred from module structure, so a
new op or buffer trick cannot silently escape the analysis: an argument the table does not know is treated as a WRITE of that
pointer (serialising).

The schedule is checked on the CPU (tests/test_sched_cpu.py): for every pair of ops with a RAW / WAR / WAW conflict the later one
must be ordered after the earlier by stream order or an event chain, also under adversarial interleavings of a simulated execution,
and the call-derived accesses are cross-checked against the ops' own input/output views.  State at the end of round 1 (the GPU
budget allowed one 8-second look): the eager parallel forward reproduces the oracle on the full-resolution parity test
(`MYOLO_PAR=1 pytest tests/test_gpu_ops.py -k full_resolution`); capturing the multi-stream list into a hipGraph dumped core, so
MYOLO_PAR=1 runs eval forwards eagerly; nothing is measured yet.  `Plan.run_fwd` only consults this module when MYOLO_PAR=1.
"""
import ctypes as C

from . import _lib as L

R, W = 'r', 'w'

# positional access modes of every C-ABI function a launch list may contain: index -> mode for pointer-like arguments
# (myolo_tensor byref or raw pointer).  Accumulating outputs are writes (a write orders against earlier readers AND writers).
ACCESS = {
    'myolo_focus_pack': {0: R, 6: W},
    'myolo_seg_upsample_fwd': {0: R, 1: W},
    'myolo_seg_upsample_bwd': {0: R, 8: W, 10: R},
    'myolo_bn_act_fwd': {0: R, 1: R, 2: R, 3: R, 4: W, 5: W, 6: W, 7: W, 11: R, 12: W},
    'myolo_bn_act_bwd_reduce': {0: R, 1: R, 2: R, 3: R, 4: R, 6: W},
    'myolo_bn_act_bwd_apply': {0: R, 1: R, 2: R, 3: R, 4: R, 6: R, 7: W, 8: W, 9: W, 10: W},
    'myolo_detect_unpermute': {0: R, 4: W},
    'myolo_detect_decode': {0: R, 9: W},
    'myolo_fill_zero': {0: W},
    'myolo_copy_up_fwd': {0: R, 1: W},
    'myolo_copy_up_bwd': {0: R, 1: W},
    'myolo_bilinear_fwd': {0: R, 1: W},
    'myolo_bilinear_bwd': {0: R, 1: W, 3: W},
    'myolo_adaptive_avgpool_fwd': {0: R, 1: W, 2: W},
    'myolo_adaptive_avgpool_bwd': {0: R, 1: W},
    'myolo_dropout_fwd': {0: R, 1: W, 2: W, 4: W},
    'myolo_dropout_bwd': {0: R, 1: R, 2: W},
    'myolo_add': {0: R, 1: W},
    'myolo_spp_pool_fwd': {0: R, 1: W, 2: W, 3: W, 4: W},
    'myolo_spp_pool_bwd': {0: R, 1: R, 2: R, 3: R, 4: W},
    'myolo_gate_fwd': {0: R, 1: R, 2: W},
    'myolo_gate_bwd': {0: R, 1: R, 2: R, 3: W, 5: W},
    'myolo_cast_from_f32': {0: R, 1: W},
}
CONV_FIELDS = {'x': R, 'y': W, 'w': R, 'scale': R, 'shift': R, 'res': R, 'stats': W}
WGRAD_FIELDS = {'x': R, 'dy': R, 'dw': W, 'db': W, 'ws': W}


class Resolver:
    """pointer -> region.  A myolo_tensor view into a registered NHWC buffer is (buffer id, 't'|'g', c0, c1): channel slices of one
    buffer interleave in memory, so address intervals would over-serialise concat members.  Anything else is keyed by its exact
    start address (arena carves, flat-gradient slices, packed weights, per-op scratch are distinct allocations or slices)."""

    def __init__(self, bufs):
        self.spans = []
        for b in bufs:
            for space, t in (('t', b.t), ('g', b.g)):
                if t is not None:
                    self.spans.append((t.data_ptr(), t.data_ptr() + t.numel() * t.element_size(), id(b), space, b.c, t.element_size()))
        self.spans.sort()

    def _find(self, ptr):
        for lo, hi, bid, space, c, es in self.spans:            # a few hundred spans, build-time only
            if lo <= ptr < hi:
                return lo, bid, space, c, es
        return None

    def view(self, ct):
        if not ct.ptr:
            return None
        hit = self._find(ct.ptr)
        if hit is None:
            return ('ptr', int(ct.ptr))
        lo, bid, space, c, es = hit
        coff = ((ct.ptr - lo) // es) % c
        return ('buf', bid, space, coff, coff + int(ct.c))

    def raw(self, ptr):
        if not ptr:
            return None
        hit = self._find(ptr)
        if hit is None:
            return ('ptr', int(ptr))
        lo, bid, space, c, es = hit
        return ('buf', bid, space, 0, c)


def _conflict(a, b):
    if a[0] != b[0]:
        return False
    if a[0] == 'ptr' or a[0] == 'cell':
        return a[1] == b[1]
    return a[1] == b[1] and a[2] == b[2] and a[3] < b[4] and b[3] < a[4]


def _arg_regions(res, arg, mode, out):
    if arg is None:
        return
    obj = getattr(arg, '_obj', None)                    # ctypes byref(...)
    if obj is not None:
        if isinstance(obj, L.Tensor):
            r = res.view(obj)
            if r is not None:
                out.append((r, mode))
            return
        raise TypeError(f'unexpected byref payload {type(obj).__name__}')
    if isinstance(arg, C.c_void_p):
        if arg.value:
            out.append((res.raw(arg.value), mode))
        else:
            out.append((('cell', id(arg)), mode))       # a plan input slot: bound per run (read-only image / tensor input)
        return
    if isinstance(arg, int) and not isinstance(arg, bool):
        return                                          # scalars (pointers never travel as python ints in engine.py)
    raise TypeError(f'unclassified pointer-like argument {type(arg).__name__}')


def call_regions(res, call):
    """[(region, 'r'|'w')] of one launch."""
    out = []
    name, args = call.name, call.args
    if name in ('myolo_conv', 'myolo_conv_wgrad'):
        d = args[0]._obj
        for f, mode in (CONV_FIELDS if name == 'myolo_conv' else WGRAD_FIELDS).items():
            v = getattr(d, f)
            if isinstance(v, L.Tensor):
                r = res.view(v)
            else:
                r = res.raw(v) if v else None
            if r is not None:
                out.append((r, mode))
        return out
    table = ACCESS.get(name)
    for i, a in enumerate(args):
        ptr_like = a is not None and (hasattr(a, '_obj') or isinstance(a, C.c_void_p))
        if not ptr_like:
            continue
        mode = W if table is None else table.get(i, W)  # unknown function / position: assume it writes
        _arg_regions(res, a, mode, out)
    return out


def op_accesses(plan, calls_of):
    """per op: (reads, writes) region lists of the given launch list (`calls_of(op)`)."""
    bufs = list(plan.bufs)
    for op in plan.ops:
        for v in vars(op).values():
            if v.__class__.__name__ == 'Buf':
                bufs.append(v)
    res = Resolver(bufs)
    acc = []
    for op in plan.ops:
        rd, wr = [], []
        for c in calls_of(op):
            for region, mode in call_regions(res, c):
                (wr if mode == W else rd).append(region)
        acc.append((rd, wr))
    return acc


def dependencies(acc):
    """deps[i] = set of earlier ops op i must run after (RAW on its reads, WAR + WAW on its writes)."""
    deps = []
    for i, (rd, wr) in enumerate(acc):
        d = set()
        for j in range(i - 1, -1, -1):
            prd, pwr = acc[j]
            hit = any(_conflict(a, b) for a in rd for b in pwr) or any(_conflict(a, b) for a in wr for b in pwr) or \
                any(_conflict(a, b) for a in wr for b in prd)
            if hit:
                d.add(j)
        deps.append(d)
    return deps


def reduce_transitively(deps):
    """drop dependencies already implied by another dependency (keeps the event count small)."""
    n = len(deps)
    reach = [set() for _ in range(n)]
    out = []
    for i in range(n):
        keep = set()
        for j in sorted(deps[i], reverse=True):
            if not any(j in reach[k] or j == k for k in keep):
                keep.add(j)
        out.append(keep)
        r = set(deps[i])
        for j in deps[i]:
            r |= reach[j]
        reach[i] = r
    return out


class Schedule:
    """stream[i]: stream index of op i (0 = the caller's stream); waits[i]: ops on OTHER streams whose completion event op i's stream
    must wait for first; events: ops that record an event when done; joins: streams stream 0 waits for at the end."""

    def __init__(self, stream, waits, events, nstreams):
        self.stream, self.waits, self.events, self.nstreams = stream, waits, events, nstreams
        self.joins = sorted(set(stream) - {0})


def schedule(deps, empty, nstreams=4):
    """list scheduling in emission order (which is a valid topological order).  An op follows a dependency that is still the tail of
    its stream (no wait needed: stream order); otherwise its inputs were produced a while ago and the chain that consumed them has
    moved on -- the op starts a branch on the least recently used side stream.  `empty[i]`: op i launches
    nothing (stays on stream 0, no edges)."""
    n = len(deps)
    stream = [0] * n
    tail = [-1] * nstreams                      # last op placed on each stream
    last_use = [-1] * nstreams
    red = reduce_transitively(deps)
    waits, events = [set() for _ in range(n)], set()
    for i in range(n):
        if empty[i]:
            continue
        d = red[i]
        s = 0
        if d:
            # continue the stream of a dependency that is still that stream's tail (stream order replaces an event); with several such
            # (a join) the lowest stream wins, which keeps the trunk of the network on the caller's stream
            live = sorted(stream[j] for j in d if tail[stream[j]] == j)
            if live:
                s = live[0]
            else:
                s = min(range(1, nstreams), key=lambda k: last_use[k]) if nstreams > 1 else 0
        stream[i] = s
        for j in d:
            if stream[j] != s:
                waits[i].add(j)
                events.add(j)
        tail[s] = i
        last_use[s] = i
    return Schedule(stream, waits, events, nstreams)


def check_schedule(deps, sched, empty):
    """every dependency must be enforced by stream order or an event chain; returns the list of violated (i, j) pairs."""
    n = len(deps)
    before = [set() for _ in range(n)]          # ops guaranteed complete before op i starts
    last = {}
    for i in range(n):
        if empty[i]:
            continue
        b = set()
        p = last.get(sched.stream[i])
        if p is not None:
            b |= before[p] | {p}
        for j in sched.waits[i]:
            b |= before[j] | {j}
        before[i] = b
        last[sched.stream[i]] = i
    return [(i, j) for i in range(n) if not empty[i] for j in deps[i] if not empty[j] and j not in before[i]]


def forward_schedule(plan, nstreams=4):
    acc = op_accesses(plan, lambda op: op.fwd_calls)
    deps = dependencies(acc)
    empty = [not op.fwd_calls for op in plan.ops]
    return deps, schedule(deps, empty, nstreams), empty
