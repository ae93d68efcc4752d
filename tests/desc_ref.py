"""Test infrastructure for the -m gpu suite: torch fp32 (CPU) evaluation of libmyolo launch DESCRIPTORS (include/myolo.h) and a launch
interposer that checks every convolution-family launch of a plan against it.

A plan's launch list is a sequence of C-ABI calls whose arguments are plain descriptors (device pointers, dims, strides, tap tables).
The dispatchers behind `myolo_conv` / `myolo_conv_dgrad_s2` / `myolo_conv_dgrad_bn` / `myolo_conv_wgrad` choose a kernel family and a
template variant from the descriptor (tile counts against the resident workgroups, channel widths, tap counts ...), so a variant tested
at batch 2 is not necessarily the one that runs at batch 16.  `LaunchChecker` closes that hole by construction: it wraps
`engine.Call.__call__`, and for every such launch it reads the operands back right before the launch, evaluates the formula of myolo.h
(:87-92 for the convolution, :131-149 for the BatchNorm-apply dgrad, the weight-gradient contraction) in fp32 on the CPU over those
very operands, runs the launch and compares what it stored (outputs, accumulations, BatchNorm statistics, BatchNorm-backward sums).
Whatever kernel the library picked for the benchmarked shapes is the one that gets checked.  Nothing here is product code."""
import ctypes as C

import torch
import torch.nn.functional as F

from multiyolov5_amd import _lib as L

_hip = None


def hip():
    """the HIP runtime torch already loaded (hipMemcpy for raw device pointers: the descriptors carry addresses, not tensors)"""
    global _hip
    if _hip is None:
        path = None
        with open('/proc/self/maps') as f:
            for line in f:
                if 'libamdhip64' in line:
                    path = line.split()[-1]
                    break
        _hip = C.CDLL(path or 'libamdhip64.so')
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemcpy.restype = C.c_int
    return _hip


_pinned = [None]


def dev_read(ptr, nbytes):
    """device bytes -> host uint8 tensor (through one reusable pinned staging buffer: pageable hipMemcpy was most of the checker's time)"""
    nbytes = int(nbytes)
    out = torch.empty(nbytes, dtype=torch.uint8)
    if nbytes:
        try:
            if _pinned[0] is None or _pinned[0].numel() < nbytes:
                _pinned[0] = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8).pin_memory()
            stage = _pinned[0]
        except RuntimeError:                                 # (no pinned memory available: straight into the pageable tensor)
            stage = out
        e = hip().hipMemcpy(stage.data_ptr(), C.c_void_p(int(ptr)), nbytes, 2)     # hipMemcpyDeviceToHost
        assert e == 0, f'hipMemcpy D2H failed: {e}'
        if stage is not out:
            out.copy_(stage[:nbytes])
    return out


def _ptr(p):
    if p is None:
        return 0
    if isinstance(p, int):
        return p
    if isinstance(p, C.c_void_p):
        return p.value or 0
    return C.cast(p, C.c_void_p).value or 0


def read_f32(ptr, n):
    return dev_read(_ptr(ptr), 4 * n).view(torch.float32).clone()


def read_u32(ptr, n):
    return dev_read(_ptr(ptr), 4 * n).view(torch.int32).clone()


def read_tensor(t, channels=None):
    """myolo_tensor (NHWC view with strides in elements) -> fp32 [n,h,w,c] on the CPU"""
    es = 2 if t.dtype == L.F16 else 4
    dt = torch.float16 if t.dtype == L.F16 else torch.float32
    c = channels if channels is not None else t.c
    if t.n * t.h * t.w * c == 0:
        return torch.zeros(t.n, t.h, t.w, c)
    span = (t.n - 1) * t.sn + (t.h - 1) * t.sh + (t.w - 1) * t.sw + c
    raw = dev_read(t.ptr, span * es).view(dt)
    return torch.as_strided(raw, (t.n, t.h, t.w, c), (t.sn, t.sh, t.sw, 1)).float()


def read_det(t, det_no):
    """Detect's forward output [n, na, h, w, no] (dense) -> [n,h,w,na*no]"""
    es = 2 if t.dtype == L.F16 else 4
    dt = torch.float16 if t.dtype == L.F16 else torch.float32
    na = t.c // det_no
    raw = dev_read(t.ptr, t.n * na * t.h * t.w * det_no * es).view(dt).float()
    return raw.view(t.n, na, t.h, t.w, det_no).permute(0, 2, 3, 1, 4).reshape(t.n, t.h, t.w, na * det_no)


def read_out(d):
    return read_det(d.y, d.det_no) if d.det_no > 0 else read_tensor(d.y)


def act_fn(v, act):
    if act == L.ACT_SILU:
        return F.silu(v)
    if act == L.ACT_SIGMOID:
        return torch.sigmoid(v)
    return v


def act_grad(z, act):
    if act == L.ACT_SILU:
        s = torch.sigmoid(z)
        return s * (1 + z * (1 - s))
    if act == L.ACT_SIGMOID:
        s = torch.sigmoid(z)
        return s * (1 - s)
    return torch.ones_like(z)


def _shifted(x, taps_dy, taps_dx, Ho, Wo, s):
    """generator of the tap views x[n, oy*s+dy, ox*s+dx, :] (zero outside the image) as [n,Ho,Wo,c]"""
    H, W = x.shape[1], x.shape[2]
    ply, plx = max(0, -min(taps_dy)), max(0, -min(taps_dx))
    phy = max(0, (Ho - 1) * s + max(taps_dy) - (H - 1))
    phx = max(0, (Wo - 1) * s + max(taps_dx) - (W - 1))
    xp = F.pad(x, (0, 0, plx, phx, ply, phy))
    for dy, dx in zip(taps_dy, taps_dx):
        ys, xs = ply + dy, plx + dx
        yield xp[:, ys:ys + (Ho - 1) * s + 1:s, xs:xs + (Wo - 1) * s + 1:s, :]


def conv_acc(d, x=None):
    """raw fp32 accumulators of a myolo_conv_desc: sum_t x[n,(oy*s+dy_t)>>up,(ox*s+dx_t)>>up,:] . W[:,tap_w[t],:]  (myolo.h:87)"""
    x = read_tensor(d.x) if x is None else x
    if d.up_shift:
        r = 1 << d.up_shift
        x = x.repeat_interleave(r, 1).repeat_interleave(r, 2)
    es = 2 if d.x.dtype == L.F16 else 4
    dt = torch.float16 if d.x.dtype == L.F16 else torch.float32
    w = dev_read(d.w, d.cout_pad * d.wtaps * d.cin_pad * es).view(dt).float().view(d.cout_pad, d.wtaps, d.cin_pad)
    cout, cin = d.y.c, d.x.c
    nt = d.ntaps
    dys, dxs, tw = list(d.tap_dy[:nt]), list(d.tap_dx[:nt]), list(d.tap_w[:nt])
    N, Ho, Wo = d.y.n, d.y.h, d.y.w
    acc = torch.zeros(N * Ho * Wo, cout)
    for t, sl in enumerate(_shifted(x, dys, dxs, Ho, Wo, d.stride)):
        acc.addmm_(sl.reshape(-1, cin), w[:cout, tw[t], :cin].t())
    return acc.view(N, Ho, Wo, cout)


def conv_epilogue(d, acc):
    v = acc
    if _ptr(d.scale):
        v = v * read_f32(d.scale, d.y.c)
    if _ptr(d.shift):
        v = v + read_f32(d.shift, d.y.c)
    v = act_fn(v, d.act)
    if d.res.ptr:
        v = v + read_tensor(d.res, channels=d.y.c)
    return v


def stat_sums(ptr, c):
    """[MYOLO_STAT_COPIES][2][c] fp32 -> (sum over copies) [2][c] in fp64"""
    return read_f32(ptr, L.STAT_COPIES * 2 * c).double().view(L.STAT_COPIES, 2, c).sum(0)


def bnb_ref(seg, gout):
    """BatchNorm-backward sums of one myolo_bn_bwd_seg (myolo.h:93-99) from the gout the launch stored"""
    Cn = seg.c1 - seg.c0
    y = read_tensor(seg.y)
    sv = read_f32(seg.saved, 2 * Cn)
    mean, istd = sv[:Cn], sv[Cn:]
    gamma, beta = read_f32(seg.gamma, Cn), read_f32(seg.beta, Cn)
    xhat = (y - mean) * istd
    dz = gout[..., seg.c0:seg.c1] * act_grad(xhat * gamma + beta, seg.act)
    return torch.stack([dz.double().sum((0, 1, 2)), (dz * xhat).double().sum((0, 1, 2))])


def apply_fold_ref(d, f):
    """dy of myolo_conv_dgrad_bn (myolo.h:131-149; bn_act.hip's folded apply pass): dy = sc*dz + cb*y + cd"""
    gout = read_tensor(d.x)
    y = read_tensor(f.y)
    K = f.y.c
    sv = read_f32(f.saved, 2 * K)
    mean, istd = sv[:K], sv[K:]
    gamma, beta = read_f32(f.gamma, K), read_f32(f.beta, K)
    ds = stat_sums(f.dsum, K)
    rM = 1.0 / (f.y.n * f.y.h * f.y.w)
    sc = gamma * istd
    d0, d1 = ds[0].float(), ds[1].float()
    cb = -sc * (d1 * rM) * istd
    cd = -sc * (d0 * rM) - cb * mean
    dz = gout * act_grad(y * sc + (beta - mean * sc), f.act)
    return sc * dz + cb * y + cd, d0, d1


def wgrad_ref(wd, dy=None):
    """dW[co,ci,t] = sum_pixels dy[n,oy,ox,co] * x[n,oy*s+dy_t,ox*s+dx_t,ci]; db[co] = sum dy  (dy: given instead of read from wd.dy)"""
    x = read_tensor(wd.x)
    dy = read_tensor(wd.dy) if dy is None else dy
    if wd.up_shift:
        r = 1 << wd.up_shift
        x = x.repeat_interleave(r, 1).repeat_interleave(r, 2)
    cout = wd.cout if wd.cout > 0 else dy.shape[-1]
    cin = wd.cin if wd.cin > 0 else wd.x.c
    nt = wd.ntaps
    N, Ho, Wo = dy.shape[0], dy.shape[1], dy.shape[2]
    dyf = dy[..., :cout].reshape(-1, cout)
    out = torch.zeros(cout, cin, nt)
    for t, sl in enumerate(_shifted(x[..., :cin], list(wd.tap_dy[:nt]), list(wd.tap_dx[:nt]), Ho, Wo, wd.stride)):
        out[:, :, t] = dyf.t() @ sl.reshape(-1, cin)
    return out, dyf.double().sum(0).float()


def _fv(a):
    return a.value if hasattr(a, 'value') else a


def _split_vec(p1, p2, C, cs):
    """per-channel parameter vector of a (possibly split, myolo_bn_split) BatchNorm launch: set 1 serves channels [0, cs), set 2 the rest"""
    if not _ptr(p1):
        return None
    v = read_f32(p1, cs)
    if cs < C:
        v = torch.cat([v, read_f32(p2, C - cs)])
    return v


class _BnArgs:
    """the common head of the three BatchNorm entry points (include/myolo.h:192-240) out of a Call's argument tuple"""

    def __init__(self, y, split):
        self.C = y.c
        self.M = y.n * y.h * y.w
        self.sp = split
        self.cs = split.c_split if split is not None else y.c
        self.scale = max(1, split.count_scale) if split is not None else 1


def bn_fwd_ref(y, stats, gamma, beta, eps, act, res):
    """out = act((y - mean) * invstd * gamma + beta) + res with batch statistics from the conv epilogue's sums (common.py:42-43, train mode)"""
    C = y.shape[-1]
    M = y.shape[0] * y.shape[1] * y.shape[2]
    if stats is None:
        out = act_fn(y, act)
        return out + (res if res is not None else 0), None, None
    mean = (stats[0] / M)
    var = (stats[1] / M - mean * mean).clamp_min(0)
    invstd = 1.0 / torch.sqrt(var + eps)
    z = (y.double() - mean) * invstd * gamma.double() + beta.double()
    out = act_fn(z.float(), act)
    return out + (res if res is not None else 0), mean, var


BN_NAMES = ('myolo_bn_act_fwd', 'myolo_bn_act_fwd_split', 'myolo_bn_act_bwd_reduce', 'myolo_bn_act_bwd_reduce_split', 'myolo_bn_act_bwd_apply',
            'myolo_bn_act_bwd_apply_split', 'myolo_bn_act_bwd_fused')
CONV_NAMES = ('myolo_conv', 'myolo_conv_dgrad_s2', 'myolo_conv_dgrad_bn', 'myolo_conv_wgrad', 'myolo_conv_pair', 'myolo_conv_bn_act', 'myolo_bn_wgrad_stem')


class LaunchChecker:
    """with LaunchChecker(check_fn) as lc: run a forward (+ backward) with engine.NATIVE_EXEC off.  lc.n[name] counts the checked launches,
    lc.bad collects failures (check_fn = tests.gpu_util.check with collect=)."""

    def __init__(self, check, tag, tol_out=3e-3, tol_stat=2e-3, tol_w=2e-3, every=1, bn=False, only=None):
        self.check, self.tag, self.tol_out, self.tol_stat, self.tol_w = check, tag, tol_out, tol_stat, tol_w
        self.only = only                       # names: evaluate only these launches (everything else just runs)
        self.bn = bn                           # also the BatchNorm forward / backward-reduce / backward-apply launches: every bn-th of them (True = 1)
        self.kbn = 0
        self.bad, self.n, self.k, self.every = [], {}, 0, every

    def __enter__(self):
        from multiyolov5_amd import engine as E
        self.E = E
        self.orig = E.Call.__call__
        me = self

        def wrapped(call, st):
            if me.only is not None and call.name not in me.only:
                return me.orig(call, st)
            if call.name not in CONV_NAMES and not (me.bn and call.name in BN_NAMES):
                return me.orig(call, st)
            if call.name in BN_NAMES:
                me.kbn += 1
                if me.kbn % int(me.bn):
                    return me.orig(call, st)
                me.k += 1
                return me.run(call, st)
            me.k += 1
            if me.k % me.every:
                return me.orig(call, st)
            return me.run(call, st)
        E.Call.__call__ = wrapped
        self.modes = (E.NATIVE_EXEC, E.GRAPH_TRAIN)
        E.NATIVE_EXEC, E.GRAPH_TRAIN = False, False
        return self

    def __exit__(self, *exc):
        self.E.Call.__call__ = self.orig
        self.E.NATIVE_EXEC, self.E.GRAPH_TRAIN = self.modes
        return False

    def _ck(self, what, got, ref, tol):
        self.check(f'{self.tag}/{what}', got, ref, tol, collect=self.bad)

    def _desc(self, a):
        return a._obj if hasattr(a, '_obj') else a.contents

    def _conv_pre(self, d):
        pre = {'acc': conv_acc(d)}
        if d.accumulate:
            pre['y0'] = read_out(d)
        if _ptr(d.stats):
            pre['s0'] = stat_sums(d.stats, d.y.c)
        if d.nbnb:
            pre['b0'] = [stat_sums(d.bnb[i].dsum, d.bnb[i].c1 - d.bnb[i].c0) for i in range(d.nbnb)]
        return pre

    def _conv_post(self, d, pre, what):
        ref = conv_epilogue(d, pre['acc'])
        if d.accumulate:
            ref = ref + pre['y0']
        got = read_out(d)
        self._ck(what + '/y', got, ref, self.tol_out)
        if _ptr(d.stats):
            a = pre['acc'].double()
            ref_s = torch.stack([a.sum((0, 1, 2)), (a * a).sum((0, 1, 2))])
            self._ck(what + '/stats', (stat_sums(d.stats, d.y.c) - pre['s0']).float(), ref_s.float(), self.tol_stat)
        for i in range(d.nbnb):
            sg = d.bnb[i]
            self._ck(what + f'/bnb{i}', (stat_sums(sg.dsum, sg.c1 - sg.c0) - pre['b0'][i]).float(), bnb_ref(sg, got).float(), self.tol_stat)

    def _what(self, name, d):
        return f'{name}[{d.x.n}x{d.x.h}x{d.x.w}x{d.x.c}->{d.y.h}x{d.y.w}x{d.y.c} t{d.ntaps}s{d.stride}' + \
            ('+acc' if d.accumulate else '') + ('+res' if d.res.ptr else '') + ('+stats' if _ptr(d.stats) else '') + \
            (f'+bnb{d.nbnb}' if d.nbnb else '') + ('+epi' if (_ptr(d.scale) or _ptr(d.shift) or d.act) else '') + \
            (f'+det{d.det_no}' if d.det_no else '') + f']#{self.k}'

    def run(self, call, st):
        torch.cuda.synchronize()
        name = call.name
        self.n[name] = self.n.get(name, 0) + 1
        if name in BN_NAMES:
            return self._run_bn(call, st, name)
        if name == 'myolo_conv':
            d = self._desc(call.args[0])
            pre = self._conv_pre(d)
            self.orig(call, st)
            torch.cuda.synchronize()
            self._conv_post(d, pre, self._what('conv', d))
        elif name == 'myolo_conv_bn_act':
            # round 6: conv (raw output + statistics) AND the BatchNorm forward pass in one launch.  The conv half against the descriptor's
            # formula as for myolo_conv; the BatchNorm half against first principles over the y and the sums the launch stored
            d, f = self._desc(call.args[0]), self._desc(call.args[1])
            split = f.split.contents if f.split else None
            pre = self._conv_pre(d)
            B = _BnArgs(d.y, split)
            C, M = B.C, B.M * B.scale
            rm0 = _split_vec(f.running_mean, split.running_mean2 if split else None, C, B.cs)
            rv0 = _split_vec(f.running_var, split.running_var2 if split else None, C, B.cs)
            res = read_tensor(f.res, channels=C) if f.res.ptr else None
            self.orig(call, st)
            torch.cuda.synchronize()
            what = self._what('conv_bn_act', d) + ('+res' if res is not None else '') + ('+split' if split is not None and B.cs < C else '')
            assert int(read_u32(f.barrier, 19 * 32)[18 * 32]) == 0, what + ': the grid barrier timed out'
            self._conv_post(d, pre, what)
            y = read_out(d)
            stats = stat_sums(d.stats, C) - pre['s0']
            gamma = _split_vec(f.gamma, split.gamma2 if split else None, C, B.cs)
            beta = _split_vec(f.beta, split.beta2 if split else None, C, B.cs)
            mean = stats[0] / M
            var = (stats[1] / M - mean * mean).clamp_min(0)
            invstd = 1.0 / torch.sqrt(var + float(f.eps))
            out_ref = act_fn(((y.double() - mean) * invstd * gamma.double() + beta.double()).float(), int(f.act))
            if res is not None:
                out_ref = out_ref + res
            self._ck(what + '/out', read_tensor(f.out), out_ref, self.tol_out)
            sv = read_f32(f.saved, 2 * C)
            self._ck(what + '/saved_mean', sv[:C], mean.float(), 1e-4)
            self._ck(what + '/saved_invstd', sv[C:], invstd.float(), 1e-4)
            if rm0 is not None:
                mom = float(f.momentum)
                rm1 = _split_vec(f.running_mean, split.running_mean2 if split else None, C, B.cs)
                rv1 = _split_vec(f.running_var, split.running_var2 if split else None, C, B.cs)
                self._ck(what + '/running_mean', rm1, ((1 - mom) * rm0.double() + mom * mean).float(), 1e-4)
                self._ck(what + '/running_var', rv1, ((1 - mom) * rv0.double() + mom * var * M / max(M - 1, 1)).float(), 1e-4)
        elif name == 'myolo_bn_wgrad_stem':
            # round 6: BatchNorm backward + weight gradient of the first layer in one pass.  dy never exists on the device: the reference forms it in
            # fp64 from (gout, y, saved, gamma, beta) -- autograd of BatchNorm (train) + SiLU -- and contracts it with x
            a = call.args
            wd, god, yd = self._desc(a[0]), self._desc(a[1]), self._desc(a[2])
            C = god.c
            M = god.n * god.h * god.w
            gout, y = read_tensor(god).double(), read_tensor(yd).double()
            sv = read_f32(a[3], 2 * C).double()
            mean, invstd = sv[:C], sv[C:]
            gamma, beta = read_f32(a[4], C).double(), read_f32(a[5], C).double()
            xhat = (y - mean) * invstd
            z = xhat * gamma + beta
            sg = torch.sigmoid(z)
            g = gout * (sg * (1 + z * (1 - sg)))
            d0, d1 = g.sum((0, 1, 2)), (g * xhat).sum((0, 1, 2))
            dy_ref = (gamma * invstd) * (g - d0 / M - xhat * d1 / M)
            ref_w, _ = wgrad_ref(wd, dy=dy_ref.float())
            cin = wd.cin if wd.cin > 0 else wd.x.c
            n = C * cin * wd.ntaps
            w0, g0, b0 = read_f32(wd.dw, n), read_f32(a[7], C), read_f32(a[8], C)
            self.orig(call, st)
            torch.cuda.synchronize()
            what = f'bn_wgrad_stem[{wd.x.n}x{wd.x.h}x{wd.x.w}x{cin}->{C} t{wd.ntaps}]#{self.k}'
            self._ck(what + '/dw', read_f32(wd.dw, n) - w0, ref_w.reshape(-1), self.tol_w)
            self._ck(what + '/dgamma', read_f32(a[7], C) - g0, d1.float(), self.tol_stat)
            self._ck(what + '/dbeta', read_f32(a[8], C) - b0, d0.float(), self.tol_stat)
        elif name == 'myolo_conv_pair':         # myolo.h: b(a(x)), the intermediate rounded to the storage type, a->y not necessarily written
            a, b = self._desc(call.args[0]), self._desc(call.args[1])
            t = conv_epilogue(a, conv_acc(a))
            t = t.half().float() if a.y.dtype == L.F16 else t
            ref = conv_epilogue(b, conv_acc(b, x=t))
            self.orig(call, st)
            torch.cuda.synchronize()
            self._ck(self._what('pair', b) + '/y', read_out(b), ref, self.tol_out)
        elif name == 'myolo_conv_dgrad_s2':
            arr, n = call.args[0], call.args[1]
            ds = [arr[i].contents for i in range(n)]
            pres = [self._conv_pre(d) for d in ds]
            self.orig(call, st)
            torch.cuda.synchronize()
            for i, (d, pre) in enumerate(zip(ds, pres)):
                self._conv_post(d, pre, self._what(f'dgrad_s2.{i}', d))
        elif name == 'myolo_conv_dgrad_bn':
            d, f = self._desc(call.args[0]), self._desc(call.args[1])
            dy_ref, d0, d1 = apply_fold_ref(d, f)
            K = f.y.c
            g0 = (read_f32(f.dgamma, K) if _ptr(f.dgamma) else None, read_f32(f.dbeta, K) if _ptr(f.dbeta) else None)
            y0 = read_out(d) if d.accumulate else None
            b0 = [stat_sums(d.bnb[i].dsum, d.bnb[i].c1 - d.bnb[i].c0) for i in range(d.nbnb)]
            self.orig(call, st)
            torch.cuda.synchronize()
            what = self._what('dgrad_bn', d)
            dy_got = read_tensor(f.dy)
            self._ck(what + '/dy', dy_got, dy_ref, self.tol_out)
            # the MFMAs consume the fp16-rounded dy the launch formed: the dgrad is checked against the convolution of ITS dy
            ref = conv_acc(d, x=dy_got)
            if d.accumulate:
                ref = ref + y0
            got = read_out(d)
            self._ck(what + '/gx', got, ref, self.tol_out)
            if g0[0] is not None:
                self._ck(what + '/dgamma', read_f32(f.dgamma, K) - g0[0], d1, self.tol_stat)
            if g0[1] is not None:
                self._ck(what + '/dbeta', read_f32(f.dbeta, K) - g0[1], d0, self.tol_stat)
            for i in range(d.nbnb):
                sg = d.bnb[i]
                self._ck(what + f'/bnb{i}', (stat_sums(sg.dsum, sg.c1 - sg.c0) - b0[i]).float(), bnb_ref(sg, got).float(), self.tol_stat)
        else:                                   # myolo_conv_wgrad
            wd = self._desc(call.args[0])
            cout = wd.cout if wd.cout > 0 else wd.dy.c
            cin = wd.cin if wd.cin > 0 else wd.x.c
            n = cout * cin * wd.ntaps
            w0 = read_f32(wd.dw, n)
            b0 = read_f32(wd.db, cout) if _ptr(wd.db) else None
            ref_w, ref_b = wgrad_ref(wd)
            self.orig(call, st)
            torch.cuda.synchronize()
            what = f'wgrad[{wd.x.n}x{wd.x.h}x{wd.x.w}x{cin}->{wd.dy.h}x{wd.dy.w}x{cout} t{wd.ntaps}s{wd.stride}]#{self.k}'
            self._ck(what + '/dw', read_f32(wd.dw, n) - w0, ref_w.reshape(-1), self.tol_w)
            if b0 is not None:
                self._ck(what + '/db', read_f32(wd.db, cout) - b0, ref_b, self.tol_w)

    # ---- BatchNorm family (include/myolo.h:192-240): forward, backward reduce, backward apply; plain and split (two parameter sets) ----
    def _run_bn(self, call, st, name):
        a = call.args
        split = None
        if name.endswith('_split'):
            split = self._desc(a[-1])
            name = name[:-6]
        if name == 'myolo_bn_act_bwd_fused':
            # round 6: reduce + apply in one launch -- everything both passes store, against first principles over the operands
            # (the sums the apply formula uses are the REFERENCE's, not the kernel's)
            split = self._desc(a[12]) if a[12] is not None else None
            god, yd, dyd, grd, racc = self._desc(a[0]), self._desc(a[1]), self._desc(a[9]), self._desc(a[10]), int(_fv(a[11]))
            B = _BnArgs(yd, split)
            C, M = B.C, B.M * B.scale
            act = int(_fv(a[5]))
            gout, y = read_tensor(god, channels=C), read_tensor(yd)
            sv = read_f32(a[2], 2 * C)
            mean, invstd = sv[:C], sv[C:]
            gamma = _split_vec(a[3], split.gamma2 if split else None, C, B.cs)
            beta = _split_vec(a[4], split.beta2 if split else None, C, B.cs)
            xhat = (y - mean) * invstd
            dz = gout * act_grad(xhat * gamma + beta, act)
            ds = torch.stack([dz.double().sum((0, 1, 2)), (dz * xhat).double().sum((0, 1, 2))])
            dy_ref = (gamma * invstd) * (dz - (ds[0] / M).float() - xhat * (ds[1] / M).float())
            d0 = stat_sums(a[6], C)
            g0 = (_split_vec(a[7], split.dgamma2 if split else None, C, B.cs), _split_vec(a[8], split.dbeta2 if split else None, C, B.cs))
            gres0 = read_tensor(grd, channels=C) if (grd.ptr and racc) else None
            self.orig(call, st)
            torch.cuda.synchronize()
            what = f'bn_bwd_fused[{yd.n}x{yd.h}x{yd.w}x{C}' + ('+gres' if grd.ptr else '') + ('+split' if split is not None and B.cs < C else '') + f']#{self.k}'
            assert int(read_u32(a[13], 19 * 32)[18 * 32]) == 0, what + ': the grid barrier timed out'
            self._ck(what + '/dsum', (stat_sums(a[6], C) - d0).float(), ds.float(), self.tol_stat)
            self._ck(what + '/dy', read_tensor(dyd), dy_ref, self.tol_out)
            if grd.ptr:
                self._ck(what + '/gres', read_tensor(grd, channels=C), gout + (gres0 if gres0 is not None else 0), self.tol_out)
            g1 = _split_vec(a[7], split.dgamma2 if split else None, C, B.cs)
            b1 = _split_vec(a[8], split.dbeta2 if split else None, C, B.cs)
            self._ck(what + '/dgamma', g1 - g0[0], (ds[1] / B.scale).float(), self.tol_stat)
            self._ck(what + '/dbeta', b1 - g0[1], (ds[0] / B.scale).float(), self.tol_stat)
            return
        if name == 'myolo_bn_act_fwd':
            yd, res_d, out_d = self._desc(a[0]), self._desc(a[11]), self._desc(a[12])
            B = _BnArgs(yd, split)
            C, M = B.C, B.M * B.scale
            has_bn = bool(_ptr(a[1]))
            y = read_tensor(yd)
            res = read_tensor(res_d, channels=C) if res_d.ptr else None
            eps, mom, act = float(_fv(a[8])), float(_fv(a[9])), int(_fv(a[10]))
            if has_bn:
                stats = stat_sums(a[1], C)
                gamma = _split_vec(a[2], split.gamma2 if split else None, C, B.cs)
                beta = _split_vec(a[3], split.beta2 if split else None, C, B.cs)
                rm0 = _split_vec(a[4], split.running_mean2 if split else None, C, B.cs)
                rv0 = _split_vec(a[5], split.running_var2 if split else None, C, B.cs)
                mean = stats[0] / M
                var = (stats[1] / M - mean * mean).clamp_min(0)
                invstd = 1.0 / torch.sqrt(var + eps)
                out_ref = act_fn(((y.double() - mean) * invstd * gamma.double() + beta.double()).float(), act)
            else:
                out_ref = act_fn(y, act)
            if res is not None:
                out_ref = out_ref + res
            self.orig(call, st)
            torch.cuda.synchronize()
            what = f'bn_fwd[{yd.n}x{yd.h}x{yd.w}x{C}' + ('+res' if res is not None else '') + ('+split' if split is not None and B.cs < C else '') + f']#{self.k}'
            self._ck(what + '/out', read_tensor(out_d), out_ref, self.tol_out)
            if has_bn:
                sv = read_f32(a[7], 2 * C)
                self._ck(what + '/saved_mean', sv[:C], mean.float(), 1e-4)
                self._ck(what + '/saved_invstd', sv[C:], invstd.float(), 1e-4)
                if rm0 is not None:
                    rm1 = _split_vec(a[4], split.running_mean2 if split else None, C, B.cs)
                    rv1 = _split_vec(a[5], split.running_var2 if split else None, C, B.cs)
                    self._ck(what + '/running_mean', rm1, ((1 - mom) * rm0.double() + mom * mean).float(), 1e-4)
                    self._ck(what + '/running_var', rv1, ((1 - mom) * rv0.double() + mom * var * M / max(M - 1, 1)).float(), 1e-4)
            return
        god, yd = self._desc(a[0]), self._desc(a[1])
        B = _BnArgs(yd, split)
        C, M = B.C, B.M * B.scale
        act = int(_fv(a[5]))
        gout, y = read_tensor(god, channels=C), read_tensor(yd)
        has_bn = bool(_ptr(a[2]))
        if has_bn:
            sv = read_f32(a[2], 2 * C)
            mean, invstd = sv[:C], sv[C:]
            gamma = _split_vec(a[3], split.gamma2 if split else None, C, B.cs)
            beta = _split_vec(a[4], split.beta2 if split else None, C, B.cs)
            xhat = (y - mean) * invstd
            dz = gout * act_grad(xhat * gamma + beta, act)
        else:
            dz = gout * act_grad(y, act)
        if name == 'myolo_bn_act_bwd_reduce':
            d0 = stat_sums(a[6], C)
            self.orig(call, st)
            torch.cuda.synchronize()
            ref = torch.stack([dz.double().sum((0, 1, 2)), (dz * xhat).double().sum((0, 1, 2))])
            self._ck(f'bn_bwd_reduce[{yd.n}x{yd.h}x{yd.w}x{C}]#{self.k}/dsum', (stat_sums(a[6], C) - d0).float(), ref.float(), self.tol_stat)
            return
        # apply: dy = gamma * invstd * (dz - dsum0 / M - xhat * dsum1 / M); dgamma += dsum1, dbeta += dsum0 (/ count_scale); gres (+)= gout
        dyd, grd, racc = self._desc(a[9]), self._desc(a[10]), int(_fv(a[11]))
        if has_bn:
            ds = stat_sums(a[6], C)
            dy_ref = (gamma * invstd) * (dz - (ds[0] / M).float() - xhat * (ds[1] / M).float())
            g0 = (_split_vec(a[7], split.dgamma2 if split else None, C, B.cs), _split_vec(a[8], split.dbeta2 if split else None, C, B.cs))
        else:
            dy_ref, g0 = dz, (None, None)
        gres0 = read_tensor(grd, channels=C) if (grd.ptr and racc) else None
        self.orig(call, st)
        torch.cuda.synchronize()
        what = f'bn_bwd_apply[{yd.n}x{yd.h}x{yd.w}x{C}' + ('+gres' if grd.ptr else '') + ('+split' if split is not None and B.cs < C else '') + f']#{self.k}'
        self._ck(what + '/dy', read_tensor(dyd), dy_ref, self.tol_out)
        if grd.ptr:
            self._ck(what + '/gres', read_tensor(grd, channels=C), gout + (gres0 if gres0 is not None else 0), self.tol_out)
        if g0[0] is not None:
            g1 = _split_vec(a[7], split.dgamma2 if split else None, C, B.cs)
            self._ck(what + '/dgamma', g1 - g0[0], (ds[1] / B.scale).float(), self.tol_stat)
        if g0[1] is not None:
            b1 = _split_vec(a[8], split.dbeta2 if split else None, C, B.cs)
            self._ck(what + '/dbeta', b1 - g0[1], (ds[0] / B.scale).float(), self.tol_stat)
