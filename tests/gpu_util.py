"""helpers for the -m gpu parity tests: error metrics with readable diagnostics."""
import json
import os

import numpy as np
import torch

LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'parity_log.jsonl')


def rel_err(a, b):
    a = torch.as_tensor(a).detach().float().cpu().reshape(-1)
    b = torch.as_tensor(b).detach().float().cpu().reshape(-1)
    assert a.shape == b.shape, (a.shape, b.shape)
    den = b.norm().item()
    return (a - b).norm().item() / (den if den > 0 else 1.0), (a - b).abs().max().item()


def check(name, got, ref, tol, atol=None, collect=None):
    """relative L2 error <= tol (and optional max-abs <= atol).  Logs every comparison to gpurun_out/parity_log.jsonl."""
    rel, mx = rel_err(got, ref)
    if atol is not None:      # absolute bound scales with the reference magnitude (1e-3 'at fp32' for O(1) logits)
        atol = atol * max(1.0, float(torch.as_tensor(ref).detach().float().abs().max()))
    bad = not np.isfinite(rel) or rel > tol or (atol is not None and mx > atol)
    rec = {'name': name, 'rel_l2': rel, 'max_abs': mx, 'tol': tol, 'ok': not bad}
    try:
        os.makedirs(os.path.dirname(LOG), exist_ok=True)
        with open(LOG, 'a') as f:
            f.write(json.dumps(rec) + '\n')
    except OSError:
        pass
    if collect is not None:
        if bad:
            collect.append(f'{name}: rel_l2={rel:.3e} max_abs={mx:.3e} tol={tol:.1e}')
        return not bad
    assert not bad, f'{name}: rel_l2={rel:.3e} max_abs={mx:.3e} tol={tol:.1e}'
    return True


TOL = {torch.float32: 2e-4, torch.float16: 2e-2}


def pool_replay(model, tie_eps=1e-4):
    """-> (fn(x, k), stats) for oracle.model_ref.forward(maxpool_fn=fn).  At 512x1024 the SPP max-pools see ~1e5 windows per plane and a
    handful have top-2 values closer than the rounding noise of two summation orders (the BatchNorm statistics in front are fp32 atomics):
    the arg-max of those differs between the product and the oracle and moves ~2e-3 of the gradient energy of everything upstream
    (a 2.5e-2 gate on the whole backbone).  The product's own choices (the index planes its backward uses) are read back and
    replayed in the oracle: the pool output is x gathered at the product's index (proven to be a maximum of the ORACLE's window up to
    `tie_eps` of max|x|), autograd then routes the gradient like the product does.  stats counts the windows where the choices differed."""
    import torch.nn.functional as F
    from multiyolov5_amd import engine as E
    planes = []
    for h in model.__dict__['_plans'].values():
        for op in h.plan.ops:
            if isinstance(op, E.SppPoolOp) and getattr(op, 'idx', None) is not None:
                x = op.x
                idx = op.idx.view(3, x.n, x.h, x.w, x.c).permute(0, 1, 4, 2, 3).cpu().long()
                for j, k in enumerate((5, 9, 13)):
                    planes.append((k, idx[j]))
    it = iter(planes)
    stats = {'pools': len(planes), 'windows': 0, 'flipped': 0, 'worst_gap': 0.0}

    def fn(x, k):
        kk, idx = next(it)
        assert kk == k and tuple(idx.shape) == tuple(x.shape), (kk, k, idx.shape, x.shape)
        n, c, h, w = x.shape
        r = k // 2
        ys = torch.arange(h).view(1, 1, h, 1) + torch.div(idx, k, rounding_mode='floor') - r
        xs = torch.arange(w).view(1, 1, 1, w) + idx % k - r
        assert int(ys.min()) >= 0 and int(ys.max()) < h and int(xs.min()) >= 0 and int(xs.max()) < w
        flat = (ys * w + xs).flatten(2)
        got = x.flatten(2).gather(2, flat).view_as(x)
        with torch.no_grad():
            ref, ridx = F.max_pool2d(x, k, 1, r, return_indices=True)
            gap = float((ref - got).max())
            lim = tie_eps * float(x.abs().max())
            assert gap <= lim, f'max-pool replay: the product chose a non-maximum (gap {gap:.3e} > {lim:.3e})'
            stats['windows'] += idx.numel()
            stats['flipped'] += int((ridx.flatten(2) != flat).sum())
            stats['worst_gap'] = max(stats['worst_gap'], gap)
        return got
    return fn, stats
