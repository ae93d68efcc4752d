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
