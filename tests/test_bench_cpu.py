"""bench.py launch contract: `python bench.py --gpus N` must really run N ranks and say so.

The driver uses two forms: `python bench.py --gpus N ...` and `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`.
Both are exercised here on the CPU (gloo, `--dry-dist`: process group + world-size checks + the header line, no model)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    e = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e['MYOLO_DIST_BACKEND'] = 'gloo'
    e['OMP_NUM_THREADS'] = '1'
    return e


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_plain_command_line_spawns_one_rank_per_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--dry-dist'],
                       env=_env(), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rec = _json_line(r.stdout)
    assert rec['n_gpus'] == 2 and rec['config']['parallelism'] == 'dp2'


@pytest.mark.timeout(300)
def test_launcher_form_and_world_size_mismatch_is_fatal():
    base = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
            '--master-port', '29731', os.path.join(ROOT, 'bench.py')]
    r = subprocess.run(base + ['--gpus', '2', '--dry-dist', '--ddp', 'stock'], env=_env(), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rec = _json_line(r.stdout)
    assert rec['n_gpus'] == 2 and rec['config']['ddp'] == 'stock'
    # two ranks started but the command line says one GPU: no line may be printed
    r = subprocess.run(base + ['--gpus', '1', '--dry-dist'], env=_env(), capture_output=True, text=True, timeout=280)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]


def test_single_gpu_form_needs_no_launcher():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--dry-dist'], env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-3000:]
    assert _json_line(r.stdout)['n_gpus'] == 1


@pytest.mark.timeout(600)
def test_eight_ranks_exchange_the_gradient_slices_in_backward_completion_order():
    """VERDICT r5 item 9 (code side only: no node with more than one GPU): `python bench.py --gpus 8 --dry-dist` = 8 gloo ranks, each with a
    dry-built training plan and parallel.GradReducer.  The slices must be handed over in the order the backward completes them (head and
    neck first, the stem last: descending `ready_after_op`), tile the flat gradient buffer exactly once, reduce to the mean over the
    8 ranks, and the N > 1 line's diagnosis fields must be populated"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0', '--dry-dist'],
                       env=_env(), capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rec = _json_line(r.stdout)
    assert rec['n_gpus'] == 8 and rec['config']['parallelism'] == 'dp8'
    d = rec['grad_exchange_detail']
    assert d['world'] == 8 and d['buckets'] == 3 and d['pre_divide_then_sum'] and d['slices_issued_last_step'] == 3
    order = [tuple(x) for x in d['slice_issue_order']]
    buckets = [tuple(b) for b in rec['bucket_completion_order']]
    assert order == [(a, b) for a, b, _ in buckets]                                       # issue order == completion order
    assert [b[2] for b in buckets] == sorted((b[2] for b in buckets), reverse=True)      # ... which is descending op index (the backward's direction)
    assert buckets[0][1] == rec['flat_grad_elements'] and buckets[-1][0] == 0             # Detect / head parameters first, the Focus conv last
    cover = sorted(order)
    assert cover[0][0] == 0 and cover[-1][1] == rec['flat_grad_elements'] and all(cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1))
    segs = rec['backward_segments']
    assert all(h > l for h, l in segs) and all(segs[i][1] == segs[i + 1][0] for i in range(len(segs) - 1)) and segs[-1][1] == 0
    assert rec['exchange_max_abs_err_over_ranks'] < 1e-6
    assert isinstance(rec['allreduce_exposed_ms'], float) and rec['allreduce_exposed_ms'] >= 0.0
