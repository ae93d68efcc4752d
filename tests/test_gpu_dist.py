"""-m gpu: the N > 1 code path with real device tensors -- 2 ranks on ONE GPU over gloo, in their own processes (an in-process RCCL /
gloo group made later tests of the same interpreter flaky): parallel.GradReducer slices issued by the staged backward, the reduced flat
gradient equals the mean of the two single-rank gradients; nn.SyncBatchNorm against the whole-batch oracle.  (RCCL itself needs 2 GPUs:
that is the driver's SCALE run.)"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_worker(script):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', script)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    assert out.count('-> OK') == 2, out[-3000:]


def test_two_ranks_on_one_gpu_reduce_to_the_mean_gradient():
    _run_worker('dist_worker.py')


def test_sync_batchnorm_two_ranks_match_the_whole_batch_oracle():
    """train.py:190-193 --sync-bn: torch's convert_sync_batchnorm on the mirror, 2 ranks x 2 images; the CPU oracle on all 4 images is the
    checker (outputs, reduced gradients, running statistics)"""
    _run_worker('dist_worker_syncbn.py')


@pytest.mark.parametrize('ddp', ['reducer', 'stock'])
def test_bench_gpus_2_runs_two_ranks_and_says_so(ddp):
    """the driver's own command form, `python bench.py --gpus 2 ...` (no launcher): two ranks (here both on the one GPU, over gloo), two
    real joint steps each, ONE JSON line with n_gpus == 2 and the whole-job batch.  `--ddp stock` wraps the model in torch's
    DistributedDataParallel as train.py:243-245 does."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY='0', MYOLO_DIST_BACKEND='gloo')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '2', '--img', '128', '256',
           '--no-infer', '--no-cpu-baseline', '--no-kernel-timing', '--ddp', ddp]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-2000:] + r.stderr.decode()[-4000:]
    lines = [ln for ln in out.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out[-2000:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['config']['global_batch'] == 4 and rec['config']['parallelism'] == 'dp2'
    assert rec['checks']['optimizer_steps_skipped'] == 0
