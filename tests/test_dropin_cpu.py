"""INTEGRATION.md section 1 on the CPU: the rebinding (multiyolov5_amd.dropin), and the checkpoint layout of train.py:481-499 written
by the mirror under it -- the file names only the reference's module paths and unpickles in a STOCK reference checkout (run here
in a subprocess when /root/reference exists; that subprocess executes the real reference forward on the unpickled object and
its outputs are checked against the oracle)."""
import io
import os
import pickle
import pickletools
import subprocess
import sys
from copy import deepcopy

import numpy as np
import pytest
import torch

from tests.util import CFG, ROOT, TAGS, load_cfg, synth_sd


def _globals_of(path):
    """every (module, name) a pickle file refers to"""
    import zipfile
    out = set()
    with zipfile.ZipFile(path) as z:
        data = z.read([n for n in z.namelist() if n.endswith('data.pkl')][0])
    strs = []
    for op, arg, _ in pickletools.genops(data):
        if op.name in ('BINUNICODE', 'SHORT_BINUNICODE', 'UNICODE'):
            strs.append(arg)
        elif op.name == 'GLOBAL':
            out.add(tuple(arg.split(' ')))
        elif op.name == 'STACK_GLOBAL':
            out.add((strs[-2], strs[-1]))
    return out


def _write_ckpt(path, tag):
    """train.py:481-499 under the drop-in: model + EMA copies as fp16, optimizer state"""
    import multiyolov5_amd.dropin as dropin
    with dropin.installed():
        from models.yolo import Model                        # the reference's import line (train.py:20)
        from multiyolov5_amd.utils.torch_utils import ModelEMA
        assert Model.__module__ == 'models.yolo'
        m = Model(os.path.join(CFG, TAGS[tag]))
        sd = synth_sd(tag)
        m.load_state_dict(sd, strict=True)
        m.names = [f'class{i}' for i in range(10)]
        m.nc, m.gr, m.hyp = 10, 1.0, {'obj': 1.0}
        ema = ModelEMA(m)
        opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, nesterov=True)
        ckpt = {'epoch': 3, 'best_fitness': 0.5, 'training_results': '',
                'model': deepcopy(m).half(), 'ema': deepcopy(ema.ema).half(), 'updates': ema.updates,
                'optimizer': opt.state_dict(), 'wandb_id': None}
        torch.save(ckpt, path)
    return sd


def test_dropin_install_uninstall_restores_everything():
    import multiyolov5_amd.dropin as dropin
    import multiyolov5_amd.models.yolo as y
    before = {k: sys.modules.get(k) for k in ('models', 'models.yolo', 'models.common', 'models.experimental')}
    dropin.install()
    try:
        assert sys.modules['models.yolo'] is y and y.Model.__module__ == 'models.yolo' and y.Detect.__module__ == 'models.yolo'
        from models.common import Conv, C3
        assert Conv.__module__ == 'models.common' and C3.__module__ == 'models.common'
    finally:
        dropin.uninstall()
    assert y.Model.__module__ == 'multiyolov5_amd.models.yolo'
    assert {k: sys.modules.get(k) for k in before} == before


@pytest.mark.parametrize('tag', ['s_psp', 's_bise'])
def test_checkpoint_written_under_dropin_names_only_reference_paths_and_loads_back(tag, tmp_path):
    from multiyolov5_amd.models import experimental as X
    from multiyolov5_amd.models.yolo import Model
    path = str(tmp_path / 'last.pt')
    sd = _write_ckpt(path, tag)
    names = _globals_of(path)
    mods = {m for m, _ in names}
    assert not any(m.startswith('multiyolov5_amd') for m in mods), sorted(n for n in names if n[0].startswith('multiyolov5_amd'))
    assert ('models.yolo', 'Model') in names and ('models.common', 'Conv') in names and ('models.yolo', 'Detect') in names
    allowed = ('models.', 'torch', 'collections', 'pathlib', 'builtins', '__builtin__', 'numpy', '_codecs', 'copyreg')
    assert all(m.startswith(allowed) for m in mods), sorted(mods)
    # and the mirror loads its own checkpoint through the bridge (no reference, no dropin active)
    m = X.attempt_load(path, map_location='cpu')
    assert isinstance(m, Model) and not m.training and m.names[3] == 'class3'
    ck = X.load_checkpoint(path, map_location='cpu')
    assert ck['epoch'] == 3 and set(ck['model'].state_dict()) == set(sd)


@pytest.mark.skipif(not os.path.isdir('/root/reference/models'), reason='needs the reference checkout (build container only)')
@pytest.mark.parametrize('tag', ['s_psp', 's_bise'])
def test_mirror_checkpoint_unpickles_and_runs_in_the_stock_reference(tag, tmp_path):
    """the reverse direction of models/experimental.py's bridge: the REAL reference (no multiyolov5_amd on its path) torch.loads a
    checkpoint written by the mirror and runs its own forward on it; outputs == oracle on the same weights"""
    from oracle import model_ref, synth
    path = str(tmp_path / 'last.pt')
    sd = _write_ckpt(path, tag)
    out = str(tmp_path / 'ref_out.npz')
    code = f'''
import sys, numpy as np, torch
sys.path.insert(0, {ROOT!r})
from oracle import ref_shim, synth
ref = ref_shim.install()
sys.path.remove({ROOT!r})
for k in [k for k in sys.modules if k.startswith('multiyolov5_amd')]:
    del sys.modules[k]
ck = torch.load({path!r}, map_location='cpu', weights_only=False)
assert not any(k.startswith('multiyolov5_amd') for k in sys.modules), 'the checkpoint pulled the mirror in'
m = ck['model'].float()
assert type(m).__module__ == 'models.yolo' and type(m) is ref.yolo.Model
m.eval()
x = synth.synth_images(1, 64, 128, seed=1)
with torch.no_grad():
    (pred, raw), seg = m(x)
e = ck['ema'].float().fuse().eval()                  # attempt_load's chain (experimental.py:119)
with torch.no_grad():
    (pe, _), se = e(x)
np.savez({out!r}, pred=pred.numpy(), seg=seg.numpy(), pred_fused=pe.numpy(), seg_fused=se.numpy())
'''
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    g = np.load(out)
    sd16 = {k: (v.half().float() if v.is_floating_point() else v.clone()) for k, v in sd.items()}      # the checkpoint is fp16
    x = synth.synth_images(1, 64, 128, seed=1)
    with torch.no_grad():
        (pred, _), seg = model_ref.forward(load_cfg(tag), {k: v.clone() for k, v in sd16.items()}, x, training=False)
        (pf, _), sf = model_ref.forward(load_cfg(tag), model_ref.fuse_state_dict({k: v.clone() for k, v in sd16.items()}), x, training=False)
    np.testing.assert_allclose(g['pred'], pred.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(g['seg'], seg.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(g['pred_fused'], pf.numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(g['seg_fused'], sf.numpy(), rtol=1e-3, atol=1e-3)
