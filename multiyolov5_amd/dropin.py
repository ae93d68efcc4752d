"""Drop-in rebinding of the reference's import paths to the MI355X mirror.

    import multiyolov5_amd.dropin as dropin; dropin.install()        # top of train.py / detect.py / test.py, or sitecustomize

After install():
  * `models.common`, `models.yolo`, `models.experimental` resolve to the mirror modules, so `parse_model`'s eval (yolo.py:381),
    `from models.yolo import Model` (train.py:20) and `attempt_load` (detect.py:34) build libmyolo launch plans;
  * the mirror classes report the REFERENCE's module names (`Model.__module__ == 'models.yolo'`), so a checkpoint written by
    train.py:481-499 (`torch.save({'model': deepcopy(model).half(), 'ema': ...})`) names `models.yolo.Model`, `models.common.Conv`
    ... exactly like one written by the reference, and unpickles in a stock reference checkout (and vice versa);
  * `utils.loss.ComputeLoss / SegmentationLosses / OhemCELoss`, `utils.general.non_max_suppression` and
    `utils.torch_utils.ModelEMA` of the reference's own utils modules are rebound when those modules are importable.
uninstall() restores everything (tests)."""
import importlib
import sys
import types

_REF = {'common': 'models.common', 'yolo': 'models.yolo', 'experimental': 'models.experimental'}
_state = None


def _mirror():
    from .models import common, experimental, yolo
    return {'common': common, 'yolo': yolo, 'experimental': experimental}


def install(rebind_utils=True):
    global _state
    if _state is not None:
        return
    saved_mods, renamed, rebound = {}, [], []
    if 'models' not in sys.modules:
        try:
            importlib.import_module('models')                  # the reference checkout's package, when on sys.path
        except Exception:  # noqa: BLE001
            pkg = types.ModuleType('models')
            pkg.__path__ = []
            sys.modules['models'] = pkg
            saved_mods['models'] = None
    for key, mod in _mirror().items():
        ref = _REF[key]
        saved_mods[ref] = sys.modules.get(ref)
        sys.modules[ref] = mod
        setattr(sys.modules['models'], key, mod)
        for name, obj in list(vars(mod).items()):
            if isinstance(obj, type) and obj.__module__ == mod.__name__:
                obj.__module__ = ref
                renamed.append((obj, mod.__name__))
    if rebind_utils:
        from .utils import general as g, loss as l, torch_utils as t
        for modname, names, src in (('utils.loss', ('ComputeLoss', 'SegmentationLosses', 'OhemCELoss'), l),
                                    ('utils.general', ('non_max_suppression',), g),
                                    ('utils.torch_utils', ('ModelEMA',), t)):
            try:
                target = importlib.import_module(modname)
            except Exception:  # noqa: BLE001  (no reference checkout on sys.path: nothing to rebind)
                continue
            if getattr(target, '__name__', '').startswith('multiyolov5_amd'):
                continue
            for n in names:
                rebound.append((target, n, getattr(target, n, None)))
                setattr(target, n, getattr(src, n))
    _state = (saved_mods, renamed, rebound)


def uninstall():
    global _state
    if _state is None:
        return
    saved_mods, renamed, rebound = _state
    for obj, orig in renamed:
        obj.__module__ = orig
    for target, n, old in rebound:
        if old is None:
            delattr(target, n)
        else:
            setattr(target, n, old)
    for ref, old in saved_mods.items():
        if old is None:
            sys.modules.pop(ref, None)
        else:
            sys.modules[ref] = old
    _state = None


class installed:
    """context manager form (tests)"""

    def __enter__(self):
        install()
        return self

    def __exit__(self, *exc):
        uninstall()
