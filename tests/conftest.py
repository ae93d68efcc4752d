import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(autouse=True)
def _release_plans():
    """launch plans are reference cycles (plan <-> ops <-> ctypes descriptors) holding every activation buffer: collect them after each
    test instead of whenever the allocator's generation counters get there (a dry CPU plan at batch 16 is gigabytes of host memory)"""
    yield
    import gc
    gc.collect()
