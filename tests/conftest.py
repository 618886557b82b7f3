import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The CPU oracle (the checker) runs through torch's intra-op pool.  The GPU box shows 256 hardware threads of which a
    # one-GPU job owns a 16-CPU share: left at the default, every oracle convolution spins 256 oversubscribed threads.
    try:
        import torch
        torch.set_num_threads(min(16, os.cpu_count() or 1))
    except Exception:           # noqa: BLE001
        pass


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')
