import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vilbert-multi-task_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    config.addinivalue_line("markers", "slow: full-size (BASELINE.json batch) comparisons against the CPU oracle")
    # the CPU oracle on a many-core host: torch's default of one thread per core (256 on the GPU box) is far
    # slower than a few dozen threads for these matrix sizes (bench.py's cpu_baseline calibrates the same way)
    import torch
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def per_op_path():
    """Tests that watch the per-op launchers (ops.py / ops16.py call counts, the dropout hand-over between autograd nodes)
    switch the whole-layer launcher (vilbert/layers.py, round 6) off for their duration."""
    from vilbert import layers
    prev = layers.set_native(False)
    yield
    layers.set_native(prev)
