import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "double-yolo-kaist_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture
def tiles(request, monkeypatch):
    """Tile choice of the plans a test compiles.  `pinned`: the autotuner is off and nothing is read from or written to the tune
    cache, every kernel runs the deterministic default tile / split-K of its descriptor -- the fp32 summation order is then part
    of the fixture (it changes only when the code does), so trajectory bounds can be tight.  `tuned`: the product default (the
    autotuner picks per box); tests that take both keep loose, clearly labelled SMOKE bounds for it."""
    mode = getattr(request, "param", "pinned")
    if mode == "pinned":
        monkeypatch.setenv("DYK_AUTOTUNE", "0")
        monkeypatch.setenv("DYK_TUNE_CACHE", "0")
    return mode
