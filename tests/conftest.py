import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "seed-story_amd")
for p in (PKG, os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    from safetensors.torch import load_file
    g = load_file(os.path.join(ROOT, "tests", "golden", "hotpath_tiny.safetensors"))
    with open(os.path.join(ROOT, "tests", "golden", "hotpath_tiny.json")) as f:
        meta = json.load(f)
    return g, meta
