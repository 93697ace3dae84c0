import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding as ob
    ob.build()
    return ob


@pytest.fixture(scope="session")
def torch():
    import torch
    assert torch.cuda.is_available(), "these tests need an MI355X"
    return torch


@pytest.fixture(scope="session")
def pt():
    import pathtracer_amd
    return pathtracer_amd
