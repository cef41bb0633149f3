import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("ray-tracer_amd")


@pytest.fixture(scope="session")
def scene():
    return importlib.import_module("ray-tracer_amd.scene")


@pytest.fixture(scope="session")
def scenes():
    return importlib.import_module("ray-tracer_amd.scenes")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding

    binding.lib()
    return binding
