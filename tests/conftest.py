import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "host_twin")):
    if p not in sys.path:
        sys.path.insert(0, p)

CORNELL = os.path.join(ROOT, "data", "cornell-box.xml")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ptrs():
    return importlib.import_module("pathtracer-rs_amd")


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def scenes():
    return importlib.import_module("pathtracer-rs_amd.scenes")
