import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("sfm-gms_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("sfm-gms_amd.synth")


@pytest.fixture(scope="session")
def oracle():
    import gms_oracle
    gms_oracle.load()
    return gms_oracle


@pytest.fixture(scope="session")
def ctx(pkg):
    c = pkg.GmsContext(0)
    yield c
    c.close()
