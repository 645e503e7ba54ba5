import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

PKG = "sph-poiseuille-flow_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module(PKG)


@pytest.fixture(scope="session")
def cfgmod():
    return importlib.import_module(PKG + ".config")


@pytest.fixture(scope="session")
def geom():
    return importlib.import_module(PKG + ".geometry")


@pytest.fixture(scope="session")
def profmod():
    return importlib.import_module(PKG + ".profile")


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc  # oracle/oracle.py -- test infrastructure
    orc.build()
    return orc


@pytest.fixture(scope="session")
def capi():
    return importlib.import_module(PKG + ".capi")


@pytest.fixture(scope="session")
def mex():
    return importlib.import_module(PKG + ".mex_surface")


@pytest.fixture(scope="session")
def driver():
    return importlib.import_module(PKG + ".driver")
