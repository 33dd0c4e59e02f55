"""Shared test plumbing: markers, package loader (the package directory carries a
hyphen, so it is loaded by path), oracle binding."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load_package():
    name = "pi_slam_fusion_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "pi-slam-fusion_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pf():
    return load_package()


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    o.lib()
    return o
