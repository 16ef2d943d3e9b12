import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _make(directory, target):
    subprocess.run(["make", "-C", os.path.join(ROOT, directory), target], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)


@pytest.fixture(scope="session")
def port():
    """oracle/liboracle.so — the plain-C restatement (the checker)."""
    from oracle import bindings
    if not os.path.exists(bindings.PORT_LIB):
        _make("oracle", "liboracle.so")
    return bindings.PortOracle()


@pytest.fixture(scope="session")
def ref():
    """oracle/_ref/libref_oracle.so — the real reference; needs /root/reference to (re)build."""
    from oracle import bindings
    if not os.path.exists(bindings.REF_LIB):
        if not os.path.exists("/root/reference/Raytracer/main.cpp"):
            pytest.skip("reference sources not present and oracle/_ref not prebuilt")
        _make("oracle", "_ref/libref_oracle.so")
    return bindings.RefOracle()


@pytest.fixture(scope="session")
def emul():
    """tests/emul/libemul.so — host build of the device functions (CPU tests only)."""
    import util
    path = os.path.join(ROOT, "tests", "emul", "libemul.so")
    _make("tests/emul", "libemul.so")   # always: make compares time stamps, and a stale host build of the device headers would test old code
    return util.Emul(path)


@pytest.fixture(scope="session")
def gpu():
    """A Renderer on cuda:0.  Fails (does not skip) if the HIP library is missing."""
    from raytracer_amd.render import Renderer
    r = Renderer(0)
    yield r
    r.close()
