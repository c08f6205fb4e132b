import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """native libraries are git-ignored build products: build them if a fresh checkout lacks them"""
    libs = [os.path.join(ROOT, "mp-mvs_amd", "csrc", "libmpmvs_hip.so"), os.path.join(ROOT, "mp-mvs_amd", "csrc", "libmpmvs_hip_q8.so"),
            os.path.join(ROOT, "mp-mvs_amd", "host", "libmpmvs_host.so"),
            os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in libs):
        import __graft_entry__ as g
        g.build()


@pytest.fixture(scope="session")
def pm():
    _ensure_built()
    return importlib.import_module("mp-mvs_amd")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.build()
    # the GPU boxes show 256 hardware threads to a container with 16 CPUs of quota: an OpenMP team of 256 is throttled
    binding.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    return binding


@pytest.fixture(scope="session")
def hostlib(pm):
    """ctypes view of libmpmvs_host.so (C++ host layer); it links libmpmvs_hip.so but needs no GPU to load"""
    return importlib.import_module("mp-mvs_amd.hostlib")


@pytest.fixture(scope="session")
def engine():
    """The HIP library; GPU tests fail (not skip) if it is missing."""
    eng = importlib.import_module("mp-mvs_amd.engine")
    eng.load()
    return eng
