"""Shared fixtures.  `-m "not gpu"` runs on the CPU (oracle, host library, ABI symbols, sharding);
`-m gpu` runs the HIP path against the oracle through the C ABI."""
import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run on the GPU box)")


def _ensure_built():
    need = [os.path.join(ROOT, "raytracer-rs_amd", "lib", "librt_hip.so"), os.path.join(ROOT, "raytracer-rs_amd", "lib", "librt_host.so"),
            os.path.join(ROOT, "oracle", "build", "liboracle_rt.so")]
    if all(os.path.exists(p) for p in need):
        return
    subprocess.run(["make", "-C", ROOT, "-j4"], check=True, stdout=subprocess.DEVNULL)


_ensure_built()


@pytest.fixture(scope="session")
def rt():
    return importlib.import_module("raytracer-rs_amd")


@pytest.fixture(scope="session")
def orc():
    import oracle_py

    return oracle_py


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def gpu_ctx(rt):
    """One device context for the whole GPU session."""
    ctx = rt.Context(0)
    yield ctx
    ctx.close()
