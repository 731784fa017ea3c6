"""pytest configuration: registers the `gpu` marker, puts the host binding and
the oracle on sys.path and makes sure both shared libraries are built."""
import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "audio-matcher_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "audio-matcher_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def gpu_identity():
    """Which physical GPU runs the tests (rocm-smi unique id), for telling a code problem
    from a machine problem when one run differs from the others."""
    import shutil
    import subprocess
    smi = shutil.which("rocm-smi")
    if not smi:
        return "gpu unique id(s): rocm-smi not found"
    try:
        out = subprocess.run([smi, "--showuniqueid"], capture_output=True, text=True, timeout=20).stdout
    except Exception:
        return "gpu unique id(s): rocm-smi failed"
    ids = [ln.split(":")[-1].strip() for ln in out.splitlines() if "Unique ID" in ln and "0x" in ln]
    return "gpu unique id(s): " + (", ".join(ids) if ids else "none visible")


def pytest_report_header(config):
    return gpu_identity()


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def amlib():
    """The HIP library through its ctypes binding (built in-tree if stale)."""
    import build as am_build
    am_build.build_library()
    import audiomatch_amd
    audiomatch_amd.lib()
    return audiomatch_amd


@pytest.fixture(scope="session")
def gpu(amlib):
    if amlib.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need a real MI355X (no CPU fallback exists)")
    amlib.gpu_identity = gpu_identity()   # quoted by tests whose failure may be the machine's
    return amlib
