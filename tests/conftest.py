"""pytest configuration: `gpu` marker, session-wide builds of the checker (oracle), the CPU emulation
fixture and — when hipcc is present — the product library."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _newer(src_files, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_files if os.path.exists(s))


@pytest.fixture(scope="session", autouse=True)
def built():
    """Builds what is missing or stale.  On the GPU box the .so files travel with the snapshot."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    csrc = os.path.join(ROOT, "needle_amd", "csrc")
    if os.path.exists(hipcc):
        subprocess.run(["make", "-s", "-j", "8", "-C", csrc], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    emu_src = os.path.join(ROOT, "tests", "cpu_emu", "emu.cpp")
    emu_so = os.path.join(ROOT, "tests", "cpu_emu", "libemu.so")
    if _newer([emu_src, os.path.join(csrc, "fp_core.h")], emu_so):
        # -mfma: the explicit fma()/fmaf() calls of fp_core.h become instructions (a software fmaf is ~50x slower);
        # -ffp-contract=off: nothing else is fused -- the same arithmetic as the device code
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-o", emu_so,
                        emu_src], check=True)
    return True


@pytest.fixture(scope="session")
def has_gpu():
    from needle_amd import capi
    return capi.device_count() > 0
