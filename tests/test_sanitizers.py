"""Sanitizer runs of the HOST side (no GPU sanitizer exists on this pool): the product's C++ host code and
the oracle are rebuilt with -fsanitize=address,undefined and the CPU test-suite is run against those builds; the
threaded host epilogue (tests/cpp/epilogue_threads.cpp) additionally runs under ThreadSanitizer.  Any report makes the
sanitized process exit non-zero (halt_on_error) and fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "needle_amd", "csrc")


def _libasan():
    out = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True)
    path = out.stdout.strip()
    return path if out.returncode == 0 and os.path.isabs(path) and os.path.exists(path) else None


@pytest.fixture(scope="module")
def asan_build():
    if _libasan() is None or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no libasan / hipcc here")
    subprocess.run(["make", "-s", "-j", "8", "-C", CSRC, "all"], check=True)
    subprocess.run(["make", "-s", "-j", "8", "-C", CSRC, "asan"], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True)
    return {"LD_PRELOAD": _libasan(), "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1",
            "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1",
            "NEEDLE_CAPI_LIB": os.path.join(ROOT, "build", "asan", "libneedle_capi.so"),
            "NEEDLE_ORACLE_LIB": os.path.join(ROOT, "oracle", "_asan", "liboracle.so")}


def test_cpu_suite_under_address_and_ub_sanitizers(asan_build):
    """C ABI constructors / errors / file formats / CLI validation / host communicator (real processes) / oracle vs
    golden vectors, all against the instrumented builds."""
    env = dict(os.environ, **asan_build)
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                          os.path.join(ROOT, "tests", "test_capi_cpu.py"), os.path.join(ROOT, "tests", "test_comm_cpu.py"),
                          os.path.join(ROOT, "tests", "test_oracle.py"), "-k", "not gloo"],
                         env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error:" not in out.stderr, out.stderr[-3000:]


def test_threaded_epilogue_under_asan_and_tsan(asan_build):
    exe = os.path.join(ROOT, "build", "asan", "epilogue_threads")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1"))
    assert out.returncode == 0 and "epilogue ok" in out.stdout, out.stdout + out.stderr[-3000:]
    subprocess.run(["make", "-s", "-C", CSRC, "tsan"], check=True)
    out = subprocess.run([os.path.join(ROOT, "build", "tsan", "epilogue_threads")], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert out.returncode == 0 and "epilogue ok" in out.stdout and "ThreadSanitizer" not in out.stderr, \
        out.stdout + out.stderr[-3000:]
