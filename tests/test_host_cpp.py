"""The C++ drop-in surface (include/flan/*.h over libflan_host.so + libflanhip.so), driven by tests/cpp/host_test.cpp."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "host_test")


def _build():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "flan_amd", "host")], check=True)
    src = os.path.join(ROOT, "tests", "cpp", "host_test.cpp")
    deps = [src, os.path.join(ROOT, "flan_amd", "libflan_host.so")]
    if not os.path.exists(BIN) or any(os.path.getmtime(d) > os.path.getmtime(BIN) for d in deps):
        subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), src, "-o", BIN,
                        "-L" + os.path.join(ROOT, "flan_amd"), "-lflan_host", "-lflanhip",
                        "-Wl,-rpath," + os.path.join(ROOT, "flan_amd"), "-lpthread"], check=True)


def _run(args):
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "flan_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    return subprocess.run([BIN] + args, capture_output=True, text=True, env=env, timeout=300)


def test_host_surface_without_device():
    """no GPU: null objects and a loud error, never a CPU fallback"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here; the no-device behaviour is checked in the CPU container")
    _build()
    r = _run(["--no-device"])
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASSED" in r.stdout


@pytest.mark.gpu
def test_host_surface_on_device():
    _build()
    r = _run([])
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASSED" in r.stdout
