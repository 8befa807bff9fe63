"""Host arithmetic of the chirp-z and residue-pair plans (flan_amd/csrc/bs_plan.h) over every even dft size up to 2^20: tests/cpp/plans_test.cpp, CPU only."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plans_over_every_even_size():
    src = os.path.join(ROOT, "tests", "cpp", "plans_test.cpp")
    binary = os.path.join(ROOT, "tests", "cpp", "plans_test")
    subprocess.run(["g++", "-O2", "-std=c++17", src, "-o", binary], check=True)
    r = subprocess.run([binary], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "PASSED" in r.stdout, r.stdout + r.stderr
