"""The exact 128 / 64 division of dev_common.h (udiv_2by1 / udiv128_by_64: two double-precision estimates with exact 128-bit remainders,
round 5) against the host compiler's 128-bit `/`: the two functions are cut out of the product header and compiled with
tests/div_host_check.cpp."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_udiv128_by_64_is_exact(tmp_path):
    src = open(os.path.join(ROOT, "databend_amd", "csrc", "dev_common.h")).read()
    a = src.index("__host__ __device__ inline uint64_t udiv_2by1(")
    b = src.index("// q = a / d (d != 0); *rem gets the remainder.")
    (tmp_path / "div_under_test.h").write_text(src[a:b])
    exe = str(tmp_path / "div_host_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", str(tmp_path), "-o", exe, os.path.join(ROOT, "tests", "div_host_check.cpp")])
    r = subprocess.run([exe, "20"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and re.match(r"ok \d+ divisions", r.stdout), r.stdout[-500:]
