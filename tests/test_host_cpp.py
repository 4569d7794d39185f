"""The C++ host mirror of the reference's operator surface (databend_amd/host/dbhip_host.hpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "databend_amd", "host")


def test_host_mirror_builds_and_links_against_the_c_abi():
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(HOST, "host_selftest"))
    out = subprocess.run(["ldd", os.path.join(HOST, "host_selftest")], capture_output=True, text=True).stdout
    assert "libdbhip.so" in out


@pytest.mark.gpu
def test_host_selftest_passes_on_gpu():
    """Evaluator / FunctionRegistry / TransformFilter / CompoundBlockOperator / TransformPartialAggregate ->
    TransformFinalAggregate / InnerHashJoin / sort driven from C++ against plain host loops."""
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(HOST, "host_selftest")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout
