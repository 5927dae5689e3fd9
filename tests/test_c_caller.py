"""The C ABI from plain C: tests/c_abi/caller.c includes include/agb_hip.h, links libagbhip.so and runs
hash insert -> kernel map -> agb_spconv_fwd on a 3-voxel input against a brute-force host evaluation."""
import os
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c_abi")


def test_c_caller_builds():
    """gcc compiles the header as C (no C++ constructs, no torch types) and links every symbol the caller uses."""
    r = subprocess.run(["make", "-s", "-B", "caller"], cwd=HERE, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert os.path.exists(os.path.join(HERE, "caller"))


@pytest.mark.gpu
def test_c_caller_runs(device):
    if not os.path.exists(os.path.join(HERE, "caller")):
        subprocess.run(["make", "-s", "caller"], cwd=HERE, check=True)
    r = subprocess.run([os.path.join(HERE, "caller")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "c caller: 9 pairs" in r.stdout     # 3 centres + 3 pairs of neighbours, both directions
    assert "balanced tiles" in r.stdout and "identical sums" in r.stdout
