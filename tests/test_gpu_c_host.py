"""The C ABI from a plain C host (tests/c_host/abi_host.c): a process with no Python and no torch in it -- what a MEX
gateway's process is -- links libvo_hip.so, calls the host-pointer entry points with MATLAB's layout and checks them
against the oracle itself.  The binary is built beforehand (make -C tests/c_host; __graft_entry__.build()), this test
only runs it."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HOST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c_host", "abi_host")


def test_c_host_runs_the_abi_without_python():
    if not os.path.exists(HOST):     # built by __graft_entry__.build(); a one-second gcc line if the tree came without it
        subprocess.run(["make", "-C", os.path.dirname(HOST)], capture_output=True, text=True, timeout=60)
    if not os.path.exists(HOST):
        pytest.skip("tests/c_host/abi_host is missing and could not be built here (gcc): python -c 'import __graft_entry__ as g; g.build()'")
    r = subprocess.run([HOST], capture_output=True, text=True, timeout=100)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "C HOST: all checks passed" in r.stdout
    assert r.stdout.count(": ok") == 3
