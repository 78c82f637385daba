"""Parity at BASELINE's full sizes against the WHOLE oracle (scripts/parity_full_size.py, a process of its own: the
oracle on 16 host threads takes about a second): all 50 000 tracks of a KITTI-shaped pair within 0.01 px with no
validity flip away from the threshold, and 2000 hypotheses x 50 000 points bit-equal under identical hypotheses."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_full_size_parity_against_the_whole_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "parity_full_size.py")], cwd=ROOT, capture_output=True,
                       text=True, timeout=110)
    print(r.stdout, r.stderr[-2000:])
    assert r.returncode == 0 and "FULL-SIZE PARITY: ok" in r.stdout, r.stdout + r.stderr[-2000:]
    assert r.stdout.count("bit-equal") == 2
