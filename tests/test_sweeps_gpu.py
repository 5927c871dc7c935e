"""Short runs of the seed sweeps under tools/ (every operator against the CPU oracle over random shapes,
sizes and parameters); the long runs are kept under profiles/."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,seeds", [("fuzz_parity.py", 4), ("fuzz_klt.py", 8), ("fuzz_geometry.py", 4), ("fuzz_ba.py", 6),
                                        ("fuzz_corners.py", 12), ("fuzz_pyramid.py", 12)])
def test_sweep(tool, seeds):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "--seeds", str(seeds)], cwd=ROOT,
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
