"""The ping-pong kernels keep DMA data in flight across barriers with counted waits: a missing wait shows up as a rare wrong
tile that a single parity run can miss.  They are deterministic, so the screen is repetition: same inputs, bit-identical
outputs, with an unrelated kernel interleaved to move the timing (scripts/dbg/race_screen.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_pingpong_kernels_are_run_to_run_identical():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dbg", "race_screen.py"), "25"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "race screen: clean" in r.stdout
