"""CPU checks of tools that model device code against the oracle (no GPU)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_matrix_core_idct_index_algebra_matches_the_oracle():
    """tools/emulate_idct32_mfma.py: the operand construction, byte planes, constants and register orders of
    tu_idct32_mfma_kernel, run through a numpy model of the MFMA fragment layout, against the oracle's 32x32 transform."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "emulate_idct32_mfma.py")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "matches the oracle" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
