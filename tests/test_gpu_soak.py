"""GPU: a short run of the randomized differential soak (tools/soak_parity.py): random circuits with every widget
family and public inputs, random sizes 2^9..2^13 with constraint counts that are not powers of two, random
blinders, alternating quotient domains, some witnesses corrupted — HIP prover and C restatement of the reference
must agree on every proof byte and on every CircuitUnsatisfied.  (profiles/r02c + r02d soak_parity.txt: 1860 circuits over round 2, profiles/r05/soak_parity.txt: 1050 on the round-5 library, no disagreement.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_randomized_parity_soak_short():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_parity.py"), "16", "77"], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert "16 circuits" in out.stdout and "soak_parity:" in out.stdout
