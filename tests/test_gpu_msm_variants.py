"""The opt-in MSM kernel variants (A/B switches read once per process) must compute the same group
elements as the default path: the edge-case MSM tests of test_gpu_msm.py are re-run in a child process per variant."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELECT = "basic or edge or skew or small_scalars or doubling"


def run_variant(env_extra, select=SELECT, marker="gpu", target="tests/test_gpu_msm.py"):
    env = dict(os.environ, **env_extra)
    return subprocess.run([sys.executable, "-m", "pytest", target, "-x", "-q", "-m", marker, "-k", select],
                          cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [
    {"PLONK_MSM_ORDER": "1"},      # lanes of msm_accumulate in order of slice length
    {"PLONK_MSM_ACC": "lds"},      # three waves per SIMD, table entries prefetched into LDS
    {"PLONK_MSM_TAIL": "serial"},  # one lane per addition in the reduction tail
], ids=lambda v: ",".join(f"{k}={x}" for k, x in v.items()))
def test_variant_matches_the_oracle_on_the_edge_cases(variant):
    r = run_variant(variant)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_child_process_harness_selects_and_runs_tests():
    """CPU-side check of the harness itself (same command line, a CPU test file)."""
    r = run_variant({"PLONK_MSM_ORDER": "1"}, select="digits or oracle_msm", marker="not gpu", target="tests/test_msm_wide_model.py")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
