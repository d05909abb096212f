"""CPU-only: `python bench.py --gpus N` starts N ranks by itself (--gpus was parsed and never read).
The dry launch runs the launcher and the control-plane rendezvous without touching a GPU."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus_flag_spawns_that_many_ranks():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout          # rank 0 alone owns stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["local_rank"] == 0


def test_world_size_mismatch_is_refused():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--dry-launch"], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0
    assert "WORLD_SIZE=2" in out.stderr and out.stdout.strip() == ""


def test_a_failing_rank_stops_the_job_instead_of_hanging_it():
    t0 = time.time()
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], env=_env(PLONK_BENCH_DRY_FAIL_RANK="1"),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 3
    assert time.time() - t0 < 90, "the surviving rank was left waiting for its rendezvous time-out"
    assert "rank 1 exited with 3" in out.stderr
