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
    return subprocess.run([sys.executable, "-m", "pytest", *target.split(), "-x", "-q", "-m", marker, "-k", select],
                          cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)


# (switches, what plonk_ctx_last_msm must report for a 300-term MSM under them): the child's
# test_basic_plan_is_the_variant_that_was_asked_for compares — a switch the library ignores fails there
K15, K15O, K15L = "nb15::msm_accumulate_kernel", "nb15::msm_accumulate_ordered_kernel", "nb15::msm_accumulate_lds_kernel"
KL, KLO = "nbl::msm_accumulate_kernel", "nbl::msm_accumulate_ordered_kernel"
VARIANTS = [
    ({"PLONK_MSM_ORDER": "1"}, {"ordered_lanes": 1, "accumulate_kernel": K15O}),     # lanes of msm_accumulate in order of slice length
    ({"PLONK_MSM_ACC": "lds"}, {"flags": 4, "accumulate_kernel": K15L}),             # three waves per SIMD, table entries prefetched into LDS
    ({"PLONK_MSM_TAIL": "serial"}, {"flags": 1 | 2, "accumulate_kernel": K15}),      # one lane per addition in the reduction tail
    ({"PLONK_MSM_TABLE": "window"}, {"table_rows": 16, "digit_width": 16, "bucket_bits": 15}),   # 16 window rows, signed 16-bit windows (the default below 2^18 points)
    ({"PLONK_MSM_TABLE": "bitpos"}, {"table_rows": 256, "digit_width": 17, "bucket_bits": 15, "accumulate_kernel": K15}),   # a row per bit position, width-17 NAF digits
    ({"PLONK_MSM_TABLE": "bitpos", "PLONK_MSM_BUCKETS": "19"},                        # width-21 NAF digits over 2^19 buckets (the default above 2^18 terms)
     {"table_rows": 256, "digit_width": 21, "bucket_bits": 19, "slice_entries": 32, "ordered_lanes": 1, "accumulate_kernel": KLO}),
    ({"PLONK_MSM_TABLE": "bitpos", "PLONK_MSM_BUCKETS": "19", "PLONK_MSM_ORDER": "0"},
     {"table_rows": 256, "bucket_bits": 19, "ordered_lanes": 0, "accumulate_kernel": KL}),
    ({"PLONK_MSM_TABLE": "bitpos", "PLONK_MSM_BUCKETS": "19", "PLONK_MSM_SORT13": "1"},   # round 5: two half-size partition workgroups per CU (13 digit slots)
     {"table_rows": 256, "bucket_bits": 19, "flags": 8 | 2, "accumulate_kernel": KLO}),
    ({"PLONK_MSM_TABLE": "halfpos", "PLONK_MSM_BUCKETS": "19", "PLONK_MSM_SORT13": "1"},
     {"table_rows": 128, "bucket_bits": 19, "flags": 8 | 2, "accumulate_kernel": KLO}),
    ({"PLONK_MSM_BSUM": "lane"}, {"flags": 2, "accumulate_kernel": K15}),            # one lane per bucket in msm_bucket_sum instead of a quad (small MSMs)
    ({"PLONK_MSM_TABLE": "halfpos"}, {"table_rows": 128, "digit_width": 16, "bucket_bits": 15}),   # round 4: a row for every second bit position, even-position digits
    ({"PLONK_MSM_TABLE": "halfpos", "PLONK_MSM_BUCKETS": "19"},                       # ... width-20 digits over 2^19 buckets (keys whose 256 rows do not fit)
     {"table_rows": 128, "digit_width": 20, "bucket_bits": 19, "accumulate_kernel": KLO}),
]


# (round 6: + the host helper threads of fetch_commitments off and at their maximum — the same proofs either way)
PROVER_VARIANTS = [{"PLONK_MSM_TABLE": "bitpos", "PLONK_MSM_BUCKETS": "19"}, {"PLONK_MSM_TABLE": "bitpos", "PLONK_MSM_BUCKETS": "15"},
                   {"PLONK_HOST_THREADS": "0"}, {"PLONK_HOST_THREADS": "7"}, {"PLONK_Z_COMMIT": "coeff"},
                   {"PLONK_WIRE_POLYS_SIDE": "0", "PLONK_SIDE_DEFER": "0"},   # the schedule of circuits above 2^19 gates: wire transforms in front of their commitments, side transforms with them
                   {"PLONK_MSM_TABLE": "halfpos", "PLONK_MSM_BUCKETS": "19"},
                   {"PLONK_MSM_TABLE": "bitpos", "PLONK_MSM_BUCKETS": "19", "PLONK_MSM_SORT13": "1"}]
PROVER_SELECT = "deterministic_v3 or random_arithmetic or host_time_slots or (proof_bytes_equal_c_oracle and not 16 and not 2p20)"
# round 6: the phased launches of msm_batch_device (host wire columns committed as they arrive) forced at small sizes, over both
# bucket-count variants; the child's test_host_wire_schedule_is_the_one_asked_for compares plonk_prover_describe
SCHEDULE_VARIANTS = [({"PLONK_WIRE_BY_COLUMN": "1"}, 3), ({"PLONK_WIRE_BY_COLUMN": "2"}, 4),
                     ({"PLONK_WIRE_BY_COLUMN": "1", "PLONK_MSM_TABLE": "bitpos", "PLONK_MSM_BUCKETS": "19"}, 3),
                     ({"PLONK_WIRE_BY_COLUMN": "0"}, 1)]
SCHEDULE_SELECT = "host_wire_schedule or deterministic_v3 or random_arithmetic"   # the latter prove from host columns too (gp.prove)


def _key(env):
    return ",".join(f"{k}={x}" for k, x in sorted(env.items()))


@pytest.fixture(scope="module")
def children():
    """Every child of this module is started when the first test asks for one, four at a time (round 6: the 16 children used
    to run one after the other, ~80 s of a GPU session in which the device and most host cores sat idle — a child is
    interpreter start-up, the Python KAT set-up and a few small proofs).  Each test still owns exactly one child and fails on
    that child's output."""
    import concurrent.futures as cf
    import json
    pool = cf.ThreadPoolExecutor(max_workers=4)
    futs = {}
    for variant in PROVER_VARIANTS:      # the long ones first
        futs["prover:" + _key(variant)] = pool.submit(run_variant, variant, PROVER_SELECT, "gpu and not slow",
                                                      "tests/test_gpu_prover.py tests/test_gpu_prove_sizes.py")   # one child, both files
    for variant, launches in SCHEDULE_VARIANTS:
        futs["schedule:" + _key(variant)] = pool.submit(run_variant, dict(variant, PLONK_TEST_EXPECT_WIRE_LAUNCHES=str(launches)), SCHEDULE_SELECT,
                                                        "gpu and not slow", "tests/test_gpu_prover.py")
    for variant, plan in VARIANTS:
        futs["edge:" + _key(variant)] = pool.submit(run_variant, dict(variant, PLONK_TEST_EXPECT_PLAN=json.dumps(plan)))
    yield futs
    pool.shutdown(wait=False, cancel_futures=True)


@pytest.mark.gpu
@pytest.mark.parametrize("variant,plan", VARIANTS, ids=[",".join(f"{k}={x}" for k, x in v.items()) for v, _ in VARIANTS])
def test_variant_matches_the_oracle_on_the_edge_cases(children, variant, plan):
    r = children["edge:" + _key(variant)].result()
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("variant", PROVER_VARIANTS, ids=lambda v: ",".join(f"{k}={x}" for k, x in v.items()))
def test_prover_parity_holds_with_every_table_and_bucket_layout(children, variant):
    """whole proofs (reference KAT digest, random circuits, widget circuits vs the C oracle at 2^12 / 2^13) with the table /
    bucket layouts that the size rules would only pick for large circuits"""
    r = children["prover:" + _key(variant)].result()
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("variant,launches", SCHEDULE_VARIANTS, ids=[",".join(f"{k}={x}" for k, x in v.items()) for v, _ in SCHEDULE_VARIANTS])
def test_host_wire_columns_commit_by_phase_with_the_same_bytes(children, variant, launches):
    """msm_batch_device's phases (bucket sort + accumulation + bucket sums per column or column pair, one reduction tail per
    group) against the grouped launch and the C oracle: 2^12-gate widget circuit (heavy buckets in every column), the KAT and
    random arithmetic circuits, proved from HOST wire columns under each forced schedule"""
    r = children["schedule:" + _key(variant)].result()
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout and "skipped" not in r.stdout.splitlines()[-1]


def test_child_process_harness_selects_and_runs_tests():
    """CPU-side check of the harness itself (same command line, a CPU test file)."""
    r = run_variant({"PLONK_MSM_ORDER": "1"}, select="digits or oracle_msm", marker="not gpu", target="tests/test_msm_wide_model.py")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
