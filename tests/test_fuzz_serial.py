"""The byte parsers that read untrusted input (plonk_prover_blob_check / plonk_public_parameters_check and the first step of
plonk_prover_from_bytes / plonk_srs_load_public_parameters) under libFuzzer + AddressSanitizer + UBSan on the host — the
hardening VERDICT r5 asked for where row f4 cannot be pinned to reference-produced files.  tests/fuzz/fuzz_serial.cpp compiles
exactly the product's checking code (plonk_amd/csrc/serial_check.hpp).  Here: every committed seed (the KAT blob, the three
PublicParameters encodings, the malformed cases of the reference's decoder tests) must run clean, and a short campaign must
find nothing; the long campaigns are tests/fuzz/run.sh (profiles/r06/fuzz_serial.txt: 30 CPU-minutes, 478 k executions)."""
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FUZZ = os.path.join(ROOT, "tests", "fuzz")
CLANG = os.environ.get("CXX") or "/opt/rocm/lib/llvm/bin/clang++"

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="no clang++ with the libFuzzer / sanitizer runtimes")


def test_seed_corpus_is_what_make_corpus_writes():
    names = sorted(os.listdir(os.path.join(FUZZ, "corpus")))
    assert "blob_kat" in names and "pp_raw" in names and "pp_compressed" in names and len(names) >= 35
    kat = open(os.path.join(FUZZ, "corpus", "blob_kat"), "rb").read()
    assert kat[:3] == b"\0\0\0" and int.from_bytes(kat[3 + 32:3 + 40], "big") == 8     # parser 0, no trim; size = 8 (the KAT circuit)


def test_parsers_survive_the_seeds_and_a_short_campaign():
    with tempfile.TemporaryDirectory() as work:
        r = subprocess.run([os.path.join(FUZZ, "run.sh"), "20", "1", work], capture_output=True, text=True, timeout=600)
        tail = (r.stdout + r.stderr)[-3000:]
        assert r.returncode == 0, tail
        assert "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail and "ERROR: libFuzzer" not in tail, tail
        assert "stat::number_of_executed_units" in tail, tail
        assert not [f for f in os.listdir(work) if f.startswith(("crash-", "leak-", "timeout-", "oom-"))]
    assert shutil.which("bash")
