"""one short line per bench run: python tools/archive_r5/quick.py <tag> [bench.py args...] (environment selects the variant)"""
import json
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag, args = sys.argv[1], sys.argv[2:]
bench = os.environ.get("QUICK_BENCH", os.path.join(ROOT, "bench.py"))
r = subprocess.run([sys.executable, bench, "--no-extras", "--no-cpu-baseline", *args], capture_output=True, text=True, cwd=os.path.dirname(bench))
try:
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print(json.dumps({"tag": tag, "ms": j["value"], "kernels": j["kernel_ms_per_prove"], "digest": j["proof_blake2b"][:12]}), flush=True)
except Exception as e:   # noqa: BLE001
    print(json.dumps({"tag": tag, "error": repr(e), "stderr": r.stderr[-400:]}), flush=True)
