"""Setup cost of the Lagrange-basis key (plonk_lagrange_key: the inverse FFT over the group, msm.hip ecfft_*): wall time per
call, D2H of the (n + 2) x 96 bytes included.  usage: python tools/lagrange_key_time.py [log_gates ...] -> one JSON line per size"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import plonk_amd  # noqa: E402


def run(log_n):
    n = 1 << log_n
    ctx = plonk_amd.Context(0)
    pts = ctx.alloc(96 * (n + 8))
    ctx.srs_generate_dev(bench.TAU, bench.G_SCALAR, n + 8, pts.ptr)
    ctx.srs_load_dev(pts.ptr, n + 8)
    pts.free()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        key = ctx.lagrange_key(log_n)
        ts.append(time.perf_counter() - t0)
    out = {"log_gates": log_n, "lib": os.path.basename(os.environ.get("PLONK_HIP_LIB", "default")), "lagrange_key_s": [round(t, 3) for t in ts],
           "key_blake2b": hashlib.blake2b(key).hexdigest()[:32]}
    ctx.close()
    return out


if __name__ == "__main__":
    for lg in [int(x) for x in sys.argv[1:]] or [16, 20]:
        print(json.dumps(run(lg)), flush=True)
