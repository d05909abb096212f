"""GPU box: prove at 2^22 gates (BASELINE configs[4] size) and check the proof with the known-tau
verifier (oracle/verifier.py).  Kept out of the pytest suite because of its 15 s Python-side setup."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, plonk_amd, time
from oracle import bls12_381 as E
from oracle.verifier import verify_with_tau
Q = E.Q
log_n = 22
ctx = plonk_amd.Context(0)
t = time.time()
prover, wbuf, srs_total = bench.build_prover(ctx, log_n, 0, 1, None)
print("setup", time.time() - t)
tau, g = 0x5EED0000 * 0x9E3779B97F4A7C15 % Q, 0xA5A5A5A5DEADBEEF
srs_g = E.g1_mul(E.G1_GEN, g)
raw = prover.vk_commitments()
vk = {name: E.g1_decompress(raw[48 * i:48 * i + 48]) for i, name in enumerate(plonk_amd.POLY_ORDER)}
bl = plonk_amd.fr_to_bytes_mont([(0xB11D0000 + i) * 0x9E3779B97F4A7C15 % Q for i in range(14)])
t = time.time(); p1 = prover.prove_dev(wbuf.ptr, {}, bl); print("prove", time.time() - t)
t = time.time(); p2 = prover.prove_dev(wbuf.ptr, {}, bl); print("prove", time.time() - t)
assert p1 == p2
print("verifies:", verify_with_tau(p1, vk, b"bench", 1 << log_n, {}, tau, srs_g))
