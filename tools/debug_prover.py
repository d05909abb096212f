"""Developer aid (GPU box): dumps the device prover's internal arrays for the reference KAT and compares
them with the oracle's intermediate values.  Uses the reference-shaped 8n quotient domain so that the
arrays line up index by index with the oracle's."""
import os
os.environ["PLONK_QUOTIENT_DOMAIN"] = "8"
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_amd
from oracle import bls12_381 as E, plonk as O
from oracle.rng import StdRng
from oracle.fft import EvaluationDomain
from tests.test_gpu_prover import gpu_prover, wires_of, FixedBlinders
pp = O.srs_setup(1 << 10, StdRng.seed_from_u64(0x9235E700), keep=23)
def circuit():
    c = O.Composer(); w = c.append_witness(7); c.assert_equal_constant(w, 7); return c
op = O.compile_circuit(pp, b"proof-compatibility", circuit())
ctx = plonk_amd.Context(0)
gp = gpu_prover(ctx, op)
n, n8, np_ = op.size, 8*op.size, op.size+8
# key state
names = plonk_amd.POLY_ORDER
for k, name in enumerate(names):
    got = gp.peek(8, k*n8, n8)
    print("evals8", name, got == op.pk.evals8[name])
print("linear", gp.peek(8, 15*n8, n8) == op.pk.evals8["linear"])
d8 = EvaluationDomain(n8)
lag = O.batch_inversion([(x-1) % E.Q for x in op.pk.evals8["linear"]])
ninv = d8.size_inv*8 % E.Q
lag = [li*vh % E.Q*ninv % E.Q for li, vh in zip(lag, op.pk.v_h_coset_8n)]
print("l1", gp.peek(8, 16*n8, n8) == lag)
for k in range(4):
    print("sigma_n", k, gp.peek(9, k*n, n) == op.sigma_evaluations[k])
rec = FixedBlinders(StdRng.seed_from_u64(0x9235E701))
tr = {}
comp = circuit()
exp, _ = O.prove(op, rec, comp, trace=tr)
try:
    got = gp.prove(wires_of(comp, n), {}, rec.drawn)
    print("proof equal", got == exp)
except Exception as e:
    print("prove raised", repr(e))
pad = lambda p, l: list(p) + [0]*(l-len(p))
for k in range(4):
    print("wpoly", k, gp.peek(0, k*np_, np_) == pad(tr["wire_polys"][k], np_))
print("zpoly", gp.peek(1, 0, np_) == pad(tr["z_poly"], np_))
print("perm(scratch)", gp.peek(10, 0, n), tr["perm"])
wp = tr["wire_polys"]
cos_exp = [d8.coset_fft(p) for p in (tr["z_poly"], *wp, tr["pi_poly"])]
for k in range(6):
    print("cos", k, gp.peek(3, k*n8, n8) == cos_exp[k])
t = gp.peek(4, 0, n8)
print("t nonzero len", max([i+1 for i, x in enumerate(t) if x] or [0]))
