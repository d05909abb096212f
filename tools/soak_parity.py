"""GPU box: randomized differential soak — N random circuits (every widget family, public inputs, sizes 2^9..2^13,
random blinders, alternating quotient domains, occasionally corrupted witnesses) proved by the HIP prover and by the
C restatement of the reference's prove() — half of them built by plonk_compile from gate columns and proved from the
witness table, half from coefficient forms and wire columns; every pair of 1008-byte proofs must be identical and every corrupted
witness must be CircuitUnsatisfied on both sides.

    python tools/soak_parity.py [N] [seed]
"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import plonk_amd  # noqa: E402
from oracle import cbind  # noqa: E402
from tests import circuits as C  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed0)
srs_cache = {}
ok = unsat = 0
t0 = time.time()
for it in range(N):
    log_n = rnd.choice((9, 10, 11, 12, 12, 13))
    n = 1 << log_n
    domain8 = bool(it & 1)
    ngates = rnd.randrange(n // 2 + 9, n + 1)          # constraints need not be a power of two
    comp = C.big_widget_circuit(ngates, seed=rnd.getrandbits(32))()
    case = C.compile_fast(comp, b"soak-%d" % it)
    from_circuit = it % 4 >= 2     # half of the circuits go through plonk_compile + plonk_prover_prove_witnesses
    if log_n not in srs_cache:
        srs_cache[log_n] = C.synthetic_srs(n + 7)
    srs = srs_cache[log_n]
    ctx = plonk_amd.Context(0, plonk_amd.GpuConfig(quotient_domain=8 if domain8 else 4))   # plonk_gpu_config, not the environment
    ctx.srs_load_bytes(srs, n + 7)
    cp = cbind.CProver(case["constraints"], case["label"], case["polys"], srs)
    cols = C.circuit_columns(comp) if from_circuit else None
    if from_circuit:   # Compiler::preprocess on the device from the gate columns
        gp = plonk_amd.Prover.compile(ctx, case["label"], cols["selectors"], cols["wires"], cols["witnesses"])
    else:
        gp = plonk_amd.Prover(ctx, case["constraints"], case["label"], case["polys"], None)
    assert gp.vk_commitments() == cp.vk(), ("vk", it)
    assert gp.describe()["quotient_domain"] == (8 if domain8 else 4), ("quotient domain not honoured", it)
    bl = C.blinders(rnd.getrandbits(32))
    wires = list(case["wires"])
    corrupt = rnd.random() < 0.15
    values = cols["values"] if from_circuit else None
    if corrupt and from_circuit:   # one witness value changes: every wire it sits on changes with it
        w = rnd.randrange(cols["witnesses"])
        v = bytearray(values)
        v[32 * w] ^= 1 << rnd.randrange(8)
        values = bytes(v)
        pad = bytes(32 * (case["size"] - case["constraints"]))
        wires = [b"".join(values[32 * i:32 * i + 32] for i in cols["wires"][col]) + pad for col in range(4)]
    elif corrupt:
        col, row = rnd.randrange(4), rnd.randrange(case["constraints"])
        w = bytearray(wires[col])
        w[32 * row] ^= 1 << rnd.randrange(8)
        wires[col] = bytes(w)
    wbuf = ctx.alloc(4 * 32 * case["size"])
    for k in range(4):
        wbuf.upload(wires[k], 32 * case["size"] * k)
    try:
        got = gp.prove_witnesses(values, case["pi"], bl) if from_circuit else gp.prove_dev(wbuf.ptr, case["pi"], bl)
        g_unsat = False
    except plonk_amd.CircuitUnsatisfied:
        g_unsat = True
    try:
        want = cp.prove(wires, case["pi_idx"], case["pi_val"], bl)
        c_unsat = False
    except cbind.CircuitUnsatisfied:
        c_unsat = True
    assert g_unsat == c_unsat, ("unsat disagreement", it, g_unsat, c_unsat, corrupt)
    if g_unsat:
        unsat += 1
    else:
        assert got == want, ("proof bytes differ", it, log_n, ngates, domain8)
        ok += 1
    wbuf.free()
    gp.close()
    cp.close()
    ctx.close()
print(f"soak_parity: {ok} identical proofs, {unsat} unsatisfied witnesses rejected by both, {N} circuits, {time.time() - t0:.0f} s, seed {seed0}")
