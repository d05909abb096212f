"""CPU: the residue-class / coefficient-range algebra of the multi-GPU prover against the oracle
(tests/shard_model.py mirrors prover.hip's prover_prove_sharded and the shard_* kernels of poly.hip):
in-process for world 2, 4 and 8, and as real world-2 / world-4 gloo jobs whose ranks exchange the
messages through the same all-gather-based transport the library's host callback uses."""
import os
import random
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.bls12_381 import GENERATOR, Q
from oracle.fft import EvaluationDomain
from oracle.plonk import poly_eval, poly_ruffini
from tests import shard_model as M


def make_t(n, Qc, seed):
    """a quotient-shaped polynomial: 4n + 7 coefficients (zeros above), and its evaluations on the Q n coset"""
    r = random.Random(seed)
    t = [r.randrange(Q) for _ in range(4 * n + 7)]
    d = EvaluationDomain(Qc * n)
    if Qc == 4:   # 4n + 7 coefficients on a 4n coset: the values are those of t mod (X^4n - g^4n) (coset_fft would truncate)
        g4n = pow(GENERATOR, 4 * n, Q)
        return t, d.coset_fft([(t[k] + (g4n * t[4 * n + k] if k < 7 else 0)) % Q for k in range(4 * n)])
    return t, d.coset_fft(t + [0] * (Qc * n - len(t)))


def rank_messages(lay, t_full_evals, rank):
    F = {}
    for j in lay.classes(rank):
        on_class = [t_full_evals[j + lay.Q * k] for k in range(lay.n)]     # class j = full index j + Q k
        F[j] = M.remainder(lay, on_class, j)
    return M.pack(lay, F, rank)


def check_rank(lay, rank, recv, t):
    n = lay.n
    low7 = t[:7] if lay.Q == 4 else None
    got = M.combine(lay, recv, rank, low7)
    lo, hi = lay.rng(rank)
    want = {}
    for idx in range(lo, min(hi, n)):
        for i1 in range(4):
            want[(i1, idx)] = t[i1 * n + idx]
    for k in range(7):
        if lo <= n + k < hi:
            want[(3, n + k)] = t[4 * n + k]
    assert got == want, (rank, len(got), len(want))


@pytest.mark.parametrize("world,n", [(2, 64), (4, 64), (8, 64), (4, 128)])
def test_class_decomposition_recovers_every_coefficient_range(world, n):
    lay = M.Layout(n, world)
    t, evals = make_t(n, lay.Q, 11 + world)
    sends = [rank_messages(lay, evals, r) for r in range(world)]
    for r in range(world):
        recv = [sends[src][r] for src in range(world)]          # all-to-all
        check_rank(lay, r, recv, t)


def test_fold_equals_the_full_coset_transform_on_the_class():
    n, world = 64, 4
    lay = M.Layout(n, world)
    r = random.Random(5)
    for extra in (2, 3):
        poly = [r.randrange(Q) for _ in range(n + extra)]
        full = EvaluationDomain(lay.Q * n).coset_fft(poly)
        for j in range(lay.Q):
            assert M.class_evals(lay, poly, j) == [full[j + lay.Q * k] for k in range(n)]
        # x^n is the constant sigma_j on a class, and the rotation X -> w_n X is the next point of the class
        assert pow(lay.shift(1) * pow(M.omega(6), 5, Q) % Q, n, Q) == lay.sigma(1)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_range_sharded_evaluation_and_ruffini(world):
    n = 64
    lay = M.Layout(n, world)
    r = random.Random(21)
    poly = [r.randrange(Q) for _ in range(n + 7)]
    z = r.randrange(1, Q)
    parts, totals, S = [], [], []
    for rank in range(world):
        lo, hi = lay.rng(rank)
        parts.append(M.eval_partial(poly, lo, hi, z))
        S.append(M.ruffini_local(poly, lo, hi, z))
        totals.append(S[-1][0])
    assert sum(pow(z, lay.rng(k)[0], Q) * parts[k] for k in range(world)) % Q == poly_eval(poly, z)
    assert sum(totals) % Q == poly_eval(poly, z)             # the identity-check value of prover.hip
    want = poly_ruffini(poly, z)
    want = want + [0] * (n + 7 - len(want))                  # the dropped remainder slot
    got = {}
    for rank in range(world):
        lo, hi = lay.rng(rank)
        got.update(M.ruffini_finish(S[rank], lo, hi, z, sum(totals[rank + 1:]) % Q))
    assert [got[i] for i in range(n + 7)] == want


@pytest.mark.parametrize("world", [2, 4, 8])
def test_range_sharded_grand_product(world):
    """round 2 of a sharded proof (round 4 of the build): every rank forms the terms of its n / W evaluation indices, scans them
    locally, the range products are exchanged, and the scaled slices concatenate to the oracle's z evaluations
    (permutation.rs:213-294) — prover.hip prover_prove_sharded, PLONK_SHARD_Z"""
    from oracle import plonk as O
    from oracle.fft import EvaluationDomain
    n = 64
    r = random.Random(33)
    wires = [[r.randrange(Q) for _ in range(n)] for _ in range(4)]
    sigma_ev = [[r.randrange(Q) for _ in range(n)] for _ in range(4)]
    beta, gamma = r.randrange(Q), r.randrange(Q)
    want = O.permutation_vec(EvaluationDomain(n), wires, beta, gamma, sigma_ev)
    cnt = n // world
    local, totals = [], []
    for rank in range(world):
        loc, tot = M.grand_product_rank(M.grand_product_terms(n, wires, sigma_ev, beta, gamma, rank * cnt, cnt))
        local.append(loc)
        totals.append(tot)
    got = []
    for rank in range(world):
        got += M.grand_product_finish(local[rank], totals, rank)
    assert got == want


# ---- the same exchange as a real gloo job: all-to-all realised with all-gather, like the library callback
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lay = M.Layout(n, world)
    t, evals = make_t(n, lay.Q, 99)
    send = rank_messages(lay, evals, rank)                    # [peer][k][per + 8]
    flat = b"".join(v.to_bytes(32, "little") for peer in send for row in peer for v in row)
    mine = torch.frombuffer(bytearray(flat), dtype=torch.uint8)
    out = torch.empty(world * mine.numel(), dtype=torch.uint8)
    dist.all_gather_into_tensor(out, mine)
    raw = out.numpy().tobytes()
    per_peer = lay.cpr * (lay.per + 8) * 32
    recv = []
    for src in range(world):
        blob = raw[src * len(flat) + rank * per_peer: src * len(flat) + (rank + 1) * per_peer]
        vals = [int.from_bytes(blob[i:i + 32], "little") for i in range(0, len(blob), 32)]
        recv.append([vals[k * (lay.per + 8):(k + 1) * (lay.per + 8)] for k in range(lay.cpr)])
    try:
        check_rank(lay, rank, recv, t)
        q.put((rank, "ok"))
    except AssertionError as e:   # noqa: BLE001
        q.put((rank, repr(e)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_class_exchange_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 64, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, "ok") for r in range(world)]
