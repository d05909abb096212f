"""CPU, world_size 2, gloo: the multi-GPU MSM protocol of prover.hip / bench.py —
contiguous SRS point ranges per rank, all-gather of the partial sums, local EC add on every
rank.  The partial MSMs are computed with the oracle here; the GPU ranks compute
the same partial sums with msm_device on their slice."""
import os
import random
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import plonk_amd
from oracle import bls12_381 as E

Q = E.Q


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, m, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r = random.Random(seed)
    pts = [E.g1_mul(E.G1_GEN, r.randrange(1, Q)) for _ in range(total)]
    sc = [r.randrange(Q) for _ in range(m)]
    lo, hi = plonk_amd.shard_range(total, rank, world)
    hi = min(hi, m)
    part = E.msm_naive(pts[lo:hi], sc[lo:hi]) if hi > lo else None
    # fixed-size payload like the 192-byte XYZZ partial: 96 B affine + flag, padded
    payload = (E.g1_to_raw96(part) + b"\0") if part is not None else bytes(96) + b"\1"
    send = torch.frombuffer(bytearray(payload.ljust(192, b"\0")), dtype=torch.uint8)
    out = torch.empty(world * 192, dtype=torch.uint8)
    dist.all_gather_into_tensor(out, send)
    raw = out.numpy().tobytes()
    acc = None
    for k in range(world):
        chunk = raw[192 * k:192 * k + 97]
        if chunk[96] == 0:
            acc = E.g1_add(acc, E.g1_from_raw96(chunk[:96]))
    q.put((rank, E.g1_compress(acc), E.g1_compress(E.msm_naive(pts, sc))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total,m", [(23, 14), (40, 40), (9, 3)])
def test_point_range_sharded_msm_world2(total, m):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, m, 99, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, got, full in res:
        assert got == full              # every rank reconstructs the full commitment
    assert res[0][1] == res[1][1]       # and all ranks agree (transcripts stay in lock-step)


def test_shard_ranges_partition_exactly():
    for total in (1, 7, 23, 1 << 10, (1 << 20) + 7):
        for world in (1, 2, 3, 4, 8):
            spans = [plonk_amd.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a0 <= a1


def test_allgather_callback_marshalling():
    """The ctypes trampoline bench.py hands to plonk_prover_create."""
    import ctypes
    world = 3

    def gather(send: bytes) -> bytes:
        return b"".join(bytes([k]) + send[1:] for k in range(world))
    seen = {}

    def _cb(user, send, recv, nbytes):
        out = gather(ctypes.string_at(send, nbytes))
        ctypes.memmove(recv, out, len(out))
        seen["n"] = nbytes
        return 0
    fn = plonk_amd.ALLGATHER_FN(_cb)
    src = ctypes.create_string_buffer(b"\xaa" * 192, 192)
    dst = ctypes.create_string_buffer(192 * world)
    assert fn(None, ctypes.cast(src, ctypes.c_void_p), ctypes.cast(dst, ctypes.c_void_p), 192) == 0
    assert seen["n"] == 192 and dst.raw[0] == 0 and dst.raw[192] == 1 and dst.raw[384] == 2


def _column_worker(rank, world, port, n, seed, q):
    """prover.hip lag_whole (round 5): the four wire commitments of round 1 split BY COLUMN over 2 / 4 ranks — rank r owns
    columns [r * 4 / W, (r + 1) * 4 / W), commits to them over the WHOLE (n + 2)-point Lagrange-basis key (wire values + the
    column's two blinders) and contributes the identity for the others; the all-gather + add of fetch_commitments is the one
    the split by point range uses."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r = random.Random(seed)
    key = [E.g1_mul(E.G1_GEN, r.randrange(1, Q)) for _ in range(n + 2)]          # [L_i(tau)] G, [tau^n] G - G, [tau^(n+1)] G - [tau] G
    cols = [[r.randrange(Q) for _ in range(n)] + [r.randrange(Q), r.randrange(Q)] for _ in range(4)]   # values ++ (b0, b1)
    per = 4 // world
    mine = range(per * rank, per * (rank + 1))
    payload = b""
    for k in range(4):
        part = E.msm_naive(key, cols[k]) if k in mine else None
        payload += ((E.g1_to_raw96(part) + b"\0") if part is not None else bytes(96) + b"\1").ljust(192, b"\0")
    send = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
    out = torch.empty(world * len(payload), dtype=torch.uint8)
    dist.all_gather_into_tensor(out, send)
    raw = out.numpy().tobytes()
    got = []
    for k in range(4):
        acc = None
        for src in range(world):
            chunk = raw[len(payload) * src + 192 * k:len(payload) * src + 192 * k + 97]
            if chunk[96] == 0:
                acc = E.g1_add(acc, E.g1_from_raw96(chunk[:96]))
        got.append(E.g1_compress(acc))
    # the same four commitments by point range (round 4's split): identical group elements
    lo, hi = plonk_amd.shard_range(n + 2, rank, world)
    by_range = [E.msm_naive(key[lo:hi], c[lo:hi]) if hi > lo else None for c in cols]
    q.put((rank, got, [E.g1_compress(E.msm_naive(key, c)) for c in cols], [E.g1_to_raw96(p) if p is not None else None for p in by_range]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_wire_group_split_by_column(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n = 8
    procs = [ctx.Process(target=_column_worker, args=(r, world, port, n, 7, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, got, full, _ in res:
        assert got == full                                   # every rank holds the four full commitments
    # and the per-rank partial sums of the split by point range add up to the same commitments
    for k in range(4):
        acc = None
        for _, _, _, parts in res:
            if parts[k] is not None:
                acc = E.g1_add(acc, E.g1_from_raw96(parts[k]))
        assert E.g1_compress(acc) == res[0][2][k]
