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
