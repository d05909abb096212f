"""ctypes binding of oracle/c/oracle.c (CPU restatement in plain C; test infrastructure).

Builds a -march=native copy keyed by the host CPU's flag set, so the same tree works on
the build container and on the GPU box's host CPU."""
from __future__ import annotations

import ctypes
import hashlib
import os
import subprocess

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c")
_lib = None


def _cpu_tag() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            flags = next((l for l in f if l.startswith("flags")), "")
    except OSError:
        flags = ""
    return hashlib.sha1(flags.encode()).hexdigest()[:10]


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    src = os.path.join(_DIR, "oracle.c")
    so = os.path.join(_DIR, f"liboracle_{_cpu_tag()}.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", src, "-o", so])
    lib = ctypes.CDLL(so)
    lib.oracle_ntt.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_int]
    lib.oracle_msm.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int]
    lib.oracle_max_threads.restype = ctypes.c_int
    _lib = lib
    return lib


def ntt_bytes(data: bytes, log_n: int, inverse: bool, coset: bool, in_len: int, threads: int = 0) -> bytes:
    n = 1 << log_n
    buf = ctypes.create_string_buffer(32 * n)
    ctypes.memmove(buf, data, min(len(data), 32 * n))
    rc = load().oracle_ntt(buf, log_n, int(inverse), int(coset), in_len, threads)
    assert rc == 0
    return buf.raw


def msm_bytes(points96: bytes, scalars_mont: bytes, m: int, threads: int = 0) -> bytes:
    out = ctypes.create_string_buffer(97)
    rc = load().oracle_msm(points96, scalars_mont, m, out, threads)
    assert rc == 0
    return out.raw


def max_threads() -> int:
    """Usable host cores: min(OpenMP default, scheduler affinity, cgroup v2 CPU quota)."""
    n = load().oracle_max_threads()
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)
