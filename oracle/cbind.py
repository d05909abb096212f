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
    srcs = [os.path.join(_DIR, "oracle_prove.c"), os.path.join(_DIR, "oracle.c")]   # oracle_prove.c includes oracle.c
    so = os.path.join(_DIR, f"liboracle_{_cpu_tag()}.so")
    def stale():
        return not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs)
    if stale():
        # several test processes may get here together (the variant children of tests/test_gpu_msm_variants.py run four at a
        # time on a fresh GPU box, where this CPU's copy does not exist yet): one builds, under a file lock, into a temporary
        # name that is renamed into place — nobody ever maps a half-written library
        import fcntl
        with open(so + ".lock", "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if stale():
                    tmp = f"{so}.tmp{os.getpid()}"
                    subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", srcs[0], "-o", tmp])
                    os.replace(tmp, so)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    lib = ctypes.CDLL(so)
    lib.oracle_ntt.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_int]
    lib.oracle_msm.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int]
    lib.oracle_max_threads.restype = ctypes.c_int
    vp, u64 = ctypes.c_void_p, ctypes.c_uint64
    lib.oracle_prover_new.restype = vp
    lib.oracle_prover_new.argtypes = [u64, vp, u64, ctypes.POINTER(vp), ctypes.POINTER(u64), vp, u64, vp, ctypes.c_int]
    lib.oracle_prover_free.argtypes = [vp]
    lib.oracle_prover_free.restype = None
    lib.oracle_prover_vk.argtypes = [vp, vp]
    lib.oracle_prover_vk.restype = None
    lib.oracle_prover_vk_trapdoor.argtypes = [vp, vp]
    lib.oracle_prover_adopt_vk_trapdoor.argtypes = [vp]
    lib.oracle_prover_adopt_vk_trapdoor.restype = ctypes.c_int
    lib.oracle_prover_vk_trapdoor.restype = ctypes.c_int
    lib.oracle_prover_set_trapdoor.argtypes = [vp, vp, vp]
    lib.oracle_prover_set_trapdoor.restype = None
    lib.oracle_prover_set_version.argtypes = [vp, ctypes.c_int]
    lib.oracle_prover_set_version.restype = None
    lib.oracle_srs_generate.argtypes = [vp, vp, u64, vp, ctypes.c_int]
    lib.oracle_prover_prove.argtypes = [vp, ctypes.POINTER(vp), vp, vp, u64, vp, vp, vp]
    _lib = lib
    return lib


def _threads(threads: int) -> int:
    """0 = every core this process may use (cgroup quota / affinity), never the machine's full core count:
    OpenMP's own default oversubscribes a quota-limited container by an order of magnitude."""
    return threads if threads > 0 else max_threads()


def ntt_bytes(data: bytes, log_n: int, inverse: bool, coset: bool, in_len: int, threads: int = 0) -> bytes:
    threads = _threads(threads)
    n = 1 << log_n
    buf = ctypes.create_string_buffer(32 * n)
    ctypes.memmove(buf, data, min(len(data), 32 * n))
    rc = load().oracle_ntt(buf, log_n, int(inverse), int(coset), in_len, threads)
    assert rc == 0
    return buf.raw


def msm_bytes(points96: bytes, scalars_mont: bytes, m: int, threads: int = 0) -> bytes:
    out = ctypes.create_string_buffer(97)
    rc = load().oracle_msm(points96, scalars_mont, m, out, _threads(threads))
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


def srs_generate(tau_mont: bytes, g_scalar_mont: bytes, n: int, threads: int = 0) -> bytes:
    """[g tau^i] G1 for i < n as n x 96 B raw points (PublicParameters::setup semantics, srs.rs:61-100)."""
    out = ctypes.create_string_buffer(96 * max(n, 1))
    rc = load().oracle_srs_generate(tau_mont, g_scalar_mont, n, out, _threads(threads))
    assert rc == 0
    return out.raw[:96 * n]


# ---- whole prove() (oracle_prove.c) ------------------------------------------------------
POLY_ORDER = ["q_m", "q_l", "q_r", "q_o", "q_f", "q_c", "q_arith", "q_range", "q_logic",
              "q_fixed_group_add", "q_variable_group_add", "s_sigma_1", "s_sigma_2", "s_sigma_3", "s_sigma_4"]


class _Trace(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in
                ("wire_polys", "z_poly", "t_poly", "w_z", "w_zw", "evals", "challenges", "seconds")]


class CircuitUnsatisfied(Exception):
    pass


class CProver:
    """oracle_prover_new / oracle_prover_prove: the C restatement of Prover::new + prove_inner.
    polys: {name: bytes of Montgomery limbs (32 B per coefficient)} in coefficient form;
    srs96: npoints x 96 B; vk48: 15 x 48 B in POLY_ORDER or None (commit here)."""

    def __init__(self, constraints: int, label: bytes, polys: dict, srs96: bytes, vk48: bytes | None = None, threads: int = 0):
        lib = load()
        self.lib = lib
        bufs = [bytes(polys.get(name, b"")) for name in POLY_ORDER]
        arr = (ctypes.c_void_p * 15)(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p) for b in bufs])
        lens = (ctypes.c_uint64 * 15)(*[len(b) // 32 for b in bufs])
        n = 1
        while n < constraints:
            n *= 2
        self.n = n
        self.threads = _threads(threads)
        self.h = lib.oracle_prover_new(constraints, label, len(label), arr, lens, srs96, len(srs96) // 96, vk48, self.threads)
        if not self.h:
            raise ValueError("oracle_prover_new failed (polynomial longer than the domain, or degree > SRS)")

    def set_trapdoor(self, tau_mont: bytes, g_scalar_mont: bytes):
        """the key is [g tau^i] G with known tau, g: commitments become [g p(tau)] G (same group elements, no MSM)"""
        self.lib.oracle_prover_set_trapdoor(self.h, tau_mont, g_scalar_mont)

    def set_version(self, version: int):
        """prove_with_version (prover.rs:365-413): 3 (default) or the legacy 2"""
        self.lib.oracle_prover_set_version(self.h, version)

    def vk(self) -> bytes:
        out = ctypes.create_string_buffer(15 * 48)
        self.lib.oracle_prover_vk(self.h, out)
        return out.raw

    def adopt_vk_trapdoor(self):
        """make the trapdoor commitments of the key polynomials this prover's VerifierKey (replaces a placeholder vk48)"""
        rc = self.lib.oracle_prover_adopt_vk_trapdoor(self.h)
        assert rc == 0, rc

    def vk_trapdoor(self) -> bytes:
        """the 15 key commitments recomputed as [g p(tau)] G (set_trapdoor first) — independent of a vk48 passed to __init__"""
        out = ctypes.create_string_buffer(15 * 48)
        rc = self.lib.oracle_prover_vk_trapdoor(self.h, out)
        assert rc == 0, rc
        return out.raw

    def prove(self, wires, pi_idx, pi_val_mont: bytes, blinders_mont: bytes, trace: bool = False):
        """wires: 4 x bytes (n x 32 B Montgomery); returns proof bytes (and a dict of stage arrays)."""
        n = self.n
        assert all(len(w) == 32 * n for w in wires) and len(blinders_mont) == 14 * 32
        wb = [ctypes.create_string_buffer(bytes(w), 32 * n) for w in wires]
        warr = (ctypes.c_void_p * 4)(*[ctypes.cast(b, ctypes.c_void_p) for b in wb])
        idx = (ctypes.c_uint64 * max(len(pi_idx), 1))(*pi_idx)
        proof = ctypes.create_string_buffer(1008)
        tr = _Trace()
        secs = (ctypes.c_double * 6)()
        tr.seconds = ctypes.cast(secs, ctypes.c_void_p)
        keep = {}
        if trace:
            for name, cnt in (("wire_polys", 4 * (n + 8)), ("z_poly", n + 8), ("t_poly", 8 * n), ("w_z", n + 8), ("w_zw", n + 8),
                              ("evals", 15), ("challenges", 10)):
                keep[name] = ctypes.create_string_buffer(32 * cnt)
                setattr(tr, name, ctypes.cast(keep[name], ctypes.c_void_p))
        rc = self.lib.oracle_prover_prove(self.h, warr, idx, pi_val_mont, len(pi_idx), blinders_mont, proof, ctypes.byref(tr))
        self.seconds = dict(zip(("ntt", "msm", "quotient", "perm", "tail", "total"), secs))
        if rc == -6:
            raise CircuitUnsatisfied()
        if rc == -3:
            raise ValueError("PolynomialDegreeTooLarge")
        if rc:
            raise ValueError(f"oracle_prover_prove rc={rc}")
        if trace:
            return proof.raw, {k: v.raw for k, v in keep.items()}
        return proof.raw

    def close(self):
        if self.h:
            self.lib.oracle_prover_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
