"""plonk_amd — MI355X (gfx950) backend for dusk-plonk's prover hot path.

Host-side mirror of the reference's two crate-private seams, over the C-ABI of
libplonk_hip.so (include/plonk_hip.h):

  Context.ntt(...)   EvaluationDomain::{fft, ifft, coset_fft, coset_ifft}
                     (reference src/fft/domain.rs:166-232)
  Context.srs_load   CommitKey { powers_of_g } (src/commitment_scheme/kzg10/key.rs:37-41)
  Context.msm / commit   CommitKey::commit -> msm_variable_base (key.rs:376-388)

There is NO CPU fallback: if the HIP library or a GPU is missing every call
raises.  Nothing in this package imports `oracle/`.
"""
from __future__ import annotations

import ctypes
import os
import weakref
from typing import Iterable, Sequence

# dmabuf IPC: RCCL between processes needs it on this platform's driver, and it must be in the environment before the
# first HIP call of the process — the host program's job (include/plonk_hip.h, "Multi-GPU"); this binding is that host
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PLONK_HIP_LIB") or os.path.join(_HERE, "lib", "libplonk_hip.so")   # override: A/B builds

# field constants needed to marshal Python ints <-> Montgomery limbs at the ABI
Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
_FR_R = (1 << 256) % Q
_FR_RINV = pow(_FR_R, -1, Q)
_FP_R = (1 << 384) % P
_FP_RINV = pow(_FP_R, -1, P)

PLONK_OK = 0
ERRORS = {
    -1: "PLONK_ERR_ARG", -2: "PLONK_ERR_HIP", -3: "PLONK_ERR_DEGREE (PolynomialDegreeTooLarge)",
    -4: "PLONK_ERR_NO_SRS", -5: "PLONK_ERR_NO_GPU", -6: "PLONK_ERR_UNSAT (CircuitUnsatisfied)",
    -7: "PLONK_ERR_STATE", -8: "PLONK_ERR_BYTES (NotEnoughBytes)", -9: "PLONK_ERR_DATA (InvalidData)",
    -10: "PLONK_ERR_POINT (PointMalformed)", -11: "PLONK_ERR_NOMEM (host allocation failed inside the library)",
}

# every symbol include/plonk_hip.h declares (checked by tests/test_capi_symbols.py)
EXPORTS = [
    "plonk_ctx_create", "plonk_ctx_destroy", "plonk_last_error", "plonk_ntt", "plonk_ntt_batch",
    "plonk_srs_load", "plonk_msm", "plonk_msm_batch", "plonk_ntt_dev", "plonk_msm_dev",
    "plonk_srs_load_dev", "plonk_srs_generate_dev", "plonk_dev_alloc", "plonk_dev_free",
    "plonk_dev_h2d", "plonk_dev_d2h", "plonk_dev_sync", "plonk_ctx_stream", "plonk_ctx_table_rows",
    "plonk_profile_enable", "plonk_profile_read", "plonk_profile_reset",
    "plonk_prover_create", "plonk_prover_destroy", "plonk_prover_vk", "plonk_prover_size",
    "plonk_prover_prove", "plonk_prover_prove_dev", "plonk_prover_peek",
    "plonk_prover_blob_check", "plonk_prover_from_bytes", "plonk_srs_validate",
    "plonk_comm_unique_id", "plonk_comm_init", "plonk_comm_info", "plonk_comm_selftest", "plonk_comm_destroy",
    "plonk_comm_measure_loopback", "plonk_comm_warning", "plonk_prover_set_version", "plonk_comm_set_library", "plonk_comm_library",
    "plonk_ctx_create_ex", "plonk_ctx_get_config", "plonk_ctx_set_config", "plonk_ctx_describe_msm", "plonk_ctx_last_msm",
    "plonk_ctx_table_bytes", "plonk_prover_describe",
    "plonk_host_alloc", "plonk_host_free", "plonk_lagrange_key",
    "plonk_compile", "plonk_prover_prove_witnesses", "plonk_prover_to_bytes", "plonk_verifier_to_bytes",
    "plonk_public_parameters_check", "plonk_srs_load_public_parameters",
]

POLY_ORDER = ["q_m", "q_l", "q_r", "q_o", "q_f", "q_c", "q_arith", "q_range", "q_logic",
              "q_fixed_group_add", "q_variable_group_add", "s_sigma_1", "s_sigma_2", "s_sigma_3", "s_sigma_4"]


TABLE_AUTO, TABLE_WINDOW, TABLE_HALFPOS, TABLE_BITPOS = 0, 16, 128, 256
PLAN_TAIL_SERIAL, PLAN_BUCKET_SUM_LANE, PLAN_ACCUMULATE_LDS, PLAN_SORT13 = 1, 2, 4, 8


class GpuConfig(ctypes.Structure):
    """plonk_gpu_config (include/plonk_hip.h): zero = default for every field; struct_size is filled in by the binding."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("reserved", ctypes.c_uint32), ("table_budget_bytes", ctypes.c_uint64),
                ("table_mode", ctypes.c_int32), ("msm_bucket_bits", ctypes.c_int32), ("quotient_domain", ctypes.c_int32),
                ("wire_commit", ctypes.c_int32), ("shard_quotient", ctypes.c_int32), ("shard_grand_product", ctypes.c_int32),
                ("shard_side_stream", ctypes.c_int32), ("ntt_elements_log2", ctypes.c_int32), ("comm_timeout_ms", ctypes.c_int32),
                ("side_stream_cus", ctypes.c_int32)]

    def __init__(self, **kw):
        super().__init__(**kw)
        self.struct_size = ctypes.sizeof(GpuConfig)

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_ if k not in ("struct_size", "reserved")}


class _MsmPlan(ctypes.Structure):
    _fields_ = [("table_rows", ctypes.c_uint32), ("bucket_bits", ctypes.c_uint32), ("digit_width", ctypes.c_uint32),
                ("slice_entries", ctypes.c_uint32), ("ordered_lanes", ctypes.c_uint32), ("wide_words", ctypes.c_uint32),
                ("flags", ctypes.c_uint32), ("reserved", ctypes.c_uint32), ("terms", ctypes.c_uint64), ("accumulate_kernel", ctypes.c_char * 64)]

    def as_dict(self) -> dict:
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["accumulate_kernel"] = d["accumulate_kernel"].decode()
        return d


class _ProverInfo(ctypes.Structure):
    _fields_ = [("size", ctypes.c_uint64), ("quotient_domain", ctypes.c_uint32), ("wire_commit_values", ctypes.c_uint32),
                ("lagrange_table_rows", ctypes.c_uint32), ("shard_world", ctypes.c_uint32), ("shard_rank", ctypes.c_uint32),
                ("sharded_quotient", ctypes.c_uint32), ("quotient_classes", ctypes.c_uint32), ("wire_group_launches", ctypes.c_uint32),
                ("lagrange_points", ctypes.c_uint64)]


class _ProverDesc(ctypes.Structure):
    _fields_ = [("constraints", ctypes.c_uint64), ("label", ctypes.c_char_p), ("label_len", ctypes.c_uint64),
                ("polys", ctypes.c_void_p * 15), ("poly_len", ctypes.c_uint64 * 15),
                ("vk_commitments", ctypes.c_char_p), ("shard_rank", ctypes.c_int), ("shard_world", ctypes.c_int),
                ("srs_total", ctypes.c_uint64), ("allgather", ctypes.c_void_p), ("allgather_user", ctypes.c_void_p),
                ("lagrange_xy96", ctypes.c_char_p), ("lagrange_count", ctypes.c_uint64)]


class _CircuitDesc(ctypes.Structure):
    _fields_ = [("constraints", ctypes.c_uint64), ("label", ctypes.c_char_p), ("label_len", ctypes.c_uint64),
                ("selectors", ctypes.c_void_p * 11), ("wires", ctypes.c_void_p * 4), ("witnesses", ctypes.c_uint64),
                ("shard_rank", ctypes.c_int), ("shard_world", ctypes.c_int),
                ("srs_total", ctypes.c_uint64), ("allgather", ctypes.c_void_p), ("allgather_user", ctypes.c_void_p),
                ("lagrange_xy96", ctypes.c_char_p), ("lagrange_count", ctypes.c_uint64)]


class _BlobInfo(ctypes.Structure):
    _fields_ = [("size", ctypes.c_uint64), ("constraints", ctypes.c_uint64), ("label_off", ctypes.c_uint64),
                ("label_len", ctypes.c_uint64), ("poly_off", ctypes.c_uint64 * 15), ("poly_len", ctypes.c_uint64 * 15),
                ("srs_off", ctypes.c_uint64), ("srs_points", ctypes.c_uint64), ("vk_off", ctypes.c_uint64)]


class _PublicParametersInfo(ctypes.Structure):
    _fields_ = [("opening_key_off", ctypes.c_uint64), ("points_off", ctypes.c_uint64), ("point_stride", ctypes.c_uint64),
                ("points_total", ctypes.c_uint64), ("points_kept", ctypes.c_uint64)]


PP_RAW_UNCHECKED, PP_RAW, PP_COMPRESSED = 0, 1, 2   # plonk_hip.h PLONK_PP_*


ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64)


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous SRS point range owned by `rank` (same rule as prover.hip)."""
    per = (total + world - 1) // world
    lo = min(per * rank, total)
    return lo, min(lo + per, total)


class CircuitUnsatisfied(Exception):
    """Mirrors Error::CircuitUnsatisfied (reference quotient_poly.rs:132)."""


class PlonkError(RuntimeError):
    def __init__(self, code: int, detail: str = ""):
        self.code = code
        super().__init__(f"{ERRORS.get(code, code)} {detail}".strip())


class NotEnoughBytes(PlonkError):
    """Mirrors Error::NotEnoughBytes (reference prover.rs:274,310; widget.rs:484; key.rs:265,281)."""


class InvalidData(PlonkError):
    """Mirrors dusk_bytes::Error::InvalidData as the reference's decoders raise it."""


class PointMalformed(PlonkError):
    """Mirrors Error::PointMalformed (reference key.rs:292)."""


_DECODE_ERRORS = {-8: NotEnoughBytes, -9: InvalidData, -10: PointMalformed}


class PolynomialDegreeTooLarge(PlonkError):
    """Mirrors Error::PolynomialDegreeTooLarge (reference key.rs:362-370)."""


_lib = None


def load_library() -> ctypes.CDLL:
    """dlopen libplonk_hip.so (built in-tree by __graft_entry__.build()).  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} not built — run `python __graft_entry__.py` (hipcc --offload-arch=gfx950)")
    lib = ctypes.CDLL(LIB_PATH)
    vp, u64, u32, ci = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
    lib.plonk_last_error.restype = ctypes.c_char_p
    lib.plonk_ctx_create.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(ci), ci]
    lib.plonk_ctx_destroy.argtypes = [vp]
    lib.plonk_ctx_destroy.restype = None
    lib.plonk_ntt.argtypes = [vp, vp, u32, ci, ci, u64]
    lib.plonk_ntt_batch.argtypes = [vp, ctypes.POINTER(vp), ci, u32, ci, ci, ctypes.POINTER(u64)]
    lib.plonk_srs_load.argtypes = [vp, vp, u64]
    lib.plonk_msm.argtypes = [vp, vp, u64, vp]
    lib.plonk_msm_batch.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(u64), ci, vp]
    lib.plonk_ntt_dev.argtypes = [vp, vp, vp, vp, u32, ci, ci, u64]
    lib.plonk_msm_dev.argtypes = [vp, vp, u64, vp]
    lib.plonk_srs_load_dev.argtypes = [vp, vp, u64]
    lib.plonk_srs_generate_dev.argtypes = [vp, vp, vp, u64, vp]
    lib.plonk_dev_alloc.argtypes = [vp, u64, ctypes.POINTER(vp)]
    lib.plonk_dev_free.argtypes = [vp, vp]
    lib.plonk_dev_h2d.argtypes = [vp, vp, vp, u64]
    lib.plonk_dev_d2h.argtypes = [vp, vp, vp, u64]
    lib.plonk_dev_sync.argtypes = [vp]
    lib.plonk_ctx_stream.argtypes = [vp]
    lib.plonk_ctx_table_rows.argtypes = [vp]
    lib.plonk_ctx_stream.restype = vp
    lib.plonk_profile_enable.argtypes = [vp, ci]
    lib.plonk_profile_read.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(u64)]
    lib.plonk_profile_reset.argtypes = [vp]
    lib.plonk_prover_create.argtypes = [vp, ctypes.POINTER(_ProverDesc), ctypes.POINTER(vp)]
    lib.plonk_prover_destroy.argtypes = [vp]
    lib.plonk_prover_destroy.restype = None
    lib.plonk_prover_vk.argtypes = [vp, vp]
    lib.plonk_prover_size.argtypes = [vp]
    lib.plonk_prover_size.restype = u64
    lib.plonk_prover_peek.argtypes = [vp, ci, u64, u64, vp]
    lib.plonk_prover_prove.argtypes = [vp, ctypes.POINTER(vp), vp, vp, u64, vp, vp]
    lib.plonk_prover_prove_dev.argtypes = [vp, vp, vp, vp, u64, vp, vp]
    lib.plonk_prover_blob_check.argtypes = [vp, u64, ctypes.POINTER(_BlobInfo)]
    lib.plonk_prover_from_bytes.argtypes = [vp, vp, u64, ctypes.POINTER(vp)]
    lib.plonk_srs_validate.argtypes = [vp, vp, u64]
    lib.plonk_public_parameters_check.argtypes = [vp, u64, u64, ci, ctypes.POINTER(_PublicParametersInfo)]
    lib.plonk_srs_load_public_parameters.argtypes = [vp, vp, u64, u64, ci, vp, ctypes.POINTER(u64)]
    lib.plonk_lagrange_key.argtypes = [vp, u32, vp]
    lib.plonk_compile.argtypes = [vp, ctypes.POINTER(_CircuitDesc), ctypes.POINTER(vp)]
    lib.plonk_prover_prove_witnesses.argtypes = [vp, vp, u64, vp, vp, u64, vp, vp]
    lib.plonk_prover_to_bytes.argtypes = [vp, vp, u64, ctypes.POINTER(u64)]
    lib.plonk_verifier_to_bytes.argtypes = [vp, vp, u64, vp, u64, vp, u64, ctypes.POINTER(u64)]
    lib.plonk_host_alloc.argtypes = [u64, ctypes.POINTER(vp)]
    lib.plonk_host_free.argtypes = [vp]
    lib.plonk_comm_unique_id.argtypes = [vp]
    lib.plonk_comm_init.argtypes = [vp, vp, ci, ci]
    lib.plonk_comm_selftest.argtypes = [vp]
    lib.plonk_comm_info.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci)]
    lib.plonk_comm_destroy.argtypes = [vp]
    lib.plonk_comm_warning.argtypes = [vp, vp, u64]
    lib.plonk_comm_measure_loopback.argtypes = [vp, ci]
    lib.plonk_ctx_create_ex.argtypes = [ctypes.POINTER(vp), ci, ctypes.POINTER(GpuConfig)]
    lib.plonk_ctx_get_config.argtypes = [vp, ctypes.POINTER(GpuConfig)]
    lib.plonk_ctx_set_config.argtypes = [vp, ctypes.POINTER(GpuConfig)]
    lib.plonk_ctx_describe_msm.argtypes = [vp, u64, ci, ci, u32, u64, ctypes.POINTER(_MsmPlan)]
    lib.plonk_ctx_last_msm.argtypes = [vp, ctypes.POINTER(_MsmPlan)]
    lib.plonk_ctx_table_bytes.argtypes = [vp, ctypes.POINTER(u64), ctypes.POINTER(u64)]
    lib.plonk_prover_describe.argtypes = [vp, ctypes.POINTER(_ProverInfo)]
    lib.plonk_comm_set_library.argtypes = [ctypes.c_char_p]
    lib.plonk_comm_library.argtypes = [vp, u64]
    lib.plonk_prover_set_version.argtypes = [vp, ci]
    _lib = lib
    return lib


def prover_blob_check(blob: bytes) -> dict:
    """Host-only decode + validation of a reference Prover::to_bytes() blob (no GPU needed):
    returns the layout (offsets into `blob`) or raises NotEnoughBytes / InvalidData / PointMalformed."""
    lib = load_library()
    info = _BlobInfo()
    rc = lib.plonk_prover_blob_check(blob, len(blob), ctypes.byref(info))
    if rc in _DECODE_ERRORS:
        raise _DECODE_ERRORS[rc](rc, (lib.plonk_last_error() or b"").decode())
    if rc != PLONK_OK:
        raise PlonkError(rc, (lib.plonk_last_error() or b"").decode())
    return {"size": info.size, "constraints": info.constraints,
            "label": blob[info.label_off:info.label_off + info.label_len],
            "polys": {name: (info.poly_off[k], info.poly_len[k]) for k, name in enumerate(POLY_ORDER)},
            "srs": (info.srs_off, info.srs_points), "vk_off": info.vk_off}


def _pp_mode(validate: bool, compressed: bool) -> int:
    return PP_COMPRESSED if compressed else (PP_RAW if validate else PP_RAW_UNCHECKED)


def public_parameters_check(data: bytes, truncated_degree: int = 0, validate: bool = True, compressed: bool = False) -> dict:
    """Host-only decode of a PublicParameters file (reference srs.rs:103-178, key.rs:215-326; no GPU needed) — the raw form
    (`to_raw_var_bytes`, checked or unchecked) or the compressed form (`to_var_bytes`): the opening-key bytes and the layout of
    the (trimmed) commit key, or NotEnoughBytes / InvalidData / PointMalformed / PlonkError(-3) for a trim beyond the key
    (Error::TruncatedDegreeTooLarge)."""
    lib = load_library()
    info = _PublicParametersInfo()
    rc = lib.plonk_public_parameters_check(data, len(data), truncated_degree, _pp_mode(validate, compressed), ctypes.byref(info))
    if rc in _DECODE_ERRORS:
        raise _DECODE_ERRORS[rc](rc, (lib.plonk_last_error() or b"").decode())
    if rc != PLONK_OK:
        raise PlonkError(rc, (lib.plonk_last_error() or b"").decode())
    return {"opening_key": data[info.opening_key_off:info.opening_key_off + 240], "points_off": info.points_off,
            "point_stride": info.point_stride, "points_total": info.points_total, "points_kept": info.points_kept}


# ---- marshalling -----------------------------------------------------------------
def fr_to_bytes_mont(vals: Iterable[int]) -> bytes:
    return b"".join(((v % Q) * _FR_R % Q).to_bytes(32, "little") for v in vals)


def fr_from_bytes_mont(buf: bytes) -> list[int]:
    return [int.from_bytes(buf[i:i + 32], "little") * _FR_RINV % Q for i in range(0, len(buf), 32)]


def g1_to_raw96(pt) -> bytes:
    x, y = pt
    return (x * _FP_R % P).to_bytes(48, "little") + (y * _FP_R % P).to_bytes(48, "little")


def g1_from_raw97(buf: bytes):
    if buf[96]:
        return None
    return (int.from_bytes(buf[:48], "little") * _FP_RINV % P,
            int.from_bytes(buf[48:96], "little") * _FP_RINV % P)


def g1_compress(pt) -> bytes:
    """Commitment::to_bytes — 48-byte compressed G1 (reference commitment.rs:46-57)."""
    if pt is None:
        return bytes([0xC0]) + bytes(47)
    x, y = pt
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= 0x80
    if y > (P - y) % P:
        b[0] |= 0x20
    return bytes(b)


class DeviceBuffer:
    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx, self.nbytes = ctx, nbytes
        p = ctypes.c_void_p()
        ctx._check(ctx.lib.plonk_dev_alloc(ctx.handle, nbytes, ctypes.byref(p)))
        self.ptr = p.value

    def upload(self, data: bytes, offset: int = 0):
        assert offset + len(data) <= self.nbytes
        self.ctx._check(self.ctx.lib.plonk_dev_h2d(self.ctx.handle, self.ptr + offset, data, len(data)))

    def download(self, nbytes: int | None = None, offset: int = 0) -> bytes:
        nbytes = self.nbytes - offset if nbytes is None else nbytes
        out = ctypes.create_string_buffer(nbytes)
        self.ctx._check(self.ctx.lib.plonk_dev_d2h(self.ctx.handle, out, self.ptr + offset, nbytes))
        return out.raw

    def free(self):
        if self.ptr:
            self.ctx.lib.plonk_dev_free(self.ctx.handle, self.ptr)
            self.ptr = None


class PinnedBuffer:
    """Pinned host memory (plonk_host_alloc): uploads from it are asynchronous, which is what lets
    plonk_srs_load overlap the streamed key with the table build."""

    def __init__(self, nbytes: int):
        self.lib = load_library()
        p = ctypes.c_void_p()
        rc = self.lib.plonk_host_alloc(nbytes, ctypes.byref(p))
        if rc != PLONK_OK:
            raise PlonkError(rc, (self.lib.plonk_last_error() or b"").decode())
        self.ptr, self.nbytes = p.value, nbytes

    def write(self, data: bytes, offset: int = 0):
        assert offset + len(data) <= self.nbytes
        ctypes.memmove(self.ptr + offset, data, len(data))

    def free(self):
        if self.ptr:
            self.lib.plonk_host_free(self.ptr)
            self.ptr = None


class Context:
    """One GPU, one stream (plonk_ctx).  One process per GPU in multi-GPU runs."""

    def __init__(self, device: int = 0, config: "GpuConfig | None" = None):
        self.lib = load_library()
        h = ctypes.c_void_p()
        if config is None:
            dev = (ctypes.c_int * 1)(device)
            rc = self.lib.plonk_ctx_create(ctypes.byref(h), dev, 1)
        else:
            rc = self.lib.plonk_ctx_create_ex(ctypes.byref(h), device, ctypes.byref(config))
        if rc != PLONK_OK:
            raise PlonkError(rc, (self.lib.plonk_last_error() or b"").decode())
        self.handle = h
        self.srs_points = 0
        self._provers = weakref.WeakSet()   # provers hold a pointer to the native context: close() destroys them first (plonk_hip.h)

    def close(self):
        if getattr(self, "handle", None):
            for p in list(getattr(self, "_provers", ())):
                p.close()
            self.lib.plonk_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc == -3:
            raise PolynomialDegreeTooLarge(rc)
        if rc == -6:
            raise CircuitUnsatisfied()
        if rc in _DECODE_ERRORS:
            raise _DECODE_ERRORS[rc](rc, (self.lib.plonk_last_error() or b"").decode())
        if rc != PLONK_OK:
            raise PlonkError(rc, (self.lib.plonk_last_error() or b"").decode())

    # ---- EvaluationDomain seam ------------------------------------------------------
    def ntt_bytes(self, data: bytes, log_n: int, inverse: bool, coset: bool, in_len: int) -> bytes:
        n = 1 << log_n
        buf = ctypes.create_string_buffer(32 * n)
        ctypes.memmove(buf, data, min(len(data), 32 * n))
        self._check(self.lib.plonk_ntt(self.handle, buf, log_n, int(inverse), int(coset), in_len))
        return buf.raw

    def ntt_batch_bytes(self, datas, log_n: int, inverse: bool, coset: bool, in_lens=None) -> list[bytes]:
        """compute_coset_evaluations' fan-out (quotient_poly.rs:139-157) as one plonk_ntt_batch call."""
        n = 1 << log_n
        bufs = []
        for d in datas:
            b = ctypes.create_string_buffer(32 * n)
            ctypes.memmove(b, d, min(len(d), 32 * n))
            bufs.append(b)
        arr = (ctypes.c_void_p * max(len(bufs), 1))(*[ctypes.cast(b, ctypes.c_void_p) for b in bufs])
        lens = None
        if in_lens is not None:
            lens = (ctypes.c_uint64 * max(len(bufs), 1))(*in_lens)
        self._check(self.lib.plonk_ntt_batch(self.handle, arr, len(bufs), log_n, int(inverse), int(coset), lens))
        return [b.raw for b in bufs]

    def msm_batch_bytes(self, scalar_sets) -> list[bytes]:
        """commit_polynomials' fan-out (prover.rs:187-210) as one plonk_msm_batch call: list of
        Montgomery scalar byte strings -> list of 97-byte raw results."""
        keep = [ctypes.create_string_buffer(bytes(s), max(len(s), 1)) for s in scalar_sets]
        arr = (ctypes.c_void_p * max(len(keep), 1))(*[ctypes.cast(b, ctypes.c_void_p) for b in keep])
        ms = (ctypes.c_uint64 * max(len(keep), 1))(*[len(s) // 32 for s in scalar_sets])
        out = ctypes.create_string_buffer(97 * max(len(keep), 1))
        self._check(self.lib.plonk_msm_batch(self.handle, arr, ms, len(keep), out))
        return [out.raw[97 * i:97 * i + 97] for i in range(len(keep))]

    def ntt(self, values: Sequence[int], log_n: int, inverse: bool = False, coset: bool = False) -> list[int]:
        """fft / ifft / coset_fft / coset_ifft on Python ints: zero-pads or truncates to
        2^log_n exactly like Vec::resize at reference domain.rs:174."""
        n = 1 << log_n
        vals = list(values[:n])
        return fr_from_bytes_mont(self.ntt_bytes(fr_to_bytes_mont(vals), log_n, inverse, coset,
                                                 len(vals) if not inverse else n))

    # ---- CommitKey seam ------------------------------------------------------------
    def srs_load(self, points) -> None:
        raw = b"".join(g1_to_raw96(p) for p in points)
        self.srs_load_bytes(raw, len(points))

    def srs_load_bytes(self, raw: bytes, npoints: int) -> None:
        self._check(self.lib.plonk_srs_load(self.handle, raw, npoints))
        self.srs_points = npoints

    def srs_load_public_parameters(self, data: bytes, truncated_degree: int = 0, validate: bool = True,
                                   compressed: bool = False) -> bytes:
        """PublicParameters::from_slice_unchecked / CommitKey::from_raw_var_bytes / PublicParameters::from_slice (compressed)
        + trim(truncated_degree) straight into the context's window tables (plonk_srs_load_public_parameters); returns the
        240 opening-key bytes."""
        ok = ctypes.create_string_buffer(240)
        n = ctypes.c_uint64(0)
        self._check(self.lib.plonk_srs_load_public_parameters(self.handle, data, len(data), truncated_degree,
                                                              _pp_mode(validate, compressed), ok, ctypes.byref(n)))
        self.srs_points = n.value
        return ok.raw

    def srs_load_host_ptr(self, ptr: int, npoints: int) -> None:
        """plonk_srs_load from a raw host address (e.g. inside a PinnedBuffer): streamed in chunks."""
        self._check(self.lib.plonk_srs_load(self.handle, ctypes.c_void_p(ptr), npoints))
        self.srs_points = npoints

    def d2h_into(self, host_ptr: int, dev_ptr: int, nbytes: int) -> None:
        self._check(self.lib.plonk_dev_d2h(self.handle, ctypes.c_void_p(host_ptr), dev_ptr, nbytes))

    def h2d_from(self, dev_ptr: int, host_ptr: int, nbytes: int) -> None:
        """plonk_dev_h2d from a raw host address (e.g. PinnedBuffer.ptr); queued on the context's stream"""
        self._check(self.lib.plonk_dev_h2d(self.handle, dev_ptr, ctypes.c_void_p(host_ptr), nbytes))

    def msm_bytes(self, scalars_mont: bytes, m: int) -> bytes:
        out = ctypes.create_string_buffer(97)
        self._check(self.lib.plonk_msm(self.handle, scalars_mont, m, out))
        return out.raw

    def msm(self, scalars: Sequence[int]):
        """msm_variable_base(&powers_of_g, scalars) -> affine point or None (identity)."""
        return g1_from_raw97(self.msm_bytes(fr_to_bytes_mont(scalars), len(scalars)))

    def commit(self, coeffs: Sequence[int]):
        """CommitKey::commit (key.rs:376-388): trailing zeros trimmed like
        Polynomial::from_coefficients_vec; degree check before the MSM."""
        c = list(coeffs)
        while c and c[-1] % Q == 0:
            c.pop()
        if len(c) > self.srs_points:
            raise PolynomialDegreeTooLarge(-3)
        return self.msm(c)

    # ---- device-resident API ----------------------------------------------------
    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    # ---- configuration / introspection ----------------------------------------------------
    def get_config(self) -> "GpuConfig":
        """the EFFECTIVE configuration (defaults and environment overrides resolved)"""
        g = GpuConfig()
        self._check(self.lib.plonk_ctx_get_config(self.handle, ctypes.byref(g)))
        return g

    def set_config(self, config: "GpuConfig"):
        self._check(self.lib.plonk_ctx_set_config(self.handle, ctypes.byref(config)))

    def describe_msm(self, m: int, count: int = 1, bit_sum_tail: bool = True, table_rows: int = 0, table_points: int = 0) -> dict:
        """what an MSM of `count` sets of <= m terms would run as (table_rows = 0: over the context's commit key)"""
        p = _MsmPlan()
        self._check(self.lib.plonk_ctx_describe_msm(self.handle, m, count, int(bit_sum_tail), table_rows, table_points, ctypes.byref(p)))
        return p.as_dict()

    def last_msm(self) -> dict:
        """what the last MSM group on this context did run as"""
        p = _MsmPlan()
        self._check(self.lib.plonk_ctx_last_msm(self.handle, ctypes.byref(p)))
        return p.as_dict()

    def table_bytes(self) -> tuple[int, int]:
        """(bytes of point tables the context holds, its budget)"""
        a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self.lib.plonk_ctx_table_bytes(self.handle, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def table_rows(self) -> int:
        """256: one table row per bit position (width-17 NAF digits), 16: window rows, 0: no key"""
        return int(self.lib.plonk_ctx_table_rows(self.handle))

    def ntt_dev(self, src: int, dst: int, tmp: int, log_n: int, inverse=False, coset=False, in_len=None):
        in_len = (1 << log_n) if in_len is None else in_len
        self._check(self.lib.plonk_ntt_dev(self.handle, src, dst, tmp, log_n, int(inverse), int(coset), in_len))

    def msm_dev(self, scalars: int, m: int, out97: int):
        self._check(self.lib.plonk_msm_dev(self.handle, scalars, m, out97))

    def srs_load_dev(self, ptr: int, npoints: int):
        self._check(self.lib.plonk_srs_load_dev(self.handle, ptr, npoints))
        self.srs_points = npoints

    def srs_generate_dev(self, tau: int, g_scalar: int, npoints: int, out_ptr: int):
        self._check(self.lib.plonk_srs_generate_dev(self.handle, fr_to_bytes_mont([tau]),
                                                    fr_to_bytes_mont([g_scalar]), npoints, out_ptr))

    def lagrange_key(self, log_n: int) -> bytes:
        """(n + 2) x 96 B: [L_i(tau)] G and the two blinding points, from the context's commit key (plonk_lagrange_key)."""
        out = ctypes.create_string_buffer(96 * ((1 << log_n) + 2))
        self._check(self.lib.plonk_lagrange_key(self.handle, log_n, out))
        return out.raw

    def sync(self):
        self._check(self.lib.plonk_dev_sync(self.handle))

    # ---- multi-GPU (RCCL inside the library) -----------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """ncclGetUniqueId on rank 0; hand the 128 bytes to the other ranks out of band."""
        lib = load_library()
        out = ctypes.create_string_buffer(128)
        rc = lib.plonk_comm_unique_id(out)
        if rc != PLONK_OK:
            raise PlonkError(rc, (lib.plonk_last_error() or b"").decode())
        return out.raw

    @staticmethod
    def comm_set_library(path: str):
        """Name the transport library (default: librccl) before the first communicator call of the process."""
        lib = load_library()
        rc = lib.plonk_comm_set_library(os.fsencode(path))
        if rc != PLONK_OK:
            raise PlonkError(rc, (lib.plonk_last_error() or b"").decode())

    @staticmethod
    def comm_library() -> str:
        """Path of the transport library that was actually loaded (dladdr of its ncclAllGather)."""
        lib = load_library()
        out = ctypes.create_string_buffer(1024)
        rc = lib.plonk_comm_library(out, 1024)
        if rc != PLONK_OK:
            raise PlonkError(rc, (lib.plonk_last_error() or b"").decode())
        return out.value.decode()

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == 128
        self._check(self.lib.plonk_comm_init(self.handle, unique_id, rank, world))

    def comm_info(self) -> tuple[int, int]:
        """(rank, size) as the RCCL communicator reports them"""
        r, w = ctypes.c_int(-1), ctypes.c_int(-1)
        self._check(self.lib.plonk_comm_info(self.handle, ctypes.byref(r), ctypes.byref(w)))
        return r.value, w.value

    def comm_warning(self) -> str:
        """The note plonk_comm_init left on this context ("" = none); a successful call never writes the last-error text."""
        out = ctypes.create_string_buffer(512)
        self._check(self.lib.plonk_comm_warning(self.handle, out, 512))
        return out.value.decode()

    def comm_selftest(self):
        self._check(self.lib.plonk_comm_selftest(self.handle))

    def comm_destroy(self):
        self._check(self.lib.plonk_comm_destroy(self.handle))

    def comm_measure_loopback(self, on: bool = True):
        """MEASUREMENT ONLY (tools/rank_alone.py): collectives of this context return the rank's own contribution."""
        self._check(self.lib.plonk_comm_measure_loopback(self.handle, int(on)))

    def profile(self, on: bool):
        self._check(self.lib.plonk_profile_enable(self.handle, int(on)))

    def profile_reset(self):
        self._check(self.lib.plonk_profile_reset(self.handle))

    def profile_read(self, slot: int):
        ms, n = ctypes.c_double(), ctypes.c_uint64()
        self._check(self.lib.plonk_profile_read(self.handle, slot, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value


class Prover:
    """Device-resident mirror of the reference `Prover` (src/compiler/prover.rs:27-42):
    built from the 15 ProverKey polynomials + label + constraint count; `prove` takes the
    padded wire columns, sparse public inputs and the 14 blinders drawn by the caller's RNG
    (prover.rs:154-161,133-135,553-555) and returns Proof::to_bytes (1008 bytes)."""

    def __init__(self, ctx: Context, constraints: int, label: bytes, polys: dict, vk_commitments: bytes | None = None,
                 rank: int = 0, world: int = 1, srs_total: int = 0, allgather=None, lagrange_slice: bytes | None = None):
        """allgather(send: bytes) -> bytes (rank-major concatenation) when world > 1 and the context has no RCCL
        communicator; lagrange_slice: this rank's points of Context.lagrange_key (multi-GPU)."""
        self.ctx = ctx
        desc = _ProverDesc()
        self._shard(desc, rank, world, srs_total, allgather, lagrange_slice)
        desc.constraints = constraints
        desc.label = label
        desc.label_len = len(label)
        self._keep = []
        for k, name in enumerate(POLY_ORDER):
            coeffs = polys.get(name, [])
            raw = coeffs if isinstance(coeffs, (bytes, bytearray)) else fr_to_bytes_mont(coeffs)
            buf = ctypes.create_string_buffer(bytes(raw), max(len(raw), 1))
            self._keep.append(buf)
            desc.polys[k] = ctypes.cast(buf, ctypes.c_void_p)
            desc.poly_len[k] = len(raw) // 32
        desc.vk_commitments = vk_commitments
        h = ctypes.c_void_p()
        ctx._check(ctx.lib.plonk_prover_create(ctx.handle, ctypes.byref(desc), ctypes.byref(h)))
        self.handle = h
        ctx._provers.add(self)
        self.size = ctx.lib.plonk_prover_size(h)
        self._keep = None

    def describe(self) -> dict:
        """what the prover was built as (plonk_prover_describe): quotient domain, wire-commitment mode, sharding"""
        info = _ProverInfo()
        self.ctx._check(self.ctx.lib.plonk_prover_describe(self.handle, ctypes.byref(info)))
        return {k: getattr(info, k) for k, _ in info._fields_ if k != "reserved"}

    @classmethod
    def from_bytes(cls, ctx: Context, blob: bytes) -> "Prover":
        """Prover::try_from_bytes (reference prover.rs:266-345) on the output of the reference's
        Prover::to_bytes(): validates the blob, loads its commit key into `ctx` and builds the device
        prover.  Raises NotEnoughBytes / InvalidData / PointMalformed like the reference."""
        self = cls.__new__(cls)
        self.ctx, self._cb, self._keep = ctx, None, None
        h = ctypes.c_void_p()
        ctx._check(ctx.lib.plonk_prover_from_bytes(ctx.handle, blob, len(blob), ctypes.byref(h)))
        self.handle = h
        ctx._provers.add(self)
        self.size = ctx.lib.plonk_prover_size(h)
        return self

    def _shard(self, desc, rank, world, srs_total, allgather, lagrange_slice):
        self._cb = None
        if world > 1:
            desc.shard_rank, desc.shard_world, desc.srs_total = rank, world, srs_total
        if world > 1 and allgather is not None:
            def _cb(user, send, recv, nbytes):
                try:
                    out = allgather(ctypes.string_at(send, nbytes))
                    assert len(out) == nbytes * world
                    ctypes.memmove(recv, out, len(out))
                    return 0
                except Exception:   # never unwind through the C frame
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = ALLGATHER_FN(_cb)
            desc.allgather = ctypes.cast(self._cb, ctypes.c_void_p)
        if lagrange_slice is not None:   # an empty slice still selects the mode (every rank must take the same one)
            desc.lagrange_xy96 = lagrange_slice if lagrange_slice else b"\0"
            desc.lagrange_count = len(lagrange_slice) // 96

    @classmethod
    def compile(cls, ctx: Context, label: bytes, selectors: dict, wires, witnesses: int, rank: int = 0, world: int = 1,
                srs_total: int = 0, allgather=None, lagrange_slice: bytes | None = None) -> "Prover":
        """Compiler::preprocess (reference src/compiler.rs:132-461) on the device.  selectors: {name: per-gate values},
        names from POLY_ORDER[:11], values as ints or Montgomery bytes, missing = zero; wires: four sequences of witness
        indices (one per gate); witnesses: how many witnesses the composer allocated."""
        self = cls.__new__(cls)
        self.ctx, self._keep = ctx, []
        desc = _CircuitDesc()
        self._shard(desc, rank, world, srs_total, allgather, lagrange_slice)
        def u32_bytes(w):   # list of ints, raw little-endian uint32 bytes, or anything with tobytes() (a uint32 array)
            if isinstance(w, (bytes, bytearray)):
                return bytes(w)
            if hasattr(w, "tobytes"):
                assert getattr(w, "itemsize", 4) == 4
                return w.tobytes()
            return bytes((ctypes.c_uint32 * len(w))(*w))
        wire_raw = [u32_bytes(w) for w in wires]
        constraints = len(wire_raw[0]) // 4
        desc.constraints, desc.label, desc.label_len, desc.witnesses = constraints, label, len(label), witnesses
        for k, name in enumerate(POLY_ORDER[:11]):
            col = selectors.get(name)
            if col is None or len(col) == 0:
                continue
            raw = bytes(col) if isinstance(col, (bytes, bytearray)) else fr_to_bytes_mont(col)
            assert len(raw) == 32 * constraints, name
            self._keep.append(raw)   # the library reads the bytes object in place
            desc.selectors[k] = ctypes.cast(ctypes.c_char_p(raw), ctypes.c_void_p)
        for w in range(4):
            assert len(wire_raw[w]) == 4 * constraints
            desc.wires[w] = ctypes.cast(ctypes.c_char_p(wire_raw[w]), ctypes.c_void_p)
        h = ctypes.c_void_p()
        ctx._check(ctx.lib.plonk_compile(ctx.handle, ctypes.byref(desc), ctypes.byref(h)))
        self.handle = h
        ctx._provers.add(self)
        self.size = ctx.lib.plonk_prover_size(h)
        self._keep = None
        return self

    def prove_witnesses(self, witnesses, public_inputs, blinders) -> bytes:
        """Proof from the witness values (ints or Montgomery bytes) on a compiled prover (prover.rs:446-460 on the device)."""
        raw = bytes(witnesses) if isinstance(witnesses, (bytes, bytearray)) else fr_to_bytes_mont(witnesses)
        bl = bytes(blinders) if isinstance(blinders, (bytes, bytearray)) else fr_to_bytes_mont(blinders)
        assert len(bl) == 14 * 32
        idx, val, cnt = self._pi(public_inputs)
        proof = ctypes.create_string_buffer(1008)
        self.ctx._check(self.ctx.lib.plonk_prover_prove_witnesses(self.handle, raw, len(raw) // 32, idx, val, cnt, bl, proof))
        return proof.raw

    def prove_witnesses_ptr(self, values_ptr: int, count: int, public_inputs, blinders_mont: bytes) -> bytes:
        """prove_witnesses on a raw host address (e.g. PinnedBuffer.ptr) holding count x 32 bytes."""
        idx, val, cnt = self._pi(public_inputs)
        proof = ctypes.create_string_buffer(1008)
        self.ctx._check(self.ctx.lib.plonk_prover_prove_witnesses(self.handle, values_ptr, count, idx, val, cnt, blinders_mont, proof))
        return proof.raw

    def to_bytes(self) -> bytes:
        """Prover::to_bytes() (reference prover.rs:238-263) of this prover and its context's commit key."""
        n = ctypes.c_uint64()
        self.ctx._check(self.ctx.lib.plonk_prover_to_bytes(self.handle, None, 0, ctypes.byref(n)))
        buf = ctypes.create_string_buffer(n.value)
        self.ctx._check(self.ctx.lib.plonk_prover_to_bytes(self.handle, buf, n.value, ctypes.byref(n)))
        return buf.raw

    def verifier_to_bytes(self, opening_key: bytes, public_input_indexes) -> bytes:
        """Verifier::to_bytes() (reference verifier.rs:88-117); opening_key = OpeningKey::to_bytes() of the caller's parameters."""
        idx = list(public_input_indexes)
        arr = (ctypes.c_uint64 * max(len(idx), 1))(*idx)
        n = ctypes.c_uint64()
        self.ctx._check(self.ctx.lib.plonk_verifier_to_bytes(self.handle, opening_key, len(opening_key), arr, len(idx), None, 0, ctypes.byref(n)))
        buf = ctypes.create_string_buffer(n.value)
        self.ctx._check(self.ctx.lib.plonk_verifier_to_bytes(self.handle, opening_key, len(opening_key), arr, len(idx), buf, n.value, ctypes.byref(n)))
        return buf.raw

    def vk_commitments(self) -> bytes:
        out = ctypes.create_string_buffer(15 * 48)
        self.ctx._check(self.ctx.lib.plonk_prover_vk(self.handle, out))
        return out.raw

    @staticmethod
    def _pi(public_inputs):
        items = sorted(public_inputs.items()) if isinstance(public_inputs, dict) else list(public_inputs or [])
        idx = (ctypes.c_uint64 * max(len(items), 1))(*[i for i, _ in items])
        val = fr_to_bytes_mont([v for _, v in items])
        return idx, val, len(items)

    def prove(self, wires, public_inputs, blinders) -> bytes:
        """wires: 4 sequences of ints (length <= size, zero padded); blinders: 14 ints."""
        assert len(blinders) == 14
        n = self.size
        bufs = []
        for w in wires:
            raw = w if isinstance(w, (bytes, bytearray)) else fr_to_bytes_mont(list(w) + [0] * (n - len(w)))
            assert len(raw) == 32 * n
            bufs.append(ctypes.create_string_buffer(bytes(raw), len(raw)))
        arr = (ctypes.c_void_p * 4)(*[ctypes.cast(b, ctypes.c_void_p) for b in bufs])
        idx, val, cnt = self._pi(public_inputs)
        proof = ctypes.create_string_buffer(1008)
        self.ctx._check(self.ctx.lib.plonk_prover_prove(self.handle, arr, idx, val, cnt,
                                                        fr_to_bytes_mont(blinders), proof))
        return proof.raw

    def prove_host_bytes(self, wires, public_inputs, blinders_mont: bytes) -> bytes:
        """plonk_prover_prove on four byte strings of size x 32 B (pageable host memory), blinders as Montgomery bytes"""
        n = self.size
        assert all(len(w) == 32 * n for w in wires)
        bufs = [ctypes.create_string_buffer(bytes(w), 32 * n) for w in wires]
        return self.prove_host_ptrs([ctypes.addressof(b) for b in bufs], public_inputs, blinders_mont)

    def prove_host_ptrs(self, wire_ptrs, public_inputs, blinders_mont: bytes) -> bytes:
        """plonk_prover_prove on four raw host addresses (e.g. PinnedBuffer.ptr): the columns are uploaded on the copy
        stream while round 1 already transforms the ones that have arrived."""
        arr = (ctypes.c_void_p * 4)(*[ctypes.c_void_p(p) for p in wire_ptrs])
        idx, val, cnt = self._pi(public_inputs)
        proof = ctypes.create_string_buffer(1008)
        self.ctx._check(self.ctx.lib.plonk_prover_prove(self.handle, arr, idx, val, cnt, blinders_mont, proof))
        return proof.raw

    def set_version(self, version: int):
        """Prover::prove_with_version (prover.rs:365-413): 3 (default) or the legacy 2 (transcript seeding only)."""
        self.ctx._check(self.ctx.lib.plonk_prover_set_version(self.handle, version))

    def prove_dev(self, wires_ptr: int, public_inputs, blinders_mont: bytes) -> bytes:
        idx, val, cnt = self._pi(public_inputs)
        proof = ctypes.create_string_buffer(1008)
        self.ctx._check(self.ctx.lib.plonk_prover_prove_dev(self.handle, wires_ptr, idx, val, cnt,
                                                            blinders_mont, proof))
        return proof.raw

    def peek(self, which: int, offset: int, count: int) -> list[int]:
        out = ctypes.create_string_buffer(32 * count)
        self.ctx._check(self.ctx.lib.plonk_prover_peek(self.handle, which, offset, count, out))
        return fr_from_bytes_mont(out.raw)

    def close(self):
        if getattr(self, "handle", None):
            if getattr(self.ctx, "handle", None):          # a closed context already destroyed its provers
                self.ctx.lib.plonk_prover_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
