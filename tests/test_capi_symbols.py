"""CPU-only: the C-ABI library loads and exports every symbol include/plonk_hip.h
declares; without a GPU the product fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

import plonk_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "plonk_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(plonk_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert set(header_symbols()) == set(plonk_amd.EXPORTS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(plonk_amd.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build_hip(verbose=False)
    lib = ctypes.CDLL(plonk_amd.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), f"{name} declared in include/plonk_hip.h but not exported"


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(plonk_amd.PlonkError) as ei:
        plonk_amd.Context(0)
    assert ei.value.code in (-5, -2)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "plonk_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "oracle/" not in txt or f.endswith(".py"), f


def test_python_binding_destroys_provers_before_their_context():
    """include/plonk_hip.h: provers hold a pointer to the context they were built on and must be destroyed first.  The ctypes
    mirror enforces it (no GPU needed: a stub library records the calls)."""
    import weakref

    import plonk_amd
    calls = []

    class Lib:
        def plonk_ctx_destroy(self, h):
            calls.append(("ctx_destroy", h))

        def plonk_prover_destroy(self, h):
            calls.append(("prover_destroy", h))

    ctx = plonk_amd.Context.__new__(plonk_amd.Context)
    ctx.lib, ctx.handle, ctx._provers = Lib(), 111, weakref.WeakSet()
    provers = []
    for h in (222, 333):
        p = plonk_amd.Prover.__new__(plonk_amd.Prover)
        p.ctx, p.handle = ctx, h
        ctx._provers.add(p)
        provers.append(p)
    provers[1].close()
    ctx.close()
    provers[0].close()
    ctx.close()
    assert calls == [("prover_destroy", 333), ("prover_destroy", 222), ("ctx_destroy", 111)]


def test_struct_layouts_of_the_binding_match_the_c_header(tmp_path):
    """plonk_gpu_config / plonk_msm_plan / plonk_prover_info as a C99 compiler lays them out (sizeof + the offset of the last
    field) against the ctypes mirrors the tests drive the library through — and against the Rust `#[repr(C)]` block of
    INTEGRATION.md section 2, which lists the same fields in the same order."""
    import subprocess
    src = tmp_path / "layout.c"
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "plonk_hip.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(plonk_gpu_config), offsetof(plonk_gpu_config, side_stream_cus),\n'
                   '  sizeof(plonk_msm_plan), offsetof(plonk_msm_plan, accumulate_kernel), sizeof(plonk_prover_info), offsetof(plonk_prover_info, lagrange_points));\n'
                   '  return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)], text=True).split()]
    G, M, P = plonk_amd.GpuConfig, plonk_amd._MsmPlan, plonk_amd._ProverInfo
    assert got == [ctypes.sizeof(G), G.side_stream_cus.offset, ctypes.sizeof(M), M.accumulate_kernel.offset,
                   ctypes.sizeof(P), P.lagrange_points.offset]
    assert plonk_amd.GpuConfig().struct_size == ctypes.sizeof(G) == 56
    # the Rust block names every field of the config struct
    rust = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name, _ in G._fields_:
        assert "pub " + name + ":" in rust, name


def test_integration_doc_accounts_for_every_entry_point():
    """every symbol of the header is either declared in the Rust `extern "C"` block or named in the
    paragraph that lists what the shim leaves out — a new entry point cannot be added without telling the maintainer of the
    reference-side binding about it."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc[doc.index('extern "C" {'):doc.index("Error codes →")]
    bound = set(re.findall(r"pub fn (plonk_[a-z0-9_]+)", block))
    rest = doc[doc.index("Error codes →"):doc.index("## 3. Call-site patches")]
    for name in header_symbols():
        if name in bound:
            continue
        stem = name[len("plonk_"):]
        # the paragraph abbreviates families: `plonk_dev_alloc / _free / _h2d / _d2h / _sync`, `plonk_profile_enable / _read / _reset`
        family_tail = "_" + stem.split("_", 1)[1] if "_" in stem else ""
        assert name in rest or (family_tail and family_tail in rest), name


def test_standin_transport_builds_and_exports_the_entry_points_comm_hip_resolves():
    """tests/fake_rccl (test infrastructure): the nine nccl* symbols plonk_amd/csrc/comm.hip looks up with dlsym, by the same
    names — read from comm.hip itself, so that a tenth symbol resolved there cannot be forgotten here."""
    import subprocess
    fake_dir = os.path.join(ROOT, "tests", "fake_rccl")
    subprocess.check_call(["make", "-s", "-C", fake_dir])
    wanted = set(re.findall(r'dlsym\(h, "(nccl\w+)"\)', open(os.path.join(ROOT, "plonk_amd", "csrc", "comm.hip")).read()))
    assert len(wanted) == 9
    lib = ctypes.CDLL(os.path.join(fake_dir, "libfakerccl.so"))
    for name in wanted:
        assert hasattr(lib, name), name
    # and the product never names the stand-in: it is selected by path, by the tests
    for f in os.listdir(os.path.join(ROOT, "plonk_amd", "csrc")):
        assert "libfakerccl" not in open(os.path.join(ROOT, "plonk_amd", "csrc", f), errors="ignore").read(), f
    assert "fakerccl" not in open(os.path.join(ROOT, "plonk_amd", "__init__.py")).read()
