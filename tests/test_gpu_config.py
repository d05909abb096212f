"""GPU: plonk_gpu_config (SURVEY.md section 5 "config / flag system"; include/plonk_hip.h) — what a context builds follows
the struct it was created with, the table layout of a key follows the context's BUDGET and what the context itself holds
(not the memory that happens to be free), and the library reports what it chose (plonk_ctx_describe_msm / _last_msm)."""
import pytest

pytestmark = pytest.mark.gpu

N_KEY = (1 << 18) + 100          # the smallest key size whose layout is chosen by the budget rule (msm.hip msm_table_rows)
BITPOS_BYTES = 256 * 128 * N_KEY  # 8.6 GB


def _load_key(ctx, n=N_KEY):
    pts = ctx.alloc(96 * n)
    ctx.srs_generate_dev(0x1234567, 0x89ABCDEF, n, pts.ptr)
    ctx.srs_load_dev(pts.ptr, n)
    pts.free()


def test_defaults_and_round_trip():
    import plonk_amd
    ctx = plonk_amd.Context(0, plonk_amd.GpuConfig())
    g = ctx.get_config()
    assert g.quotient_domain == 4 and g.wire_commit == 0 and g.table_mode == plonk_amd.TABLE_AUTO and g.comm_timeout_ms == 120000
    # default budget: 80 % of the device's total memory — an MI355X reports 287.98 GiB (no torch here: a second HIP runtime in
    # the process, torch's bundled one, is not something this test should depend on)
    assert 0.79 * 288 * 2**30 <= g.table_budget_bytes <= 0.80 * 288 * 2**30
    assert ctx.table_bytes() == (0, g.table_budget_bytes)
    g.table_mode, g.quotient_domain, g.comm_timeout_ms = plonk_amd.TABLE_WINDOW, 8, 5000
    ctx.set_config(g)
    h = ctx.get_config()
    assert (h.table_mode, h.quotient_domain, h.comm_timeout_ms) == (16, 8, 5000)
    assert plonk_amd.Context(0).get_config().as_dict() == plonk_amd.Context(0, plonk_amd.GpuConfig()).get_config().as_dict()
    ctx.close()


def test_a_refused_reconfiguration_changes_nothing():
    """plonk_ctx_set_config used to overwrite the context's configuration BEFORE it rejected a change of
    side_stream_cus (the streams are created once), so the rejected call still replaced every other field.  It now resolves
    into a temporary and commits only what it accepts."""
    import plonk_amd
    ctx = plonk_amd.Context(0, plonk_amd.GpuConfig())
    before = ctx.get_config().as_dict()
    g = ctx.get_config()
    g.side_stream_cus, g.quotient_domain, g.table_mode, g.comm_timeout_ms = 32, 8, plonk_amd.TABLE_WINDOW, 777
    with pytest.raises(plonk_amd.PlonkError) as ei:
        ctx.set_config(g)
    assert ei.value.code == -7
    assert ctx.get_config().as_dict() == before
    ctx.close()


def test_invalid_configurations_are_refused():
    import plonk_amd
    for kw in ({"table_mode": 17}, {"msm_bucket_bits": 16}, {"quotient_domain": 2}, {"ntt_elements_log2": 4}, {"shard_quotient": 2},
               {"comm_timeout_ms": -1}):
        with pytest.raises(plonk_amd.PlonkError) as ei:
            plonk_amd.Context(0, plonk_amd.GpuConfig(**kw))
        assert ei.value.code == -1, kw
    bad = plonk_amd.GpuConfig()
    bad.struct_size = 0
    with pytest.raises(plonk_amd.PlonkError):
        plonk_amd.Context(0, bad)


def test_same_budget_same_layout_whatever_the_neighbours_hold():
    """Round 4 chose a key's rows from hipMemGetInfo's "free right now": a second context on the device saw less free memory and
    silently took another layout.  The rule now reads the context's budget and its own tables only."""
    import plonk_amd
    big = plonk_amd.GpuConfig(table_budget_bytes=24 << 30)      # bit-position rows of the key (8.6 GB + 3.2 GB of build scratch) are within 60 % of it
    a = plonk_amd.Context(0, big)
    _load_key(a)
    assert a.table_rows() == 256 and a.table_bytes() == (BITPOS_BYTES, 24 << 30)
    ballast = a.alloc(150 << 30)                                # a neighbour that takes most of the device
    b = plonk_amd.Context(0, big)
    _load_key(b)
    assert b.table_rows() == 256                                 # same budget -> same layout
    small = plonk_amd.Context(0, plonk_amd.GpuConfig(table_budget_bytes=12 << 30))
    _load_key(small)
    assert small.table_rows() == 16 and small.table_bytes()[0] == 16 * 128 * N_KEY
    # what the MSMs over these keys run as, predicted and observed
    pa, ps = a.describe_msm(N_KEY, 4), small.describe_msm(N_KEY, 4)
    assert (pa["table_rows"], pa["bucket_bits"], pa["digit_width"], pa["accumulate_kernel"]) == (256, 19, 21, "nbl::msm_accumulate_ordered_kernel")
    assert (ps["table_rows"], ps["bucket_bits"], ps["digit_width"]) == (16, 15, 16)
    import numpy as np
    raw = np.random.default_rng(1).integers(0, 256, size=(N_KEY, 32), dtype=np.uint8)
    raw[:, 31] &= 0x3F                                          # Montgomery limbs of scalars below 2^254 < q
    r_a = a.msm_bytes(raw.tobytes(), N_KEY)
    assert a.last_msm() == dict(a.describe_msm(N_KEY, 1), terms=N_KEY)
    r_s = small.msm_bytes(raw.tobytes(), N_KEY)
    assert small.last_msm()["accumulate_kernel"].startswith("nb15::")
    assert r_a == r_s                                           # same commitment from either layout
    # forcing a layout for ONE load (a key that only feeds a derivation): set_config, load, restore
    g = a.get_config()
    g.table_mode = plonk_amd.TABLE_WINDOW
    a.set_config(g)
    _load_key(a)
    assert a.table_rows() == 16 and a.table_bytes()[0] == 16 * 128 * N_KEY
    ballast.free()
    for c in (a, b, small):
        c.close()
