"""-m gpu: the engine reports the order in which its kernels add a segment's terms (glrm_hip_sum_order, include/glrm_hip.h) and the CPU
oracle, adding in THAT order (glrm_cpu_set_sum_order; pinned to a lane-by-lane simulation of the kernels in tests/test_sum_order.py),
lands on the engine's factors BIT FOR BIT -- for every sweep family of the scalar-loss path.  That is the attribution tool SURVEY.md
section 7.3 item 1 asks for: whatever separates an engine trajectory from the reference-order oracle (tests/test_gpu_jref.py: 8e-6 after
100 iterations on 1e8 observations) is summation order and nothing else, because with the order matched nothing is left.
Reference lines whose sums these are: src/algorithms/proxgrad.jl:122-132,143,165-175,187; src/evaluate_fit.jl:24-55.
Losses: the formulas both sides evaluate identically (Quad, Huber, OrdinalHinge ...); the exp / log based ones use in-kernel routines
that differ from libm in the last bits (csrc/glrm_fastmath.hpp) and are compared at 1e-5 elsewhere."""
import numpy as np
import pytest

import cases
import lowrankmodels.jl_amd as L
import oracle as O
from lowrankmodels.jl_amd import _capi

pytestmark = pytest.mark.gpu
TILED_R, TILED_C, BLOCKED_R, BLOCKED_C, CACHED = 1, 2, 16, 32, 64


def problem(m, n, k, q, reg, value_model=0, mixed=False):
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(m, n, k, q, value_model=value_model, loss_mix=1 if mixed else 0)
    if mixed:  # a descriptor per column, formulas without exp / log: Quad / Huber / OrdinalHinge(1, 5) by column mod 3
        kinds = [L.QuadLoss(0.8).descriptor(), L.HuberLoss(1.1, crossover=0.7).descriptor(), L.OrdinalHingeLoss(1, 5, 0.9).descriptor()]
        losses = np.array([kinds[f % 3] for f in range(n)], dtype=_capi.LOSS_DTYPE)
    else:
        losses = np.array([(0, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)
    r = np.array([reg], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, losses, r, r)
    if reg[0] == 3:
        X0, Y0 = np.abs(X0) / 8.0, np.abs(Y0) / 8.0
    else:
        X0, Y0 = 0.3 * X0, 0.3 * Y0
    return pa, np.asfortranarray(X0), np.asfortranarray(Y0)


def engine_and_oracle_in_its_order(pa, X0, Y0, iters, want_flags, want_families, forbid_flags=0, **create_kw):
    api, oapi = _capi.hip_api(), O.oracle_api()
    prm = L.ProxGradParams(max_iter=iters, abs_tol=-1e300, rel_tol=-1e300)
    h = api.create(pa, **create_kw)
    try:
        flags = api.kernel_stats(h)["tiled"]
        assert flags & want_flags == want_flags and not flags & forbid_flags, flags
        orders = [api.sum_order(h, 0), api.sum_order(h, 1)]
        assert [o.asdict()["family_name"] for o in orders] == list(want_families), [o.asdict() for o in orders]
        Xg, Yg = X0.copy(order="F"), Y0.copy(order="F")
        og, _ = api.fit(h, prm, Xg, Yg)
        st_g = api.kernel_stats(h)
    finally:
        api.destroy(h)
    O.set_threads(O.usable_cores())
    res = {}
    for name, use in (("engine order", orders), ("reference order", [None, None])):
        ho = oapi.create(pa)
        try:
            for w, o in enumerate(use):
                O.set_sum_order(ho, w, o)
            Xc, Yc = X0.copy(order="F"), Y0.copy(order="F")
            oc, _ = oapi.fit(ho, prm, Xc, Yc)
            res[name] = (oc, Xc, Yc, oapi.kernel_stats(ho))
        finally:
            oapi.destroy(ho)
    oc, Xc, Yc, st_c = res["engine order"]
    assert np.array_equal(Xg, Xc), ("X differs from the oracle in the engine's order", np.abs(Xg - Xc).max(), [o.asdict() for o in orders])
    assert np.array_equal(Yg, Yc), ("Y differs from the oracle in the engine's order", np.abs(Yg - Yc).max())
    for key in ("trials_x", "trials_y", "accepts_x", "accepts_y"):
        assert st_g[key] == st_c[key], (key, st_g[key], st_c[key])        # every accept / reject decision the same
    assert cases.rel_err(og[1:], oc[1:]) < 1e-13                          # the recorded objective: only the final sum over columns differs
    orf, Xr, Yr, _ = res["reference order"]
    assert cases.rel_err(og, orf) < 1e-5 and cases.fro_err(Xg, Xr) < 1e-5 and cases.fro_err(Yg, Yr) < 1e-5
    return orders


def test_gather_sweeps_one_and_four_wave_segments(monkeypatch):
    """sweep_kernel: rows of 50 observations on one wave, columns of 2 000 on four (the class comes from the segment's own length)."""
    monkeypatch.setenv("GLRM_HIP_CACHED", "0")
    pa, X0, Y0 = problem(20000, 500, 64, 50, (3, 0, 1.0), value_model=1)
    o = engine_and_oracle_in_its_order(pa, X0, Y0, 8, 0, ("strided", "strided"), forbid_flags=TILED_R | TILED_C | BLOCKED_R | BLOCKED_C | CACHED, tiled=1)
    assert o[0].lanes == 8 and o[0].comps == 8 and o[0].waves == 0 and o[0].cached_maxlen == -1


def test_cached_row_sweep_two_waves_per_row(monkeypatch):
    """regcached_persist_kernel (the C4 X half-step): rows of <= 104 observations on two waves, longer ones on the gather sweep."""
    monkeypatch.setenv("GLRM_HIP_CACHED", "1")
    monkeypatch.setenv("GLRM_HIP_BLOCKED", "0")
    pa, X0, Y0 = problem(12000, 2000, 64, 100, (3, 0, 1.0), value_model=1)
    o = engine_and_oracle_in_its_order(pa, X0, Y0, 8, CACHED, ("strided", "strided"))
    assert o[0].cached_maxlen == 104 and o[0].cached_waves == 2


@pytest.mark.parametrize("k", [32, 64])
def test_lds_tiled_sweeps_quadloss(k):
    """tiled_sweep_kernel / tiled_col_pass_kernel + col_reduce / col_decide (the C2 families): list order in tile windows, two loss
    partial sums per lane group, super-tile partials added in order."""
    pa, X0, Y0 = problem(6000, 1500, k, 300, (1, 0, 1.0))
    o = engine_and_oracle_in_its_order(pa, X0, Y0, 8, TILED_R | TILED_C, ("windowed", "windowed"), tiled=2)
    assert o[0].windows_per_sup == 0 and o[1].windows_per_sup >= 1 and o[0].batch == 2 and not o[0].private_order
    # padded rank 32: both sides run the lane-per-segment form of the passes (csrc/glrm_lane.hpp), reported as the two-lane layout with the
    # rotated chunk walk; rank 64 stays on the four / eight-lane kernels
    assert [(x.lanes, x.comps, x.rotate) for x in o] == ([(2, 16, 2)] * 2 if k == 32 else [(8, 8, 0)] * 2)


def test_lds_tiled_sweeps_quadloss_on_the_four_lane_kernels(monkeypatch):
    """GLRM_HIP_LANE=0: the four-lane tiled kernels at rank 32 (what rows of models with a loss per column, views beyond 2e9 observations and
    the sparse solver's fixed-step sweeps keep running on)."""
    monkeypatch.setenv("GLRM_HIP_LANE", "0")
    pa, X0, Y0 = problem(6000, 1500, 32, 300, (1, 0, 1.0))
    o = engine_and_oracle_in_its_order(pa, X0, Y0, 8, TILED_R | TILED_C, ("windowed", "windowed"), tiled=2)
    assert [(x.lanes, x.comps, x.rotate, x.batch) for x in o] == [(4, 8, 0, 2)] * 2


def test_phase_aligned_passes(monkeypatch):
    """tiled_col_pass_kernel<..., L2 = true> (the C4 Y half-step) on both views, several super-tiles and launch slices."""
    monkeypatch.setenv("GLRM_HIP_BLOCKED", "3")
    monkeypatch.setenv("GLRM_HIP_BLOCKED_TPS", "1")
    monkeypatch.setenv("GLRM_HIP_BLOCKED_FILL", "3")
    monkeypatch.setenv("GLRM_HIP_CACHED", "0")
    pa, X0, Y0 = problem(20000, 2000, 64, 100, (3, 0, 1.0), value_model=1)
    o = engine_and_oracle_in_its_order(pa, X0, Y0, 8, BLOCKED_R | BLOCKED_C, ("windowed", "windowed"), tiled=1)
    assert o[0].windows_per_sup == 1 and o[1].windows_per_sup == 1


def test_heterogeneous_columns_on_the_lds_tiles(monkeypatch):
    """A loss descriptor per column (Quad / Huber / OrdinalHinge): the whole batch of four observations per step with one loss partial
    per lane, the conflict-free chunk walk of the column passes (rotate), the LDS descriptor table.  The row view is NOT regrouped by
    loss kind here (GLRM_HIP_GROUP_KINDS=0): a regrouped private copy adds in an order the caller's lists do not determine."""
    monkeypatch.setenv("GLRM_HIP_GROUP_KINDS", "0")
    monkeypatch.setenv("GLRM_HIP_LANE", "0")
    pa, X0, Y0 = problem(5000, 1500, 32, 300, (1, 0, 1.0), mixed=True)
    o = engine_and_oracle_in_its_order(pa, X0, Y0, 6, TILED_R | TILED_C, ("windowed", "windowed"), tiled=2)
    assert o[0].batch == 4 and o[1].batch == 4 and o[1].rotate == 1 and o[0].rotate == 0


def test_heterogeneous_columns_on_the_lane_per_segment_passes(monkeypatch):
    """The same model with both views on the lane-per-segment passes (the default at rank 32 since session r6_33) -- the column view with ONE
    loss descriptor per column, the row view with a descriptor per OBSERVATION: its one-byte id rides in the SELL offset word, the descriptors
    sit in LDS behind the tile, and the row view keeps the caller's order (no kind grouping: every lane evaluates its own observation's
    formula).  The trial rounds after the first run all three forms of the pass (full grid / gathered from the layout / CSR)."""
    monkeypatch.delenv("GLRM_HIP_LANE_PER_OBS", raising=False)
    pa, X0, Y0 = problem(5000, 1500, 32, 300, (1, 0, 1.0), mixed=True)
    o = engine_and_oracle_in_its_order(pa, X0, Y0, 6, TILED_R | TILED_C, ("windowed", "windowed"), tiled=2)
    assert [(x.lanes, x.comps, x.batch, x.rotate, x.private_order) for x in o] == [(2, 16, 2, 2, 0)] * 2


def test_heterogeneous_rows_on_the_four_lane_kernels(monkeypatch):
    """GLRM_HIP_LANE_PER_OBS=0 (the default until session r6_33, and what views beyond 6e9 observations or models of more than 256 distinct
    descriptors run): the lane-per-segment passes on the column view, the four-lane kernels (kind-grouped windows, private_order = 2) on
    the row view."""
    monkeypatch.setenv("GLRM_HIP_LANE_PER_OBS", "0")
    pa, X0, Y0 = problem(5000, 1500, 32, 300, (1, 0, 1.0), mixed=True)
    o = engine_and_oracle_in_its_order(pa, X0, Y0, 6, TILED_R | TILED_C, ("windowed", "windowed"), tiled=2)
    assert (o[0].lanes, o[0].batch, o[0].rotate, o[0].private_order) == (4, 4, 0, 2) and (o[1].lanes, o[1].comps, o[1].batch, o[1].rotate) == (2, 16, 2, 2)


def test_heterogeneous_columns_on_the_gather_sweeps():
    """sweep_pass SCATTER: four observations per trip, four loss partials per lane group (rows: every wave count; columns: one-wave
    segments only)."""
    pa, X0, Y0 = problem(4000, 600, 32, 60, (1, 0, 1.0), mixed=True)
    o = engine_and_oracle_in_its_order(pa, X0, Y0, 6, 0, ("strided", "strided"), forbid_flags=TILED_R | TILED_C | BLOCKED_R | BLOCKED_C | CACHED, tiled=1)
    assert o[0].batch == 4 and o[0].batch_one_wave_only == 0 and o[1].batch == 4 and o[1].batch_one_wave_only == 1


def test_rows_regrouped_by_loss_kind_are_followed_by_the_oracle(monkeypatch):
    """With the kind grouping of the four-lane kernels (GLRM_HIP_LANE_PER_OBS=0) the row view of a heterogeneous model is walked in a private order -- inside every tile window the
    entries grouped by ascending loss kind, stably (glrm_tiled.hpp: group_rows_by_kind_kernel).  That is a function of the caller's list
    alone: the engine reports private_order = 2 and the oracle restates the grouping (glrm_cpu_set_sum_order) -- bit-identical factors.
    A TILE-SORTED private copy (private_order = 1) stays out of the oracle's reach."""
    monkeypatch.setenv("GLRM_HIP_LANE_PER_OBS", "0")
    pa, X0, Y0 = problem(5000, 1500, 32, 300, (1, 0, 1.0), mixed=True)
    o = engine_and_oracle_in_its_order(pa, X0, Y0, 6, TILED_R | TILED_C, ("windowed", "windowed"), tiled=2)
    assert o[0].private_order == 2 and o[1].private_order == 0 and o[0].batch == 4
    ho = O.oracle_api().create(pa)
    try:
        o[0].private_order = 1
        with pytest.raises(_capi.GLRMError):
            O.set_sum_order(ho, 0, o[0])
    finally:
        O.oracle_api().destroy(ho)