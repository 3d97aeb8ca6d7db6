"""-m gpu: oracle parity for exactly the kernel families the BASELINE north-star configuration (C4: 10M x 100k, rank 64, 100 sorted
observations per row, NonNegConstraint) dispatches at full size -- the phase-aligned gather passes (csrc/glrm_blocked.hip,
`tiled_col_pass_kernel<..., L2 = true>`) and the two-wave register-cached row sweep `regcached_sweep_kernel<8, 8, *, 7, 2>`
(csrc/glrm_cached.hip) -- forced onto problems the oracle finishes in seconds, with super-tiles and launch slices shrunk so that
several of each are exercised; and the shard-invariance of every kernel choice on ragged data (bit-identical
for any number of shards).  Reference being restated: src/algorithms/proxgrad.jl:118-201, src/evaluate_fit.jl:24-55.
"""
import os

import numpy as np
import pytest

import cases
import lowrankmodels.jl_amd as L
import oracle as O
from lowrankmodels.jl_amd import _capi

pytestmark = pytest.mark.gpu
TOL = 1e-5
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BLOCKED_ROWS, BLOCKED_COLS, CACHED = 16, 32, 64


def hip():
    return _capi.hip_api()


def force_blocked(monkeypatch, tps="1", fill="3"):
    """Phase-aligned passes on both views; one LDS-tile unit per super-tile (many super-tiles: the partial sums per (segment, super-tile)
    and their fixed-order reduction are exercised) and 3 % of the chip's residency per launch slice (many slices per pass)."""
    monkeypatch.setenv("GLRM_HIP_BLOCKED", "3")
    monkeypatch.setenv("GLRM_HIP_BLOCKED_TPS", tps)
    monkeypatch.setenv("GLRM_HIP_BLOCKED_FILL", fill)
    monkeypatch.setenv("GLRM_HIP_CACHED", "0")


def against_oracle(pa, X0, Y0, params, want_flags, tol=TOL, **create_kw):
    O.set_threads(4)
    o_c, X_c, Y_c, st_c = cases.run_engine(O.oracle_api(), pa, X0, Y0, params)
    o_g, X_g, Y_g, st_g = cases.run_engine(hip(), pa, X0, Y0, params, **create_kw)
    assert st_g["tiled"] & want_flags == want_flags, (st_g["tiled"], want_flags)
    assert len(o_g) == len(o_c)
    e = (cases.rel_err(o_g, o_c), cases.fro_err(X_g, X_c), cases.fro_err(Y_g, Y_c))
    assert max(e) < tol, e
    for key in ("trials_x", "trials_y", "accepts_x", "accepts_y"):
        assert abs(st_g[key] - st_c[key]) <= max(5, 0.03 * st_c[key]), (key, st_g[key], st_c[key])
    assert st_g["nnz_rows"] == st_c["nnz_rows"] and st_g["nnz_cols"] == st_c["nnz_cols"]
    return e


def c4_problem(m, n, q, k=64):
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(m, n, k, q, value_model=1)
    one = np.array([(0, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(3, 0, 1.0)], dtype=_capi.REG_DTYPE)  # NonNegConstraint, src/regularizers.jl:101-114
    pa = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, one, reg, reg)
    return pa, np.asfortranarray(np.abs(X0) / 8.0), np.asfortranarray(np.abs(Y0) / 8.0), X0, Y0


# ------------------------------------------------------------------------------------------------ phase-aligned gather passes

@pytest.mark.parametrize("start", ["nonneg", "randn"])
def test_phase_aligned_passes_c4_recipe(monkeypatch, start):
    """The C4 recipe at 20 000 x 2 000 on the family that runs the full-size Y half-step (and, with the cached sweep off, the X
    half-step): 7 / 70 super-tiles per view, ~60 slices per pass.  `randn`: the reference default start, objective Inf, collapse to X = 0."""
    force_blocked(monkeypatch)
    pa, Xn, Yn, Xr, Yr = c4_problem(20000, 2000, 100)
    X0, Y0 = (Xn, Yn) if start == "nonneg" else (Xr, Yr)
    against_oracle(pa, X0, Y0, L.ProxGradParams(max_iter=12), BLOCKED_ROWS | BLOCKED_COLS, tiled=1)


def test_phase_aligned_passes_c4_golden_fixture(monkeypatch):
    force_blocked(monkeypatch)
    pa, X0, Y0, params, z = cases.load_case(os.path.join(GOLDEN, "c4.npz"))
    obj, X, Y, st = cases.run_engine(hip(), pa, X0, Y0, params, tiled=1)
    assert st["tiled"] & (BLOCKED_ROWS | BLOCKED_COLS) == BLOCKED_ROWS | BLOCKED_COLS
    assert len(obj) == len(z["objective"]) and cases.rel_err(obj, z["objective"]) < TOL
    assert cases.fro_err(X, z["X"]) < TOL and cases.fro_err(Y, z["Y"]) < TOL


@pytest.mark.parametrize("k", [32, 64])
def test_phase_aligned_passes_mixed_losses_sorted_lists(monkeypatch, k):
    """Quad / Logistic / OrdinalHinge columns (the C5 recipe), a loss descriptor per column: the per-observation and the per-segment
    loss variants of the pass kernels, four- and eight-lane layouts."""
    force_blocked(monkeypatch)
    m, n, q = 2500, 2000, 100
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(m, n, k, q, value_model=0, loss_mix=1)
    kinds = [L.QuadLoss().descriptor(), L.LogisticLoss().descriptor(), L.OrdinalHingeLoss(1, 5).descriptor()]
    losses = np.array([kinds[f % 3] for f in range(n)], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(1, 0, 1.0)], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, losses, reg, reg)
    against_oracle(pa, 0.3 * X0, 0.3 * Y0, L.ProxGradParams(max_iter=10), BLOCKED_ROWS | BLOCKED_COLS, tiled=1)
    hub = np.array([L.HuberLoss(1.0, crossover=0.5).descriptor()], dtype=_capi.LOSS_DTYPE)   # one non-quadratic loss: the segment-uniform variant
    pa2 = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, hub, reg, reg)
    against_oracle(pa2, 0.3 * X0, 0.3 * Y0, L.ProxGradParams(max_iter=8), BLOCKED_ROWS | BLOCKED_COLS, tiled=1)


def test_phase_aligned_passes_row_chunks_and_sparse_solver(monkeypatch):
    """glrm_hip_step_x_range in chunks = one full sweep (bitwise), and fit!(::SparseProxGradParams) -- the fixed-step gradient passes
    of the same family -- against the oracle."""
    force_blocked(monkeypatch)
    pa, X0, Y0, _, _ = c4_problem(6000, 1500, 100)
    api = hip()
    res = []
    for chunks in (None, [(0, 1000), (1000, 1001), (1001, 4096), (4096, 6000)]):
        h = api.create(pa, tiled=1)
        assert api.kernel_stats(h)["tiled"] & (BLOCKED_ROWS | BLOCKED_COLS) == BLOCKED_ROWS | BLOCKED_COLS
        api.set_factors(h, X0, Y0)
        api.reset_stepsizes(h, 1.0)
        for _ in range(3):
            if chunks is None:
                api.step_x(h, 0.01)
            else:
                for b, e in chunks:
                    api.step_x_range(h, b, e, 0.01)
            api.step_y(h, 0.01)
        X, Y = np.zeros_like(X0), np.zeros_like(Y0)
        api.get_factors(h, X, Y)
        st = api.kernel_stats(h)
        api.destroy(h)
        res.append((X, Y, st["trials_x"], st["accepts_x"]))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and res[0][2:] == res[1][2:]
    sp = L.SparseProxGradParams(max_iter=12)
    outs = []
    for a_ in (O.oracle_api(), api):
        h = a_.create(pa, tiled=1) if a_ is api else a_.create(pa)
        X, Y = np.array(X0, order="F"), np.array(Y0, order="F")
        obj, _ = a_.fit_sparse(h, sp, X, Y)
        a_.destroy(h)
        outs.append((np.array(obj), X, Y))
    assert len(outs[0][0]) == len(outs[1][0]) and cases.rel_err(outs[1][0], outs[0][0]) < TOL
    assert cases.fro_err(outs[1][1], outs[0][1]) < TOL and cases.fro_err(outs[1][2], outs[0][2]) < TOL


# ------------------------------------------------------------------------------------------------ lockstep windows (column view)

def force_lockstep(monkeypatch, wt="1", spin="4000"):
    """Column passes as one persistent kernel that walks the opposing factor in windows of `wt` LDS-tile units (288 rows at k = 64) and
    meets the other workgroups of its XCD after every window (csrc/glrm_blocked.hip: lockstep_col_pass_kernel); rows on the phase-aligned
    launches.  spin = polls before a workgroup gives the meeting up (1: practically at once -- results must not depend on it)."""
    force_blocked(monkeypatch)
    monkeypatch.setenv("GLRM_HIP_LOCKSTEP", "1")
    monkeypatch.setenv("GLRM_HIP_LOCKSTEP_WT", wt)
    monkeypatch.setenv("GLRM_HIP_LOCKSTEP_SPIN", spin)


@pytest.mark.parametrize("spin", ["4000", "1"])
def test_lockstep_windows_c4_recipe(monkeypatch, spin):
    """The C4 recipe at 20 000 x 2 000 with 70 one-tile windows; meeting the XCD's other workgroups or giving up on them changes no bit."""
    force_lockstep(monkeypatch, spin=spin)
    pa, Xn, Yn, _, _ = c4_problem(20000, 2000, 100)
    against_oracle(pa, Xn, Yn, L.ProxGradParams(max_iter=12), BLOCKED_ROWS | BLOCKED_COLS, tiled=1)
    a = cases.run_engine(hip(), pa, Xn, Yn, L.ProxGradParams(max_iter=6), tiled=1)
    monkeypatch.setenv("GLRM_HIP_LOCKSTEP_SPIN", "4000" if spin == "1" else "1")
    monkeypatch.setenv("GLRM_HIP_LOCKSTEP_WT", "3")   # another window size: another grouping of the loss sums only
    b = cases.run_engine(hip(), pa, Xn, Yn, L.ProxGradParams(max_iter=6), tiled=1)
    assert cases.rel_err(b[0], a[0]) < 1e-12 and cases.fro_err(b[1], a[1]) < 1e-9 and cases.fro_err(b[2], a[2]) < 1e-9


def test_lockstep_windows_two_residency_rounds_shards_and_mixed_losses(monkeypatch):
    """40 000 columns: more column groups than the chip holds at once (two rounds per pass); ragged column shards reproduce the single handle
    bit for bit; k = 32 with Quad / Logistic / OrdinalHinge columns on the per-observation-descriptor variant."""
    force_lockstep(monkeypatch, wt="2")
    pa, Xn, Yn, _, _ = c4_problem(3000, 40000, 100)
    params = L.ProxGradParams(max_iter=5)
    against_oracle(pa, Xn, Yn, params, BLOCKED_ROWS | BLOCKED_COLS, tiled=1)
    api = hip()
    o1, X1, Y1, _ = cases.run_engine(api, pa, Xn, Yn, params, tiled=1)
    o2, X2, Y2, _ = cases.run_shards_on_one_device(api, pa, Xn, Yn, params, [0, 1000, 3000], [0, 33001, 40000], tiled=1)
    assert np.array_equal(o2, o1[1:]) and np.array_equal(X2, X1) and np.array_equal(Y2, Y1)
    m, n, k, q = 2500, 2000, 32, 100
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(m, n, k, q, value_model=0, loss_mix=1)
    kinds = [L.QuadLoss().descriptor(), L.LogisticLoss().descriptor(), L.OrdinalHingeLoss(1, 5).descriptor()]
    losses = np.array([kinds[f % 3] for f in range(n)], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(1, 0, 1.0)], dtype=_capi.REG_DTYPE)
    pa3 = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, losses, reg, reg)
    against_oracle(pa3, 0.3 * X0, 0.3 * Y0, L.ProxGradParams(max_iter=10), BLOCKED_ROWS | BLOCKED_COLS, tiled=1)


# ------------------------------------------------------------------------------------------------ register-cached row sweep, MAXT = 7

def test_regcached_two_waves_seven_trips_at_full_c4_density(monkeypatch):
    """3 000 x 100 000, rank 64, 100 observations per row: a row is 13 trips of the eight-lane layout, so the cached sweep runs as
    regcached_sweep_kernel<8, 8, 0, 7, 2> -- the instantiation of the full-size C4 X half-step -- and is compared with the oracle."""
    monkeypatch.setenv("GLRM_HIP_CACHED", "1")
    pa, X0, Y0, _, _ = c4_problem(3000, 100000, 100)
    h = hip().create(pa)
    st = hip().kernel_stats(h)
    hip().destroy(h)
    assert st["tiled"] & CACHED and st["ld"] == 64 and not st["tiled"] & 3
    against_oracle(pa, X0, Y0, L.ProxGradParams(max_iter=6), CACHED)


# ------------------------------------------------------------------------------------------------ shard invariance on ragged data

def ragged_problem(rng, m, n, k, q, long_rows=(), long_len=150, long_cols=(), nonneg=True):
    """q sorted observations per row, except `long_rows` (long_len each) and `long_cols` (fully observed)."""
    A = rng.random((m, n))
    feats = []
    for e in range(m):
        cnt = long_len if e in long_rows else q
        feats.append(np.sort(rng.choice(n, size=cnt, replace=False)))
    mask = np.zeros((m, n), dtype=bool)
    for e, f in enumerate(feats):
        mask[e, f] = True
    for c in long_cols:
        mask[:, c] = True
    I, J = np.nonzero(mask)
    X0, Y0 = np.abs(rng.standard_normal((k, m))) / 8, np.abs(rng.standard_normal((k, n))) / 8
    reg = L.NonNegConstraint() if nonneg else L.QuadReg(0.1)
    g = L.GLRM(A, L.QuadLoss(), reg, reg, k, obs=(I, J), X=X0, Y=Y0)
    return g.problem_arrays(), np.asfortranarray(X0), np.asfortranarray(Y0)


@pytest.mark.parametrize("cached", ["1", "auto"])
@pytest.mark.parametrize("x_chunks", [1, 3])
def test_two_shards_equal_one_shard_with_heavy_tailed_rows(monkeypatch, cached, x_chunks):
    """One 150-observation row among 100-observation rows at rank 64: the cached sweep holds rows of up to 104 observations in
    registers (two waves per row), the long row runs the gather sweep -- decided per ROW, so the shard that holds the long row and the
    shard that does not sum every row exactly like the single-shard fit.  Round 2 chose the family from the shard's longest row:
    the two halves then ran different kernels."""
    if cached == "1":
        monkeypatch.setenv("GLRM_HIP_CACHED", "1")
    rng = np.random.default_rng(2024)
    m, n, k = 600, 400, 64
    pa, X0, Y0 = ragged_problem(rng, m, n, k, 100, long_rows=(17,), long_len=150)
    params = L.ProxGradParams(max_iter=6)
    api = hip()
    o1, X1, Y1, st1 = cases.run_engine(api, pa, X0, Y0, params, tiled=0)
    if cached == "1":
        assert st1["tiled"] & CACHED
    for rb, cb in (([0, 300, m], [0, 200, n]), ([0, 10, 20, m], [0, 399, 399, n])):
        o2, X2, Y2, sts = cases.run_shards_on_one_device(api, pa, X0, Y0, params, rb, cb, x_chunks=x_chunks, tiled=0)
        assert all(s["tiled"] == st1["tiled"] for s in sts if s["nnz_rows"] > 0), ([s["tiled"] for s in sts], st1["tiled"])
        assert np.array_equal(X1, X2) and np.array_equal(Y1, Y2) and np.array_equal(o1[1:], o2)
    # and the whole thing against the oracle
    against_oracle(pa, X0, Y0, params, st1["tiled"] & CACHED, tiled=0)


def test_two_shards_equal_one_shard_with_skewed_segment_lengths():
    """A few long columns and one long row (>= 1536 observations: four waves each) next to short ones (one wave): the wave count is a
    function of the segment's own length, so a shard whose MEAN length differs from the whole problem's picks the same kernels.  Also
    through glrm_hip_step_x_range (the pipelined hosts) and the in-library multi-shard fit on one device."""
    rng = np.random.default_rng(7)
    m, n, k = 5000, 1700, 16
    pa, X0, Y0 = ragged_problem(rng, m, n, k, 20, long_rows=(4321,), long_len=1600, long_cols=(3, 900), nonneg=False)
    params = L.ProxGradParams(max_iter=5)
    api = hip()
    o1, X1, Y1, st1 = cases.run_engine(api, pa, X0, Y0, params, tiled=1)
    assert st1["waves_row"] == 1 and st1["waves_col"] == 1
    for x_chunks in (1, 4):
        o2, X2, Y2, sts = cases.run_shards_on_one_device(api, pa, X0, Y0, params, [0, 4000, m], [0, 10, n], x_chunks=x_chunks, tiled=1)
        assert np.array_equal(X1, X2) and np.array_equal(Y1, Y2) and np.array_equal(o1[1:], o2)
    mh = api.multi_create(pa, 3, device_ids=[0, 0, 0], tiled=1)
    try:
        X3, Y3 = np.array(X0, order="F"), np.array(Y0, order="F")
        o3, _ = api.multi_fit(mh, params, X3, Y3)
    finally:
        api.multi_destroy(mh)
    assert np.array_equal(X1, X3) and np.array_equal(Y1, Y3) and np.array_equal(o1[1:], o3[1:])
    assert o3[0] == pytest.approx(o1[0], rel=1e-12)
    against_oracle(pa, X0, Y0, params, 0, tiled=1)


def test_deferred_handle_refuses_to_step_before_finalize():
    rng = np.random.default_rng(3)
    pa, X0, Y0 = ragged_problem(rng, 60, 40, 8, 10)
    api = hip()
    h = api.create(cases.shard_of(pa, 0, 30, 0, 20), defer=True)
    try:
        with pytest.raises(_capi.GLRMError) as ei:
            api.step_x(h, 0.01)
        assert ei.value.code == _capi.ERR_INVALID and "glrm_hip_finalize" in ei.value.message
        small = _capi.CSignature()                       # smaller than the shard's own contribution: refused
        with pytest.raises(_capi.GLRMError):
            api.finalize(h, small)
        api.finalize(h, api.signature(h))
        with pytest.raises(_capi.GLRMError):
            api.finalize(h, None)                        # twice
        api.set_factors(h, X0, Y0)
        api.step_x(h, 0.01)
    finally:
        api.destroy(h)


# ------------------------------------------------------------------------------------------------ lists borrowed in place

@pytest.mark.parametrize("tiled,mix", [(1, 0), (2, 0), (2, 1)])
def test_borrowed_device_arrays_give_the_bits_of_copied_ones(tiled, mix):
    """GLRM_PROBLEM_BORROW_DEVICE_ARRAYS: the engine reads the caller's device arrays in place (what lets BASELINE configs[4], 120 GB of
    lists, run on one 288 GB device); views it has to reorder (mix = 1: the row view regrouped by loss kind) get a private copy.  Same
    bits as the copying hand-over, and the caller's arrays are untouched."""
    import torch
    from lowrankmodels.jl_amd import synth
    m, n, k, q = 4000, 1200, 32, 60
    w = synth.DeviceWorkload(m, n, k, q, loss_mix=mix)
    api = hip()
    params = L.ProxGradParams(max_iter=5)
    snap = [t.clone() for t in (w.rowptr, w.colidx, w.rowvals, w.colptr, w.rowidx, w.colvals)]
    out = []
    for borrow in (False, True):
        h = api.create(w.problem(borrow=borrow), tiled=tiled)
        ld = api.factor_ld(h)
        dX, dY = w.init_factors(ld)
        X = np.asfortranarray(0.3 * dX.cpu().numpy().reshape(m, ld)[:, :k].T)
        Y = np.asfortranarray(0.3 * dY.cpu().numpy().reshape(n, ld)[:, :k].T)
        obj, _ = api.fit(h, params, X, Y)
        st = api.kernel_stats(h)
        api.destroy(h)
        out.append((obj, X, Y, st["tiled"]))
        for a, b in zip(snap, (w.rowptr, w.colidx, w.rowvals, w.colptr, w.rowidx, w.colvals)):
            assert torch.equal(a, b)
    # bit0 / bit1: the LDS tiles on both sides; bit8 / bit9: in their lane-per-segment form (rank 32; rows only with ONE loss descriptor)
    assert out[0][3] == out[1][3] == ((3 | 512 | 256) if tiled == 2 else 0)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    with pytest.raises(_capi.GLRMError):   # host arrays cannot be borrowed
        pa, X0, Y0 = ragged_problem(np.random.default_rng(1), 30, 20, 4, 5)
        pa.flags = _capi.PROBLEM_BORROW_DEVICE_ARRAYS
        api.create(pa)


def test_persistent_cached_sweep_gives_the_bits_of_one_row_per_workgroup(monkeypatch):
    """regcached_persist_kernel (a workgroup walks many rows and hands the next row's list over through LDS while it works on the
    current one: C4 X half-step 85.7 -> 74.8 ms) against regcached_sweep_kernel (one row per workgroup): same lanes, same order, same
    bits -- uniform rows (13 trips at rank 64, 7 per wave), short rows (MAXT = 4), ragged rows with a class list, row chunks."""
    monkeypatch.setenv("GLRM_HIP_CACHED", "1")
    api = hip()
    params = L.ProxGradParams(max_iter=5)
    rng = np.random.default_rng(11)
    problems = [c4_problem(3000, 100000, 100)[:3], ragged_problem(rng, 500, 300, 64, 30), ragged_problem(rng, 600, 400, 64, 100, long_rows=(17, 333), long_len=150)]
    for pa, X0, Y0 in problems:
        out = []
        for persist in ("1", "0"):
            monkeypatch.setenv("GLRM_HIP_CACHED_PERSIST", persist)
            o, X, Y, st = cases.run_engine(api, pa, X0, Y0, params, tiled=0)
            assert st["tiled"] & CACHED
            out.append((o, X, Y, st["trials_x"], st["accepts_x"]))
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
        assert out[0][3:] == out[1][3:]
    monkeypatch.setenv("GLRM_HIP_CACHED_PERSIST", "1")
    pa, X0, Y0 = problems[2]
    o1, X1, Y1, _ = cases.run_engine(api, pa, X0, Y0, params, tiled=0)
    o2, X2, Y2, _ = cases.run_shards_on_one_device(api, pa, X0, Y0, params, [0, 250, 600], [0, 100, 400], x_chunks=3, tiled=0)
    assert np.array_equal(X1, X2) and np.array_equal(Y1, Y2) and np.array_equal(o1[1:], o2)


# ------------------------------------------------------------------------------------------------ LDS-tiled sweeps: rounds over the still-searching segments

@pytest.mark.parametrize("mixed", [False, True])
def test_lds_tiled_rounds_over_the_searching_segments_give_the_bits_of_the_in_kernel_search(monkeypatch, mixed):
    """Round 4: the LDS-tiled row sweep makes the gradient pass and the first trial in one kernel and runs every further trial as a round
    over the rows that are still searching (compacted: csrc/glrm_tiled.hpp TiledArgs::actlist_out); the column passes compact their trial
    rounds the same way.  Which workgroup slot a segment sits in changes no sum: factors, step sizes' effects and trial counts must equal
    the round-3 forms (GLRM_HIP_TILE_ROUNDS=0: the whole search inside the row kernel, column rounds over every workgroup that holds an
    active column) bit for bit -- from a start and a step size that make most segments reject several trials."""
    monkeypatch.setenv("GLRM_HIP_LANE", "0")   # the four-lane kernels (rank 32 would otherwise run the lane-per-segment passes, whose search always runs in rounds)
    m, n, k, q = 4000, 1500, 32, 150
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(m, n, k, q, value_model=0, loss_mix=1 if mixed else 0)
    if mixed:
        kinds = [L.QuadLoss().descriptor(), L.LogisticLoss().descriptor(), L.OrdinalHingeLoss(1, 5).descriptor()]
        losses = np.array([kinds[f % 3] for f in range(n)], dtype=_capi.LOSS_DTYPE)
    else:
        losses = np.array([(0, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(1, 0, 0.5)], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, losses, reg, reg)
    params = L.ProxGradParams(stepsize=40.0, max_iter=6, abs_tol=0.0, rel_tol=-1.0)   # alpha starts 40 x too large: every segment rejects several trials in its first half-steps
    res = {}
    for rounds in ("0", "3", "1", "2"):
        monkeypatch.setenv("GLRM_HIP_TILE_ROUNDS", rounds)
        res[rounds] = cases.run_engine(hip(), pa, X0, Y0, params, tiled=2)
        assert res[rounds][3]["tiled"] & 3 == 3
    base = res["0"]
    assert base[3]["trials_x"] > 2 * m * 6 and base[3]["trials_y"] > 2 * n * 6      # the search really ran for several rounds per segment
    for rounds in ("3", "1", "2"):
        r = res[rounds]
        assert np.array_equal(r[0], base[0]) and np.array_equal(r[1], base[1]) and np.array_equal(r[2], base[2]), rounds
        for key in ("trials_x", "trials_y", "accepts_x", "accepts_y"):
            assert r[3][key] == base[3][key], (rounds, key)
    O.set_threads(4)
    o_c, X_c, Y_c, st_c = cases.run_engine(O.oracle_api(), pa, X0, Y0, params)
    assert cases.rel_err(res["3"][0], o_c) < TOL and cases.fro_err(res["3"][1], X_c) < TOL and cases.fro_err(res["3"][2], Y_c) < TOL


@pytest.mark.parametrize("family", ["tiled", "blocked", "gather"])
def test_a_line_search_of_two_hundred_rounds_runs_to_its_end(monkeypatch, family):
    """fit!(glrm, ProxGradParams(1e30, min_stepsize=1e-3)): every segment shrinks its step ~200 times before a trial is accepted
    (src/algorithms/proxgrad.jl:136-155 has no bound on the trials of a line search).  The pass families run the search as host-driven
    rounds; until round 4 the loop stopped after 64 rounds and silently left the segments mid-search.  Trial counts and factors against the oracle."""
    if family == "blocked":
        force_blocked(monkeypatch)
    m, n, k, q = 3000, 1200, 32, 120
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(m, n, k, q, value_model=0)
    one = np.array([(0, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(0, 0, 1.0)], dtype=_capi.REG_DTYPE)   # ZeroReg: nothing shrinks the trial point, the step size has to come down by itself
    pa = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, one, reg, reg)
    params = L.ProxGradParams(stepsize=1e30, max_iter=2, abs_tol=0.0, rel_tol=-1.0, min_stepsize=1e-3)   # (the default minimum is stepsize / 100)
    O.set_threads(4)
    o_c, X_c, Y_c, st_c = cases.run_engine(O.oracle_api(), pa, X0, Y0, params)
    o_g, X_g, Y_g, st_g = cases.run_engine(hip(), pa, X0, Y0, params, tiled={"tiled": 2, "blocked": 1, "gather": 1}[family])
    want = {"tiled": 3, "blocked": BLOCKED_ROWS | BLOCKED_COLS, "gather": 0}[family]
    assert st_g["tiled"] & want == want
    assert st_c["trials_x"] > 150 * m and st_c["trials_y"] > 150 * n
    for key in ("trials_x", "trials_y", "accepts_x", "accepts_y"):
        assert abs(st_g[key] - st_c[key]) <= max(5, 1e-3 * st_c[key]), (key, st_g[key], st_c[key])
    assert cases.rel_err(o_g, o_c) < TOL and cases.fro_err(X_g, X_c) < TOL and cases.fro_err(Y_g, Y_c) < TOL


@pytest.mark.parametrize("family", ["tiled", "blocked", "gather"])
def test_a_step_size_that_lands_exactly_on_the_minimum_ends_the_search(monkeypatch, family):
    """`while alpha > min_stepsize` (src/algorithms/proxgrad.jl:136,180): a rejected trial that shrinks alpha to EXACTLY min_stepsize
    neither resets it to 1.1 min (that needs alpha < min) nor takes another trial.  stepsize 1.0, min_stepsize 0.7: 1.0 * 0.7 is the double
    0.7.  A = 0, X0 = 0: every gradient is exactly 0, every trial point equals the current point, J' == J rejects -- in any summation
    order.  Iteration 1: one trial per segment, alpha = 0.7; iteration 2: no trial at all.  (the round state machines of the
    pass families went on while `!(alpha < min)` and took a second trial, ending at alpha = 0.77.)"""
    if family == "blocked":
        force_blocked(monkeypatch)
    m, n, k, q = 3000, 1200, 32, 120
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(m, n, k, q, value_model=0)
    rowvals[:] = 0.0
    colvals[:] = 0.0
    X0 = np.zeros_like(X0)
    one = np.array([(0, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(0, 0, 1.0)], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, one, reg, reg)
    params = L.ProxGradParams(stepsize=1.0, max_iter=2, abs_tol=0.0, rel_tol=-1.0, min_stepsize=0.7)
    O.set_threads(4)
    o_c, X_c, Y_c, st_c = cases.run_engine(O.oracle_api(), pa, X0, Y0, params)
    assert (st_c["trials_x"], st_c["trials_y"], st_c["accepts_x"], st_c["accepts_y"]) == (m, n, 0, 0)
    o_g, X_g, Y_g, st_g = cases.run_engine(hip(), pa, X0, Y0, params, tiled={"tiled": 2, "blocked": 1, "gather": 1}[family])
    want = {"tiled": 3, "blocked": BLOCKED_ROWS | BLOCKED_COLS, "gather": 0}[family]
    assert st_g["tiled"] & want == want
    assert (st_g["trials_x"], st_g["trials_y"], st_g["accepts_x"], st_g["accepts_y"]) == (m, n, 0, 0)
    assert np.array_equal(o_g, o_c) and np.array_equal(X_g, X_c) and np.array_equal(Y_g, Y_c)


# ------------------------------------------------------------------------------------------------ the Y half-step while X is arriving

@pytest.mark.parametrize("dynamic", ["1", "0"])
def test_arrival_order_y_half_step_gives_the_bits_of_step_y(monkeypatch, dynamic):
    """glrm_hip_step_y_arrival (include/glrm_hip.h): the phase-aligned column passes launch each super-tile of X behind the events of the
    blocks it touches, in the order the host announces them -- own rows first, then the peers' chunks.  Partial sums are per (column,
    super-tile) and col_reduce adds them in super-tile order, so the result is glrm_hip_step_y's bit for bit whatever the order.  The
    events here fire late (a side stream that sleeps before each record): the launches really stand behind them.  Profile: the waits are
    accounted in kernel_stats.ms_wait_y.  Bad block lists are refused.
    dynamic = 1 (default): the host enqueues a super-tile once the events of all its blocks have FIRED (hipEventQuery) and polls for the
    rest -- true arrival order, the stream never stands behind a lagging block while other super-tiles are ready; 0: every super-tile is
    enqueued at once in the announced order behind in-stream waits."""
    import torch
    monkeypatch.setenv("GLRM_HIP_ARRIVAL_DYNAMIC", dynamic)
    force_blocked(monkeypatch)           # one tile unit per super-tile: 6000 rows of X = dozens of super-tiles
    pa, X0, Y0, _, _ = c4_problem(6000, 600, 100)
    api = hip()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream(device=dev)

    def run(arrival):
        h = api.create(pa, stream=stream, profile=1)
        try:
            assert api.kernel_stats(h)["tiled"] & BLOCKED_COLS
            api.set_factors(h, X0, Y0)
            api.reset_stepsizes(h, 1.0)
            objs = []
            for it in range(3):
                api.step_x(h, 0.01)
                if arrival:
                    # "own" rows 1500..3000 are there; the rest is announced in chunks of 750 rows, peers interleaved, by events the
                    # side stream records one after the other with a pause in front of each
                    chunks = [(lo, lo + 750) for lo in (0, 3000, 750, 3750, 4500, 5250)]
                    blocks, keep = [(1500, 3000, None)], []
                    with torch.cuda.stream(side):
                        side.wait_stream(torch.cuda.current_stream())
                        for lo, hi in chunks:
                            torch.cuda._sleep(400_000)         # a pause in front of every arrival (0.2 - 4 ms: the counter's rate differs between devices)
                            ev = torch.cuda.Event()
                            ev.record(side)
                            keep.append(ev)
                            blocks.append((lo, hi, ev.cuda_event))
                    if it == 1:
                        blocks = blocks[::-1]                    # any order the host likes
                    api.step_y_arrival(h, 0.01, blocks)
                else:
                    api.step_y(h, 0.01)
                ld = api.factor_ld(h)
                X, Y = np.zeros_like(X0), np.zeros_like(Y0)
                api.get_factors(h, X, Y)
                objs.append((X.copy(), Y.copy()))
            st = api.kernel_stats(h)
            if arrival:  # refusals: a gap, an overlap, rows out of range
                for bad in ([(0, 3000, None), (3500, 6000, None)], [(0, 3500, None), (3000, 6000, None)], [(0, 7000, None)], [(0, 3000, None)]):
                    with pytest.raises(_capi.GLRMError) as ei:
                        api.step_y_arrival(h, 0.01, bad)
                    assert ei.value.code == _capi.ERR_INVALID
        finally:
            api.destroy(h)
        return objs, st

    plain, st0 = run(False)
    arr, st1 = run(True)
    for (Xa, Ya), (Xp, Yp) in zip(arr, plain):
        assert np.array_equal(Xa, Xp) and np.array_equal(Ya, Yp)
    for key in ("trials_x", "trials_y", "accepts_x", "accepts_y"):
        assert st1[key] == st0[key]
    assert st0["ms_wait_y"] == 0.0
    if dynamic == "0":
        assert st1["ms_wait_y"] > 0.0      # the launch stream did stand in front of late blocks (most of the pauses hide behind the super-tiles already there)


@pytest.mark.parametrize("family", ["gather", "tiled-four-lane", "tiled-lane"])
def test_arrival_order_on_the_other_column_families(monkeypatch, family):
    """Beyond the phase-aligned passes.  The gather sweep walks X in one kernel: step_y_arrival waits for all events and runs step_y.  The
    LDS-tiled column passes and their lane-per-segment form (round 6) launch their gradient pass in RUNS of super-tiles,
    each behind the in-stream waits for the blocks of X it reads, in the order the host announces them; partial sums are per (column,
    super-tile) and col_reduce adds them in super-tile order, so the bits are glrm_hip_step_y's whatever the order -- with events that fire late
    (a side stream that sleeps in front of each record), in two announced orders, over three iterations."""
    import torch
    k = 64 if family == "gather" else 32
    if family == "tiled-four-lane":
        monkeypatch.setenv("GLRM_HIP_LANE", "0")
    monkeypatch.setenv("GLRM_HIP_COL_WORKGROUPS", "4096")   # many super-tiles on this small problem
    m, n = 6000, 600
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(m, n, k, 100, value_model=0)
    one = np.array([(0, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(1, 0, 0.5)], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, one, reg, reg)
    api = hip()
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    res = {}
    for arrival in (False, True):
        h = api.create(pa, stream=torch.cuda.current_stream().cuda_stream, tiled=1 if family == "gather" else 2, profile=1)
        try:
            flags = api.kernel_stats(h)["tiled"]
            assert not flags & BLOCKED_COLS and bool(flags & 2) == (family != "gather") and bool(flags & 512) == (family == "tiled-lane")
            api.set_factors(h, X0, Y0)
            api.reset_stepsizes(h, 1.0)
            for it in range(3):
                api.step_x(h, 0.01)
                if arrival:
                    chunks = [(lo, lo + 750) for lo in (0, 3000, 750, 3750, 4500, 5250)]
                    blocks, keep = [(1500, 3000, None)], []
                    with torch.cuda.stream(side):
                        side.wait_stream(torch.cuda.current_stream())
                        for lo, hi in chunks:
                            torch.cuda._sleep(400_000)
                            ev = torch.cuda.Event()
                            ev.record(side)
                            keep.append(ev)
                            blocks.append((lo, hi, ev.cuda_event))
                    if it == 1:
                        blocks = blocks[::-1]
                    api.step_y_arrival(h, 0.01, blocks)
                else:
                    api.step_y(h, 0.01)
            X, Y = np.zeros_like(X0), np.zeros_like(Y0)
            api.get_factors(h, X, Y)
            res[arrival] = (X, Y, api.kernel_stats(h))
        finally:
            api.destroy(h)
    assert np.array_equal(res[True][0], res[False][0]) and np.array_equal(res[True][1], res[False][1])
    for key in ("trials_x", "trials_y", "accepts_x", "accepts_y"):
        assert res[True][2][key] == res[False][2][key]
    assert res[False][2]["ms_wait_y"] == 0.0 and res[True][2]["ms_wait_y"] > 0.0   # the launch stream did stand in front of late blocks


@pytest.mark.parametrize("mixed", [False, True])
def test_lane_trial_rounds_give_the_same_bits_in_every_form(monkeypatch, mixed):
    """The trial rounds of the lane-per-segment passes (csrc/glrm_lane.hip: glrm_run_lane) pick one of three forms of the same kernel by the
    fraction of segments that still searches: the SELL layout over the full grid, the CSR form over the compact list, and (session r6_25) the
    layout read by gathered waves -- wave lists built chunk by chunk, or packed from the decide kernel's list for the tail rounds.  Every form
    adds the same terms in the same order: forcing each of them on all rounds (thresholds through the environment) gives the bits of the
    default mix, with the same number of trials and accepts.  Rows of 300 observations started far from a minimum search for several rounds
    (start step sizes of 64: every row rejects a few trials, then the searches thin out); QuadLoss and a loss per column (descriptor ids in
    the stream's offset word, rows and columns)."""
    m, n, k = 9000, 1500, 32
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(m, n, k, 300, value_model=0, loss_mix=1 if mixed else 0)
    if mixed:
        kinds = [L.QuadLoss(0.8).descriptor(), L.HuberLoss(1.1, crossover=0.7).descriptor(), L.OrdinalHingeLoss(1, 5, 0.9).descriptor()]
        losses = np.array([kinds[f % 3] for f in range(n)], dtype=_capi.LOSS_DTYPE)
    else:
        losses = np.array([(0, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(1, 0, 0.5)], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, losses, reg, reg)
    api = hip()
    forms = {
        "default": {},
        "older forms (full grid / CSR)": {"GLRM_HIP_LANE_ROUNDS": "0"},
        "gathered, chunk lists": {"GLRM_HIP_LANE_GATHER_TO": "101", "GLRM_HIP_LANE_GATHER_PACKED": "0"},
        "gathered, packed lists": {"GLRM_HIP_LANE_GATHER_TO": "101", "GLRM_HIP_LANE_GATHER_PACKED": "101", "GLRM_HIP_LANE_GATHER_SPREAD": "1000"},
        # the COMPACT form of the stream (2-byte offsets padded, values unpadded: what views beyond 2e9 observations run) on every side it
        # serves -- both views of the QuadLoss model, the column view of the model with a loss per column; its rounds: full grid / CSR
        "compact stream": {"GLRM_HIP_LANE_COMPACT": "1"},
        # every round after the first (and a first trial of few rows) on the tail kernel: a wave per (segment, super-tile), no tile
        "a wave per row": {"GLRM_HIP_LANE_TAIL": "101", "GLRM_HIP_LANE_TAIL_COLS": "101"},
        "no tail kernel": {"GLRM_HIP_LANE_TAIL": "0", "GLRM_HIP_LANE_TAIL_COLS": "0"},
    }
    res = {}
    for name, env in forms.items():
        for key in ("GLRM_HIP_LANE_ROUNDS", "GLRM_HIP_LANE_GATHER_TO", "GLRM_HIP_LANE_GATHER_PACKED", "GLRM_HIP_LANE_GATHER_SPREAD", "GLRM_HIP_LANE_COMPACT", "GLRM_HIP_LANE_TAIL", "GLRM_HIP_LANE_TAIL_COLS"):
            monkeypatch.delenv(key, raising=False)
        for key, v in env.items():
            monkeypatch.setenv(key, v)
        h = api.create(pa, tiled=2)
        try:
            assert api.kernel_stats(h)["tiled"] & (256 | 512) == (256 | 512)
            api.set_factors(h, 0.3 * X0, 0.3 * Y0)
            api.reset_stepsizes(h, 64.0)
            for it in range(4):
                api.step_x(h, 0.01)
                api.step_y(h, 0.01)
            X, Y = np.zeros_like(X0), np.zeros_like(Y0)
            api.get_factors(h, X, Y)
            res[name] = (X, Y, api.kernel_stats(h))
        finally:
            api.destroy(h)
    X, Y, st = res["default"]
    assert st["trials_x"] > 2 * 4 * m and st["trials_y"] > 4 * n   # the searches did run for several rounds
    for name, (Xf, Yf, stf) in res.items():
        assert np.array_equal(Xf, X) and np.array_equal(Yf, Y), name
        for key in ("trials_x", "trials_y", "accepts_x", "accepts_y"):
            assert stf[key] == st[key], (name, key)


@pytest.mark.parametrize("lane", ["0", "3"])
def test_arrival_order_with_one_super_tile_waits_for_every_block(monkeypatch, lane):
    """A problem of at most one tile of rows (m <= 560 at rank 32) has ONE super-tile, which reads every row of X: under
    glrm_hip_step_y_arrival its gradient pass must stand behind ALL announced blocks.  (Session r6_69: it was launched without any wait --
    found by tests/perf/soak_lane_shards.py, seed 5004: 345 rows on six shards raced with the exchange, in two runs of ten.)  Here the
    announced blocks really ARRIVE late: their rows of the bound X buffer are zeroed on the launch stream and restored by a side stream that
    sleeps first and records the block's event afterwards -- a pass that does not wait reads zeros and lands on other bits than glrm_hip_step_y."""
    import torch
    monkeypatch.setenv("GLRM_HIP_LANE", lane)
    m, n, k = 400, 600, 32
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(m, n, k, 60, value_model=0)
    one = np.array([(0, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(1, 0, 0.5)], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, one, reg, reg)
    api = hip()
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    res = {}
    for arrival in (False, True):
        h = api.create(pa, stream=torch.cuda.current_stream().cuda_stream, tiled=2, profile=1)
        try:
            assert api.kernel_stats(h)["tiled"] & 2
            ld = api.factor_ld(h)
            dX, dY = torch.zeros(m * ld, dtype=torch.float64, device=dev), torch.zeros(n * ld, dtype=torch.float64, device=dev)
            dX.view(m, ld)[:, :k] = torch.as_tensor(np.ascontiguousarray(X0.T), device=dev)
            dY.view(n, ld)[:, :k] = torch.as_tensor(np.ascontiguousarray(Y0.T), device=dev)
            dC, dR = torch.zeros(n, dtype=torch.float64, device=dev), torch.zeros(m, dtype=torch.float64, device=dev)
            api.bind_buffers(h, dX.data_ptr(), dY.data_ptr(), dC.data_ptr(), dR.data_ptr())
            api.reset_stepsizes(h, 1.0)
            for it in range(2):
                api.step_x(h, 0.01)
                if arrival:
                    blocks, keep = [(100, 200, None)], []
                    late = ((0, 100), (200, 300), (300, 400))
                    Xc = dX.clone()
                    for lo, hi in late:
                        dX.view(m, ld)[lo:hi] = 0.0          # (on the launch stream: the rows are not there yet)
                    with torch.cuda.stream(side):
                        side.wait_stream(torch.cuda.current_stream())
                        for lo, hi in late:
                            torch.cuda._sleep(2_000_000)
                            dX.view(m, ld)[lo:hi] = Xc.view(m, ld)[lo:hi]
                            ev = torch.cuda.Event()
                            ev.record(side)
                            keep.append(ev)
                            blocks.append((lo, hi, ev.cuda_event))
                    api.step_y_arrival(h, 0.01, blocks)
                    torch.cuda.current_stream().wait_stream(side)
                else:
                    api.step_y(h, 0.01)
            torch.cuda.synchronize()
            res[arrival] = (dX.cpu().numpy().copy(), dY.cpu().numpy().copy(), api.kernel_stats(h))
        finally:
            api.destroy(h)
    assert np.array_equal(res[True][0], res[False][0]) and np.array_equal(res[True][1], res[False][1])
    assert res[False][2]["ms_wait_y"] == 0.0 and res[True][2]["ms_wait_y"] > 0.0
