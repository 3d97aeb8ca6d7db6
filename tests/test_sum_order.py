"""The oracle's configurable summation order ("make the oracle's summation order configurable ... so forks
can be attributed"; include/glrm_hip.h: glrm_sum_order, oracle/glrm_oracle.c: eng_pass / eng_step).

CPU part (this file): the C restatement of the engine's orders is checked against a literal lane-by-lane simulation of the kernels'
control flow (tests/lane_orders.py), bit for bit; the degenerate layout reproduces the reference order bit for bit; every order stays
within rounding of the reference order on trajectories that do not fork.  The GPU part (tests/test_gpu_sum_order.py) closes the loop:
the HIP engine reports its order and the oracle in that order reproduces the engine's factors bit for bit.
Reference lines whose order is at stake: src/algorithms/proxgrad.jl:122-132,143,165-175,187; src/evaluate_fit.jl:24-55."""
import numpy as np
import pytest

import lane_orders as LO
import lowrankmodels.jl_amd as L
import oracle as O
from lowrankmodels.jl_amd import _capi

QUAD = np.array([(0, 0, 0.75, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)


def small_problem(m, n, k, lens_rows, seed, reg=(0, 0, 1.0), sorted_lists=True):
    rng = np.random.default_rng(seed)
    rows = []
    for e in range(m):
        c = rng.integers(0, n, lens_rows[e % len(lens_rows)])
        rows.append(np.sort(c) if sorted_lists else c)
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    colidx = np.concatenate(rows).astype(np.int32)
    rowvals = rng.standard_normal(len(colidx))
    # the column view lists the same entries column by column, rows ascending (duplicates kept)
    order = np.lexsort((np.repeat(np.arange(m), np.diff(rowptr)), colidx))
    ri = np.repeat(np.arange(m), np.diff(rowptr))[order].astype(np.int32)
    cj = colidx[order]
    colptr = np.concatenate([[0], np.cumsum(np.bincount(cj, minlength=n))]).astype(np.int64)
    colvals = rowvals[order]
    r = np.array([reg], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, ri, colvals, QUAD, r, r)
    X0 = np.asfortranarray(rng.standard_normal((k, m)) * 0.5)
    Y0 = np.asfortranarray(rng.standard_normal((k, n)) * 0.5)
    return pa, X0, Y0


def oracle_half_steps(pa, X0, Y0, order_r, order_c, alpha0=1.0):
    api = O.oracle_api()
    h = api.create(pa)
    try:
        O.set_sum_order(h, 0, order_r)
        O.set_sum_order(h, 1, order_c)
        api.set_factors(h, X0, Y0)
        api.reset_stepsizes(h, alpha0)
        api.step_x(h, 0.01)
        X1, Y1 = np.zeros_like(X0), np.zeros_like(Y0)
        api.get_factors(h, X1, Y1)
        api.step_y(h, 0.01)
        X2, Y2 = np.zeros_like(X0), np.zeros_like(Y0)
        api.get_factors(h, X2, Y2)
        st = api.kernel_stats(h)
    finally:
        api.destroy(h)
    return X1, Y2, st


def simulate(pa, X0, Y0, rows, passfn_of, reg, G, R, alpha0=1.0):
    """The half-step of every segment of one view through tests/lane_orders.py."""
    k = pa.k
    ptr, idx, vals = (pa.rowptr, pa.colidx, pa.rowvals) if rows else (pa.colptr, pa.rowidx, pa.colvals)
    own, fac = (X0, Y0) if rows else (Y0, X0)
    facl = [list(fac[:, i]) for i in range(fac.shape[1])]
    out = own.copy(order="F")
    trials = 0
    for s in range(len(ptr) - 1):
        b, e = int(ptr[s]), int(ptr[s + 1])
        ix, vv = [int(v) for v in idx[b:e]], [float(v) for v in vals[b:e]]
        passfn = passfn_of(s, ix, vv, facl)
        if reg[0] == 1:
            regfn = lambda x: LO.reg_quad(reg[2], x, k, G, R)  # noqa: E731
            proxfn = lambda x, a: [1 / (1 + 2 * a * reg[2]) * v for v in x]  # noqa: E731
        else:
            regfn = lambda x: 0.0  # noqa: E731
            proxfn = lambda x, a: x  # noqa: E731
        xn, _, _, t = LO.half_step(passfn, regfn, proxfn, [float(v) for v in own[:, s]], alpha0, e - b)
        out[:, s] = xn
        trials += t
    return out, trials


@pytest.mark.parametrize("k,G,R,waves,cached", [(10, 4, 4, 1, False), (64, 8, 8, 0, True), (20, 4, 8, 4, False)])
def test_strided_order_equals_the_lane_simulation_of_the_gather_and_cached_sweeps(k, G, R, waves, cached):
    """sweep_pass / block_combine and reg_pass / row_combine: group q of W x 64 / G takes q, q + T, ...; butterfly over the groups of a
    wave; waves in order.  cached: rows of <= 13 trips on two waves (regcached_sweep_kernel), the 40-observation rows of this problem."""
    lens = [3, 17, 40, 150] if not cached else [5, 40, 104, 130]
    pa, X0, Y0 = small_problem(8, 60, k, lens, seed=k, reg=(1, 0, 0.3))
    o = O.make_sum_order("strided", G, R, waves=waves, cached_maxlen=13 * (64 // G) if cached else -1, cached_waves=2 if cached else 0)
    oc = O.make_sum_order("strided", G, R, waves=waves)
    X1, Y2, st = oracle_half_steps(pa, X0, Y0, o, oc)

    def waves_of(n, rows):
        if rows and cached and n <= 13 * (64 // G):
            return 2
        return waves or (1 if n < 1536 else 4)

    def pf_rows(s, ix, vv, facl):
        return lambda x, grad: LO.strided_pass(ix, vv, x, facl, k, G, R, waves_of(len(ix), True), 0.75, grad)
    Xs, tx = simulate(pa, X0, Y0, True, pf_rows, (1, 0, 0.3), G, R)
    assert np.array_equal(Xs, X1)

    def pf_cols(s, ix, vv, facl):
        return lambda x, grad: LO.strided_pass(ix, vv, x, facl, k, G, R, waves_of(len(ix), False), 0.75, grad)
    Ys, ty = simulate(pa, Xs, Y0, False, pf_cols, (1, 0, 0.3), G, R)
    assert np.array_equal(Ys, Y2)
    assert (tx, ty) == (st["trials_x"], st["trials_y"])


def test_strided_order_with_four_loss_partials_per_group():
    """sweep_pass SCATTER (G = 4, a non-uniform loss table): the group's loss sum is four lane partials (observation u of each trip of 4)."""
    k, G, R = 12, 4, 4
    pa, X0, Y0 = small_problem(6, 50, k, [9, 30, 70], seed=5)
    o = O.make_sum_order("strided", G, R, waves=1, batch=4)
    X1, _, _ = oracle_half_steps(pa, X0, Y0, o, o)

    def pf(s, ix, vv, facl):
        return lambda x, grad: LO.strided_pass(ix, vv, x, facl, k, G, R, 1, 0.75, grad, scatter=True)
    Xs, _ = simulate(pa, X0, Y0, True, pf, (0, 0, 1.0), G, R)
    assert np.array_equal(Xs, X1)


@pytest.mark.parametrize("k,G,R,tile,tps,four,rot", [(10, 4, 4, 16, 2, False, 0), (64, 8, 8, 7, 3, False, 0), (30, 4, 8, 5, 0, True, 0),
                                                     (32, 4, 8, 16, 2, True, 1), (60, 8, 8, 9, 1, True, 1)])
def test_windowed_order_equals_the_lane_simulation_of_the_tiled_passes(k, G, R, tile, tps, four, rot):
    """tiled_pass with its batches of G entries, the tile bound, re-anchoring and the parity / lane partial sums of the losses; rows in
    one pass over all tiles, columns in super-tiles of `tps` tiles added in order; `rot`: the chunk walk i ^ ((column & 7) >> 1)."""
    pa, X0, Y0 = small_problem(12, 70, k, [1, 2, 8, 23, 64, 65], seed=100 + k, reg=(1, 0, 0.2))
    orow = O.make_sum_order("windowed", G, R, window=tile, windows_per_sup=0, batch=G if four else 2)
    ocol = O.make_sum_order("windowed", G, R, window=tile, windows_per_sup=tps, batch=G if four else 2, rotate=rot)
    X1, Y2, st = oracle_half_steps(pa, X0, Y0, orow, ocol)

    def pf_rows(s, ix, vv, facl):
        return lambda x, grad: LO.windowed_pass(ix, vv, x, facl, k, G, R, tile, 0, pa.n, 0.75, grad, four=four)
    Xs, tx = simulate(pa, X0, Y0, True, pf_rows, (1, 0, 0.2), G, R)
    assert np.array_equal(Xs, X1)

    def pf_cols(s, ix, vv, facl):
        r_ = ((s & 7) >> 1) if rot else 0
        return lambda x, grad: LO.windowed_pass(ix, vv, x, facl, k, G, R, tile, tps, pa.m, 0.75, grad, four=four, rot=r_)
    Ys, ty = simulate(pa, Xs, Y0, False, pf_cols, (1, 0, 0.2), G, R)
    assert np.array_equal(Ys, Y2)
    assert (tx, ty) == (st["trials_x"], st["trials_y"])


@pytest.mark.parametrize("k", [32, 27])
def test_lane_per_segment_order_equals_the_simulation_of_the_lane_kernel(k):
    """csrc/glrm_lane.hpp (round 6): one lane per segment, chunk walk i ^ (gseg & 15), two fma chains over the even / odd registers.  The engine
    reports that family as WINDOWED with lanes = 2, comps = 16, batch = 2, rotate = 2 -- the two chains ARE the two lanes of that layout, each
    walking its own chunks in the order i ^ ((gseg >> 1) & 7) -- and the oracle's restatement of that order must equal a literal simulation of
    the kernel bit for bit (rows: one super-tile; columns: super-tiles of `tps` tiles; the regularizer sums in the two-lane layout of
    col_reduce / col_decide<2, 16>)."""
    tile, tps = 7, 3
    pa, X0, Y0 = small_problem(40, 70, k, [1, 2, 8, 23, 64, 65], seed=300 + k, reg=(1, 0, 0.2))
    orow = O.make_sum_order("windowed", 2, 16, window=tile, windows_per_sup=0, batch=2, rotate=2)
    ocol = O.make_sum_order("windowed", 2, 16, window=tile, windows_per_sup=tps, batch=2, rotate=2)
    X1, Y2, st = oracle_half_steps(pa, X0, Y0, orow, ocol)

    def pf_rows(s, ix, vv, facl):
        return lambda x, grad: LO.lane_kernel_pass(ix, vv, x, facl, k, tile, 0, pa.n, 0.75, grad, s)
    Xs, tx = simulate(pa, X0, Y0, True, pf_rows, (1, 0, 0.2), 2, 16)
    assert np.array_equal(Xs, X1)

    def pf_cols(s, ix, vv, facl):
        return lambda x, grad: LO.lane_kernel_pass(ix, vv, x, facl, k, tile, tps, pa.m, 0.75, grad, s)
    Ys, ty = simulate(pa, Xs, Y0, False, pf_cols, (1, 0, 0.2), 2, 16)
    assert np.array_equal(Ys, Y2)
    assert (tx, ty) == (st["trials_x"], st["trials_y"])
    # ... and it is a different order from the four-lane kernels' (the reason the family choice comes from the whole problem's signature)
    X4, _, _ = oracle_half_steps(pa, X0, Y0, O.make_sum_order("windowed", 4, 8, window=tile, batch=2), ocol)
    assert not np.array_equal(X4, X1) and np.abs(X4 - X1).max() < 1e-12


def test_windowed_order_of_the_phase_aligned_passes():
    """tiled_pass<..., L2 = true>: the super-tile is one window walked in a single go (glrm_blocked.hip)."""
    k, G, R, unit, tps = 64, 8, 8, 4, 3
    pa, X0, Y0 = small_problem(10, 90, k, [2, 11, 37, 80], seed=9)
    o = O.make_sum_order("windowed", G, R, window=unit * tps, windows_per_sup=1)
    X1, Y2, _ = oracle_half_steps(pa, X0, Y0, o, o)

    def pf(n_other):
        def of(s, ix, vv, facl):
            return lambda x, grad: LO.windowed_pass(ix, vv, x, facl, k, G, R, unit, tps, n_other, 0.75, grad, L2=True)
        return of
    Xs, _ = simulate(pa, X0, Y0, True, pf(pa.n), (0, 0, 1.0), G, R)
    assert np.array_equal(Xs, X1)
    Ys, _ = simulate(pa, Xs, Y0, False, pf(pa.m), (0, 0, 1.0), G, R)
    assert np.array_equal(Ys, Y2)


def fit_in_order(pa, X0, Y0, orders, iters):
    api = O.oracle_api()
    h = api.create(pa)
    try:
        for w, o in enumerate(orders):
            O.set_sum_order(h, w, o)
        X, Y = X0.copy(order="F"), Y0.copy(order="F")
        obj, _ = api.fit(h, L.ProxGradParams(max_iter=iters, abs_tol=-1e300, rel_tol=-1e300), X, Y)
    finally:
        api.destroy(h)
    return obj, X, Y


def test_one_lane_layout_is_the_reference_order_and_engine_orders_stay_within_rounding():
    """lanes = 1, one accumulator, one window: list order with a single chain per sum = the reference's order (bit for bit, ZeroReg /
    NonNegConstraint have no rounding of their own).  The engine orders are re-orderings of the same sums: 1e-13 on a short trajectory."""
    m, n, k, q = 1500, 400, 64, 40
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(m, n, k, q, value_model=1)
    nonneg = np.array([(3, 0, 1.0)], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, QUAD, nonneg, nonneg)
    X0, Y0 = np.asfortranarray(np.abs(X0) / 8), np.asfortranarray(np.abs(Y0) / 8)
    O.set_threads(4)
    ref = fit_in_order(pa, X0, Y0, [None, None], 15)
    deg = O.make_sum_order("windowed", 1, 64, window=1 << 40, windows_per_sup=0, batch=1)
    one = fit_in_order(pa, X0, Y0, [deg, deg], 15)
    assert np.array_equal(one[0], ref[0]) and np.array_equal(one[1], ref[1]) and np.array_equal(one[2], ref[2])
    strided = [O.make_sum_order("strided", 8, 8, cached_maxlen=104, cached_waves=2), O.make_sum_order("strided", 8, 8)]
    windowed = [O.make_sum_order("windowed", 8, 8, window=288), O.make_sum_order("windowed", 8, 8, window=288, windows_per_sup=2)]
    for orders in (strided, windowed):
        got = fit_in_order(pa, X0, Y0, orders, 15)
        assert not np.array_equal(got[1], ref[1])                       # the switch does change the last bits ...
        assert np.max(np.abs(got[0] - ref[0]) / ref[0]) < 1e-13         # ... and nothing else
        assert np.abs(got[1] - ref[1]).max() < 1e-12 and np.abs(got[2] - ref[2]).max() < 1e-12


def test_sum_order_is_refused_where_the_oracle_cannot_follow():
    pa, X0, Y0 = small_problem(5, 30, 8, [6], seed=1, sorted_lists=False)
    api = O.oracle_api()
    h = api.create(pa)
    try:
        assert api.sum_order(h, 0).asdict()["family_name"] == "reference"
        with pytest.raises(_capi.GLRMError):  # a windowed walk needs window-ordered lists (the engine checks the same at create)
            O.set_sum_order(h, 0, O.make_sum_order("windowed", 4, 2, window=4))
        with pytest.raises(_capi.GLRMError):
            O.set_sum_order(h, 0, O.make_sum_order("strided", 4, 1))      # 4 x 1 does not hold rank 8
        o = O.make_sum_order("strided", 4, 2)
        o.private_order = 1
        with pytest.raises(_capi.GLRMError):
            O.set_sum_order(h, 0, o)
        O.set_sum_order(h, 0, O.make_sum_order("strided", 4, 2))
        # (64 / lanes) x waves lane groups per segment must fit the oracle's 128-group buffers (two lanes on eight waves
        # smashed the stack); the same layouts pinned to few enough waves are fine
        for lanes, comps in ((2, 4), (1, 8)):
            with pytest.raises(_capi.GLRMError):
                O.set_sum_order(h, 0, O.make_sum_order("strided", lanes, comps))
            with pytest.raises(_capi.GLRMError):
                O.set_sum_order(h, 0, O.make_sum_order("strided", lanes, comps, waves=8))
        O.set_sum_order(h, 0, O.make_sum_order("strided", 2, 4, waves=4))
        O.set_sum_order(h, 0, O.make_sum_order("strided", 1, 8, waves=2))
        O.set_sum_order(h, 0, O.make_sum_order("strided", 2, 4, waves=2, cached_maxlen=10, cached_waves=4))
        with pytest.raises(_capi.GLRMError):
            O.set_sum_order(h, 0, O.make_sum_order("strided", 1, 8, waves=2, cached_maxlen=10, cached_waves=4))
        assert api.sum_order(h, 0).asdict()["family_name"] == "strided" and api.sum_order(h, 1).asdict()["family_name"] == "reference"
        O.set_sum_order(h, 0, None)
        assert api.sum_order(h, 0).asdict()["family_name"] == "reference"
    finally:
        api.destroy(h)


def test_kind_grouped_rows_equal_a_host_regrouped_row_view():
    """glrm_sum_order.private_order = 2 (rows of a model with several loss kinds on the LDS tiles): inside every window the entries are
    grouped by ascending loss kind, stably (csrc/glrm_tiled.hpp: group_rows_by_kind_kernel).  The oracle's restatement of that grouping
    must give the bits of the plain windowed order on a row view regrouped HERE, with numpy, before the hand-over."""
    pa, X0, Y0 = small_problem(40, 96, 8, [30, 17, 44, 9], seed=11, reg=(1, 0, 0.5))
    n, win = pa.n, 16
    kinds = np.array([0, 7, 6, 1])                                    # Quad, Logistic, OrdinalHinge, L1 by column (f mod 4)
    lt = np.zeros(n, dtype=_capi.LOSS_DTYPE)
    for f in range(n):
        lt[f] = (kinds[f % 4], 0, 1.0 + 0.25 * (f % 3), 1.0, 5.0)
    vals = pa.rowvals.copy()
    cls = lt["kind"][pa.colidx]
    vals[cls == 7] = (vals[cls == 7] > 0).astype(np.float64)         # Logistic labels
    vals[cls == 6] = np.clip(np.round(3 + vals[cls == 6]), 1, 5)      # OrdinalHinge levels
    order = np.lexsort((np.repeat(np.arange(pa.m), np.diff(pa.rowptr)), pa.colidx))
    mixed = _capi.ProblemArrays(pa.m, n, pa.k, pa.rowptr, pa.colidx, vals, pa.colptr, pa.rowidx, vals[order], lt, pa.rx, pa.ry)
    # the same grouping on the host: stable sort of every row by (window, kind)
    gi, gv = pa.colidx.copy(), vals.copy()
    for e in range(pa.m):
        b, en = int(pa.rowptr[e]), int(pa.rowptr[e + 1])
        key = np.lexsort((lt["kind"][pa.colidx[b:en]], pa.colidx[b:en] // win))
        gi[b:en], gv[b:en] = pa.colidx[b:en][key], vals[b:en][key]
    assert not np.array_equal(gi, pa.colidx)
    grouped = _capi.ProblemArrays(pa.m, n, pa.k, pa.rowptr, gi, gv, pa.colptr, pa.rowidx, vals[order], lt, pa.rx, pa.ry)
    o_plain = O.make_sum_order("windowed", 4, 2, window=win, batch=4)
    o_priv = O.make_sum_order("windowed", 4, 2, window=win, batch=4)
    o_priv.private_order = 2
    oc = O.make_sum_order("windowed", 4, 2, window=win, windows_per_sup=2, batch=4)
    Xa, Ya, sa = oracle_half_steps(mixed, X0, Y0, o_priv, oc)
    Xb, Yb, sb = oracle_half_steps(grouped, X0, Y0, o_plain, oc)
    assert np.array_equal(Xa, Xb) and np.array_equal(Ya, Yb) and sa["trials_x"] == sb["trials_x"]
    Xc, _, _ = oracle_half_steps(mixed, X0, Y0, o_plain, oc)          # ungrouped walk: same sums up to rounding, not the same bits
    assert np.abs(Xa - Xc).max() < 1e-9
