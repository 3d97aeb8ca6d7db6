"""-m gpu: power-law Omega (real observation patterns are heavy-tailed --
/root/reference/test/hello_world.jl:48-50 samples its entries with replacement -- while every BASELINE recipe holds exactly q observations
per row).  lowrankmodels.jl_amd/synth.py: ZipfWorkload draws row degrees and column popularities from a Zipf law; here every sweep
family runs such a problem against the oracle: gather sweeps (rows and columns spread over the 1 / 4 / 8-wave classes by their own
length), LDS-tiled sweeps, phase-aligned passes, the cached row sweep (short rows in registers, long rows on the gather sweep of the same
half-step), the reference-order mode; and two ragged shards against one handle, bit for bit."""
import numpy as np
import pytest
import torch

import cases
import lowrankmodels.jl_amd as L
import oracle as O
from lowrankmodels.jl_amd import _capi, synth

pytestmark = pytest.mark.gpu
TOL = 1e-5


def problem(m=30000, n=3000, k=32, nnz=3_000_000, s=0.8, nonneg=False, seed=5):
    reg = (3, 0, 1.0) if nonneg else (1, 0, 0.5)
    w = synth.ZipfWorkload(m, n, k, nnz, s_rows=s, s_cols=s, seed=seed, value_model=1 if nonneg else 0, rx=reg, ry=reg, chunk=1 << 20)
    pa = w.host_problem()
    X0, Y0 = w.init_factors(k)
    X0, Y0 = X0.numpy().reshape(m, k).T, Y0.numpy().reshape(n, k).T
    if nonneg:
        X0, Y0 = np.abs(X0) / k ** 0.5, np.abs(Y0) / k ** 0.5
    return w, pa, np.asfortranarray(X0), np.asfortranarray(Y0)


FAMILIES = {  # name: (environment, create kwargs, kernel_stats.tiled bits that must be set)
    "gather": ({"GLRM_HIP_CACHED": "0"}, {"tiled": 1}, 0),
    "tiled": ({}, {"tiled": 2}, 3),
    "blocked": ({"GLRM_HIP_BLOCKED": "3", "GLRM_HIP_BLOCKED_TPS": "2", "GLRM_HIP_BLOCKED_FILL": "5", "GLRM_HIP_CACHED": "0"}, {}, 16 | 32),
    "cached": ({"GLRM_HIP_CACHED": "1"}, {"tiled": 1}, 64),
    "reference-order": ({}, {"sum_order": 1}, 128),
}


@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_every_family_on_a_power_law_omega_against_the_oracle(monkeypatch, family):
    w, pa, X0, Y0 = problem(nonneg=family in ("blocked", "cached"))
    d = w.degree_summary()
    assert d["rows"]["max"] >= 6 * d["rows"]["median"] and d["cols"]["max"] > 20 * d["cols"]["median"], d   # heavy tails on both sides
    assert d["cols"]["max"] >= 1536                                                                        # beyond the one-wave class
    env, kw, bits = FAMILIES[family]
    for k_, v in env.items():
        monkeypatch.setenv(k_, v)
    params = L.ProxGradParams(max_iter=6, abs_tol=0.0, rel_tol=-1.0)
    O.set_threads(4)
    o_c, X_c, Y_c, st_c = cases.run_engine(O.oracle_api(), pa, X0, Y0, params)
    o_g, X_g, Y_g, st_g = cases.run_engine(_capi.hip_api(), pa, X0, Y0, params, **kw)
    assert st_g["tiled"] & bits == bits, (family, st_g["tiled"])
    assert st_g["nnz_rows"] == st_c["nnz_rows"] == w.nnz_rows
    e = (cases.rel_err(o_g, o_c), cases.fro_err(X_g, X_c), cases.fro_err(Y_g, Y_c))
    assert max(e) < (1e-12 if family == "reference-order" else TOL), (family, e)
    for key in ("trials_x", "trials_y", "accepts_x", "accepts_y"):
        assert abs(st_g[key] - st_c[key]) <= max(5, 0.01 * st_c[key]), (family, key, st_g[key], st_c[key])


@pytest.mark.parametrize("family", ["gather", "tiled", "blocked"])
def test_two_ragged_shards_of_a_power_law_omega_equal_one_handle(monkeypatch, family):
    """Blocks balanced by observation count are far from equal in rows / columns on such data; every per-segment choice (waves per segment,
    which rows the cached sweep takes) is a function of the segment's own length, so the shards give the single handle's bits."""
    w, pa, X0, Y0 = problem(m=12000, n=1600, nnz=1_200_000, nonneg=family == "blocked")
    env, kw, bits = FAMILIES[family]
    for k_, v in env.items():
        monkeypatch.setenv(k_, v)
    api = _capi.hip_api()
    params = L.ProxGradParams(max_iter=4, abs_tol=0.0, rel_tol=-1.0)
    o1, X1, Y1, st1 = cases.run_engine(api, pa, X0, Y0, params, **kw)
    assert st1["tiled"] & bits == bits
    rb, cb = [0, 4100, pa.m], [0, 500, pa.n]                 # ragged on purpose (a random permutation of the degrees balances halves)
    o2, X2, Y2, _ = cases.run_shards_on_one_device(api, pa, X0, Y0, params, rb, cb, x_chunks=2, **kw)
    assert np.array_equal(o2, o1[1:]) and np.array_equal(X2, X1) and np.array_equal(Y2, Y1)


@pytest.mark.parametrize("family", ["blocked", "tiled"])
def test_pass_families_on_skewed_columns_add_in_their_reported_order(monkeypatch, family):
    """Round 5: the phase-aligned column passes hand their columns out longest first (which slot a column sits in changes no sum), and both
    pass families -- phase-aligned and LDS-tiled -- divert the columns of at least glrm_sum_order.long_from observations to the 8-wave
    gather sweep beside the passes.  The engine reports that rule, the oracle adds in the reported order -- windowed for the short columns,
    strided on 8 waves for the diverted ones -- and lands on the engine's factors BIT FOR BIT; two ragged shards divert the same columns."""
    import test_gpu_sum_order as S
    env, kw, bits = FAMILIES[family]
    for k_, v in env.items():
        monkeypatch.setenv(k_, v)
    monkeypatch.setenv("GLRM_HIP_BLOCKED_LONG_FROM" if family == "blocked" else "GLRM_HIP_TILED_LONG_FROM", "3000")
    w, pa, X0, Y0 = problem(m=20000, n=1500, k=64, nnz=1_500_000, nonneg=True)
    lens = np.diff(pa.colptr)
    assert (lens >= 3000).sum() >= 5 and (lens < 3000).sum() > 1000          # both kinds of column exist
    o = S.engine_and_oracle_in_its_order(pa, X0, Y0, 6, bits & 34, ("windowed", "windowed"), **kw)
    assert o[1].long_from == 3000 and o[0].long_from == 0
    api = _capi.hip_api()
    params = L.ProxGradParams(max_iter=4, abs_tol=0.0, rel_tol=-1.0)
    o1, X1, Y1, _ = cases.run_engine(api, pa, X0, Y0, params, **kw)
    o2, X2, Y2, _ = cases.run_shards_on_one_device(api, pa, X0, Y0, params, [0, 7000, pa.m], [0, 400, pa.n], x_chunks=2, **kw)
    assert np.array_equal(o2, o1[1:]) and np.array_equal(X2, X1) and np.array_equal(Y2, Y1)
