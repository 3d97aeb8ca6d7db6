"""glrm_*_multi_* (one process, N shards; include/glrm_hip.h): the library-side sharding, exchange bookkeeping and outer loop give
the single-shard bits for every shard count.  CPU: the oracle's twin (shared buffers).  -m gpu: the HIP engine with several shards
on ONE device (device_ids repeat; the peer pushes become device-to-device copies) against the single-device fit and the oracle."""
import numpy as np
import pytest

import cases
import lowrankmodels.jl_amd as L
import oracle as O
from lowrankmodels.jl_amd import _capi

GOLDEN_NAMES = ["c1", "c4", "nnmf", "mixed", "kmeans", "mnl_ordinal", "loss_test"]


def run_multi(api, pa, X0, Y0, params, n_shards, **kw):
    mh = api.multi_create(pa, n_shards, **kw)
    try:
        X, Y = np.array(X0, order="F"), np.array(Y0, order="F")
        obj, sec = api.multi_fit(mh, params, X, Y)
        info = api.multi_info(mh, n_shards)
    finally:
        api.multi_destroy(mh)
    return obj, X, Y, info


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_oracle_multi_equals_single(name):
    kwargs, params = cases.build_golden_case(name)
    pa = L.GLRM(**kwargs).problem_arrays()
    O.set_threads(1)
    api = O.oracle_api()
    o1, X1, Y1, _ = cases.run_engine(api, pa, kwargs["X"], kwargs["Y"], params)
    for n in (1, 2, 3, 8):
        o, X, Y, info = run_multi(api, pa, kwargs["X"], kwargs["Y"], params, n)
        assert len(o) == len(o1) and np.array_equal(o[1:], o1[1:]), (name, n)
        assert cases.rel_err(o[:1], o1[:1]) < 1e-12       # initial objective: per-column sums first
        assert np.array_equal(X, X1) and np.array_equal(Y, Y1), (name, n)
        assert info["row_bounds"][0] == 0 and info["row_bounds"][-1] == pa.m and info["col_bounds"][-1] == pa.n
        assert info["row_bounds"] == L.partition(pa.rowptr, n) and info["col_bounds"] == L.partition(pa.colptr, n)


def test_oracle_multi_host_level_and_regularizer_swap():
    """HipProxGradParams(ngpus=N) through fit!: cached multi handle, warm start, scale_regularizer! keeps the handle."""
    kwargs, params = cases.build_golden_case("c1")
    api = O.oracle_api()
    g1, g2 = L.GLRM(**kwargs), L.GLRM(**{**kwargs, "X": kwargs["X"].copy(), "Y": kwargs["Y"].copy()})
    p1, p2 = L.HipProxGradParams(max_iter=6), L.HipProxGradParams(max_iter=6, ngpus=3)
    for g, p in ((g1, p1), (g2, p2)):
        L.fit_b(g, p, verbose=False, engine=api)
        L.scale_regularizer_(g, 0.5)
        L.fit_b(g, p, verbose=False, engine=api)
    assert len(g2._handle_cache) == 6 and g2._handle_cache[5] == "multi"
    assert np.array_equal(g1.X, g2.X) and np.array_equal(g1.Y, g2.Y)
    g1.close(); g2.close()


def test_multi_argument_checks():
    kwargs, params = cases.build_golden_case("c1")
    pa = L.GLRM(**kwargs).problem_arrays()
    api = O.oracle_api()
    with pytest.raises(_capi.GLRMError):
        api.multi_create(pa, 0)
    shard = _capi.ProblemArrays(pa.m, pa.n, pa.k, pa.rowptr, pa.colidx, pa.rowvals, pa.colptr, pa.rowidx, pa.colvals, pa.losses, pa.rx,
                                pa.ry, row_begin=0, row_end=50)
    with pytest.raises(_capi.GLRMError):
        api.multi_create(shard, 2)
    with pytest.raises(ValueError):
        L.HipProxGradParams(ngpus=2, device_ids=[0])


# ------------------------------------------------------------------------------------------------ GPU

@pytest.mark.gpu
@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_hip_multi_shards_on_one_device_equal_single(name):
    kwargs, params = cases.build_golden_case(name)
    pa = L.GLRM(**kwargs).problem_arrays()
    api = _capi.hip_api()
    o1, X1, Y1, _ = cases.run_engine(api, pa, kwargs["X"], kwargs["Y"], params)
    for n, chunks in ((2, 0), (3, 2), (8, 0)):
        o, X, Y, info = run_multi(api, pa, kwargs["X"], kwargs["Y"], params, n, device_ids=[0] * n, x_chunks=chunks)
        assert len(o) == len(o1) and np.array_equal(o[1:], o1[1:]), (name, n)
        assert cases.rel_err(o[:1], o1[:1]) < 1e-12
        assert np.array_equal(X, X1) and np.array_equal(Y, Y1), (name, n)
        assert info["exchange"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("tiled", [1, 2])
def test_hip_multi_c4_recipe_vs_oracle(tiled):
    """The north-star recipe on 4 shards of one device (gather and LDS-tiled sweeps, pipelined X exchange) against the oracle."""
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(20000, 2000, 64, 100, value_model=1)
    one = np.array([(0, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(3, 0, 1.0)], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(20000, 2000, 64, rowptr, colidx, rowvals, colptr, rowidx, colvals, one, reg, reg)
    X0, Y0 = np.asfortranarray(np.abs(X0) / 8.0), np.asfortranarray(np.abs(Y0) / 8.0)
    params = L.ProxGradParams(max_iter=8)
    O.set_threads(4)
    o_c, X_c, Y_c, _ = cases.run_engine(O.oracle_api(), pa, X0, Y0, params)
    o, X, Y, info = run_multi(_capi.hip_api(), pa, X0, Y0, params, 4, device_ids=[0] * 4, x_chunks=4, tiled=tiled, profile=1)
    assert len(o) == len(o_c) and cases.rel_err(o, o_c) < 1e-5
    assert cases.fro_err(X, X_c) < 1e-5 and cases.fro_err(Y, Y_c) < 1e-5
    assert info["exchange_ms"] >= 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("quad_gram", [False, True])
def test_hip_multi_dense_handover_and_host_level(quad_gram):
    """Fully observed QuadLoss: the dense hand-over (matrix cores) on two shards, through HipProxGradParams(ngpus=2); with
    glrm_options.quad_gram the Gram matrix of the replicated opposing factor is the same on every shard."""
    rng = np.random.default_rng(5)
    m, n, k = 300, 200, 16
    A = rng.standard_normal((m, k)) @ rng.standard_normal((k, n)) / 4 + 0.1 * rng.standard_normal((m, n))
    X0, Y0 = rng.standard_normal((k, m)), rng.standard_normal((k, n))
    g1 = L.GLRM(A, L.QuadLoss(), L.ZeroReg(), L.ZeroReg(), k, X=X0.copy(), Y=Y0.copy())
    g2 = L.GLRM(A, L.QuadLoss(), L.ZeroReg(), L.ZeroReg(), k, X=X0.copy(), Y=Y0.copy())
    _, _, ch1 = L.fit_b(g1, L.HipProxGradParams(max_iter=10, quad_gram=quad_gram), verbose=False)
    _, _, ch2 = L.fit_b(g2, L.HipProxGradParams(max_iter=10, ngpus=2, device_ids=[0, 0], quad_gram=quad_gram), verbose=False)
    assert cases.rel_err(ch2.objective, ch1.objective) < 1e-9 and cases.fro_err(g2.X, g1.X) < 1e-9
    g1.close(); g2.close()


@pytest.mark.gpu
def test_hip_multi_rccl_single_device_falls_back_or_runs():
    """exchange=1 (RCCL) with one shard is a no-op exchange; with repeated devices the library must fall back to the direct path."""
    kwargs, params = cases.build_golden_case("c1")
    pa = L.GLRM(**kwargs).problem_arrays()
    api = _capi.hip_api()
    o1, X1, Y1, _ = cases.run_engine(api, pa, kwargs["X"], kwargs["Y"], params)
    for n in (1, 2):
        o, X, Y, info = run_multi(api, pa, kwargs["X"], kwargs["Y"], params, n, device_ids=[0] * n, exchange=1)
        assert np.array_equal(o[1:], o1[1:]) and np.array_equal(X, X1)
        assert info["exchange"] == 0


def test_oracle_step_y_arrival_is_step_y_and_validates_the_block_list():
    """The checker's twin of glrm_hip_step_y_arrival: nothing to wait for in one address space; the block list is validated like the
    engine validates it (the blocks tile the rows [0, m) of X) and the half-step is glrm_cpu_step_y."""
    kwargs, params = cases.build_golden_case("nnmf")
    pa = L.GLRM(**kwargs).problem_arrays()
    api = O.oracle_api()
    res = {}
    for arrival in (False, True):
        h = api.create(pa)
        try:
            api.set_factors(h, np.asfortranarray(kwargs["X"]), np.asfortranarray(kwargs["Y"]))
            api.reset_stepsizes(h, params.stepsize)
            for _ in range(3):
                api.step_x(h, params.min_stepsize)
                if arrival:
                    api.step_y_arrival(h, params.min_stepsize, [(20, 45, None), (0, 20, None), (45, pa.m, None), (7, 7, None)])
                else:
                    api.step_y(h, params.min_stepsize)
            X, Y = np.zeros_like(kwargs["X"], order="F"), np.zeros_like(kwargs["Y"], order="F")
            api.get_factors(h, X, Y)
            res[arrival] = (X, Y)
            if arrival:
                for bad in ([(0, 20, None), (25, pa.m, None)], [(0, 30, None), (20, pa.m, None)], [(0, pa.m + 1, None)], [(0, 20, None)]):
                    with pytest.raises(_capi.GLRMError):
                        api.step_y_arrival(h, params.min_stepsize, bad)
        finally:
            api.destroy(h)
    assert np.array_equal(res[True][0], res[False][0]) and np.array_equal(res[True][1], res[False][1])


@pytest.mark.gpu
@pytest.mark.parametrize("emulate", ["off", "wait-value", "delay-kernel"])
def test_hip_multi_arrival_order_and_link_emulation(monkeypatch, emulate):
    """The in-library host on 4 shards of one device, phase-aligned passes forced (many super-tiles per block of X): arrival order on
    (glrm_multi_options.arrival = 0 / 1) and off (2), 1 / 4 row chunks -- every combination gives the single-device bits.  With the
    link emulator (GLRM_EXCHANGE_EMULATE_GBPS; both mechanisms) the bits stay and an iteration cannot be faster than its X block needs
    on the emulated link."""
    import time
    monkeypatch.setenv("GLRM_HIP_BLOCKED", "3")
    monkeypatch.setenv("GLRM_HIP_BLOCKED_TPS", "1")
    monkeypatch.setenv("GLRM_HIP_BLOCKED_FILL", "3")
    monkeypatch.setenv("GLRM_HIP_CACHED", "0")
    m, n, k = 8000, 800, 64
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(m, n, k, 100, value_model=1)
    one = np.array([(0, 0, 1.0, 0.0, 0.0)], dtype=_capi.LOSS_DTYPE)
    reg = np.array([(3, 0, 1.0)], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, one, reg, reg)
    X0, Y0 = np.asfortranarray(np.abs(X0) / 8.0), np.asfortranarray(np.abs(Y0) / 8.0)
    iters = 5
    params = L.ProxGradParams(max_iter=iters, abs_tol=0.0, rel_tol=-1.0)
    api = _capi.hip_api()
    o1, X1, Y1, st1 = cases.run_engine(api, pa, X0, Y0, params)
    assert st1["tiled"] & 32
    gbps, dilate = 0.5, 4                       # X block: 2000 rows x 64 x 8 B = 1.02 MB -> 8.2 ms per iteration at 0.5 / 4 GB/s
    if emulate != "off":  # the link emulator lives in the test build of the engine only (libglrm_hip_testing.so); the product library refuses the variable
        monkeypatch.setenv("GLRM_EXCHANGE_EMULATE_GBPS", str(gbps))
        monkeypatch.setenv("GLRM_EXCHANGE_EMULATE_MODE", {"wait-value": "1", "delay-kernel": "2"}[emulate])
        with pytest.raises(_capi.GLRMError) as ei:
            api.multi_create(pa, 4, device_ids=[0] * 4, x_chunks=4, arrival=1, profile=1)
        assert ei.value.code == _capi.ERR_UNSUPPORTED and "test build" in str(ei.value)
        api = _capi.hip_testing_api()
    for arrival, chunks in ((1, 4), (2, 4), (0, 1), (2, 1)):
        t0 = time.perf_counter()
        try:
            mh = api.multi_create(pa, 4, device_ids=[0] * 4, x_chunks=chunks, arrival=arrival, profile=1)
        except _capi.GLRMError as e:
            if emulate == "wait-value" and e.code == _capi.ERR_UNSUPPORTED:
                pytest.skip("hipStreamWaitValue64 is not supported on this device")
            raise
        try:
            X, Y = np.array(X0, order="F"), np.array(Y0, order="F")
            o, sec = api.multi_fit(mh, params, X, Y)
            info = api.multi_info(mh, 4)
        finally:
            api.multi_destroy(mh)
        assert np.array_equal(o[1:], o1[1:]) and np.array_equal(X, X1) and np.array_equal(Y, Y1), (arrival, chunks)
        assert info["exchange"] == 0 and info["exchange_ms"] >= 0.0
        if emulate != "off":
            link_s = (m // 4) * k * 8 / (gbps * 1e9 / dilate)
            assert sec[-1] >= 0.9 * iters * link_s, (emulate, arrival, chunks, sec[-1], iters * link_s)
