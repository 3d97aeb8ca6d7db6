"""N>1 path on CPU: world_size-2 gloo run of the sharded fit (rows for the X half-step, columns for the Y
half-step, all-gather of the updated factor between half-steps) equals the single-process fit bit for bit
(SURVEY.md section 8(e) determinism contract)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import cases
import lowrankmodels.jl_amd as L
import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_world(tmp_path, names, nproc=2, env_extra=None):
    out = str(tmp_path / "dist")
    env = dict(os.environ, OMP_NUM_THREADS="1", **(env_extra or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(HERE, "_dist_worker.py"), out] + names
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return [np.load(f"{out}.rank{i}.npz") for i in range(nproc)]


def test_partition_balances_nnz():
    ptr = np.concatenate([[0], np.cumsum([10] * 50 + [1000] * 2 + [10] * 48)])
    b = L.partition(ptr, 4)
    assert b[0] == 0 and b[-1] == 100 and all(b[i] <= b[i + 1] for i in range(4))
    loads = [ptr[b[i + 1]] - ptr[b[i]] for i in range(4)]
    assert max(loads) <= 1000 + ptr[-1] / 4
    assert L.partition(np.arange(0, 101, 1) * 7, 4) == [0, 25, 50, 75, 100]  # uniform -> equal-count blocks
    assert L.partition(np.zeros(11, dtype=np.int64), 2) == [0, 5, 10]


@pytest.mark.parametrize("mode", ["auto", "allgather", "p2p"])
def test_two_ranks_equal_one_rank(tmp_path, mode):
    names = ["c1", "mixed", "kmeans"]
    ranks = run_world(tmp_path, names, 2, {"GLRM_GATHER": mode})
    O.set_threads(1)
    for name in names:
        kwargs, params = cases.build_golden_case(name)
        g = L.GLRM(**kwargs)
        X, Y, ch = L.fit_b(g, params, verbose=False, engine=O.oracle_api())
        for z in ranks:  # every rank ends with the full replicated factors
            assert np.array_equal(z[name + "_X"], X), name
            assert np.array_equal(z[name + "_Y"], Y), name
            o = z[name + "_obj"]
            assert len(o) == len(ch.objective)
            assert np.array_equal(o[1:], np.array(ch.objective[1:])), name  # fixed-order sum of gathered columns
            assert cases.rel_err(o[:1], ch.objective[:1]) < 1e-12  # initial objective: column-blocked vs one accumulator


@pytest.mark.parametrize("nproc", [2, 8])
def test_c4_recipe_shards_equal_one_rank(tmp_path, nproc):
    """The north-star recipe (rank 64, QuadLoss, NonNegConstraint on X and Y; tests/golden/c4.npz) on 2 and on 8 shards -- the world
    size the scaling bench ends at; 150 rows / 80 columns over 8 ranks are ragged blocks."""
    ranks = run_world(tmp_path, ["c4"], nproc)
    O.set_threads(1)
    kwargs, params = cases.build_golden_case("c4")
    g = L.GLRM(**kwargs)
    X, Y, ch = L.fit_b(g, params, verbose=False, engine=O.oracle_api())
    for z in ranks:
        assert np.array_equal(z["c4_X"], X) and np.array_equal(z["c4_Y"], Y)
        assert np.array_equal(z["c4_obj"][1:], np.array(ch.objective[1:]))


@pytest.mark.parametrize("mode", ["auto", "p2p"])
def test_three_ranks_ragged_blocks(tmp_path, mode):
    """m, n not divisible by the world size -> ragged blocks -> one broadcast per owner (auto) or one group of point-to-point
    transfers, every owner straight to every peer (p2p)."""
    ranks = run_world(tmp_path, ["nnmf"], 3, {"GLRM_GATHER": mode})
    kwargs, params = cases.build_golden_case("nnmf")
    g = L.GLRM(**kwargs)
    X, Y, ch = L.fit_b(g, params, verbose=False, engine=O.oracle_api())
    for z in ranks:
        assert np.array_equal(z["nnmf_X"], X) and np.array_equal(z["nnmf_Y"], Y)
        assert np.array_equal(z["nnmf_obj"][1:], np.array(ch.objective[1:]))


@pytest.mark.parametrize("arrival", ["1", "0"])
def test_pipelined_x_exchange_equals_one_rank(tmp_path, arrival):
    """GLRM_X_CHUNKS=2: the X half-step runs in two row chunks whose all-gather is pipelined behind the next chunk
    (glrm_*_step_x_range); rows are independent, so the result is still the single-process bits.  GLRM_ARRIVAL=1 (default): the first inner
    Y sweep is told which rows every chunk exchange fills (glrm_*_step_y_arrival) instead of waiting for the pipeline; 0: the wait."""
    ranks = run_world(tmp_path, ["c1", "kmeans"], 2, {"GLRM_X_CHUNKS": "2", "GLRM_ARRIVAL": arrival})
    O.set_threads(1)
    for name in ["c1", "kmeans"]:
        kwargs, params = cases.build_golden_case(name)
        g = L.GLRM(**kwargs)
        X, Y, ch = L.fit_b(g, params, verbose=False, engine=O.oracle_api())
        for z in ranks:
            assert np.array_equal(z[name + "_X"], X) and np.array_equal(z[name + "_Y"], Y), name
            assert np.array_equal(z[name + "_obj"][1:], np.array(ch.objective[1:])), name


def test_two_ranks_multidimensional_losses(tmp_path):
    """Columns that own several vectors of Y: the Y exchange moves the span [ystart[cb], ystart[ce]) of each rank's block."""
    names = ["mnl_ordinal", "loss_test"]
    ranks = run_world(tmp_path, names, 2, {"GLRM_GATHER": "allgather"})
    O.set_threads(1)
    for name in names:
        kwargs, params = cases.build_golden_case(name)
        g = L.GLRM(**kwargs)
        X, Y, ch = L.fit_b(g, params, verbose=False, engine=O.oracle_api())
        for z in ranks:
            assert np.array_equal(z[name + "_X"], X) and np.array_equal(z[name + "_Y"], Y), name
            assert np.array_equal(z[name + "_obj"][1:], np.array(ch.objective[1:])), name


@pytest.mark.parametrize("nproc,chunks", [(4, "1"), (4, "2"), (3, "1")])
def test_exchange_probe_picks_one_mode_on_every_rank(tmp_path, nproc, chunks):
    """More than two ranks on RCCL: ShardedFit times an all-gather and a grouped send / recv exchange of the X blocks at set-up and keeps
    the faster one on every rank (fit.py: _probe_exchange).  GLRM_GATHER_PROBE=1 runs that code on gloo: whichever mode wins, the fit
    equals the single-process bits -- equal blocks (4 ranks, also with pipelined row chunks) and ragged ones (3 ranks: the all-gather leg
    of the probe runs as one broadcast per owner)."""
    names = ["c4"] if nproc == 4 else ["nnmf"]
    ranks = run_world(tmp_path, names, nproc, {"GLRM_GATHER_PROBE": "1", "GLRM_X_CHUNKS": chunks})
    assert all("probe_ms" in z.files and np.all(np.isfinite(z["probe_ms"])) and np.all(z["probe_ms"] > 0) for z in ranks)  # both legs ran
    assert len({bool(z["probe_chose_p2p"]) for z in ranks}) == 1                                                            # one choice
    O.set_threads(1)
    for name in names:
        kwargs, params = cases.build_golden_case(name)
        g = L.GLRM(**kwargs)
        X, Y, ch = L.fit_b(g, params, verbose=False, engine=O.oracle_api())
        for z in ranks:
            assert np.array_equal(z[name + "_X"], X) and np.array_equal(z[name + "_Y"], Y), name
            assert np.array_equal(z[name + "_obj"][1:], np.array(ch.objective[1:])), name
