"""-m gpu: the bench.py modes added in round 3, on scaled-down problems: --emulate-rank (one rank's shard of the N-way job on one GPU,
kernel families from the whole problem's signature) and lists handed over in place (--borrow on)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUIET = ["--no-convergence-run", "--no-cpu-baseline", "--no-jref", "--pmc", "off"]


def run(args, env=None):
    r = subprocess.run([sys.executable, "bench.py"] + args, env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_emulated_rank_runs_the_families_of_the_whole_problem():
    """2M x 100k rows of the C4 recipe (2e8 observations): the whole problem picks the cached row sweep and the phase-aligned passes; rank 3
    of 4 holds a quarter of the rows and columns (5e7 observations: below every threshold on its own) and must run the same two."""
    common = ["--config", "C4", "--rows", "2000000", "--steps", "3", "--warmup", "1"]
    whole = run(common + QUIET)
    shard = run(common + ["--emulate-rank", "3", "--of", "4"])
    assert whole["config"]["row_sweep"] == "cached" and whole["config"]["col_sweep"] == "blocked"
    assert shard["mode"] == "emulate-rank" and shard["families"]["row_sweep"] == "cached" and shard["families"]["col_sweep"] == "blocked"
    assert shard["shard_rows"] == 500_000 and shard["shard_cols"] == 25_000 and shard["whole_signature"]["nnz_rows"] == 2 * 10 ** 8
    ms = shard["measured_ms"]
    assert 0 < ms["step_x"] < whole["kernels"]["row_sweep_ms"] and 0 < ms["step_y"] < whole["kernels"]["col_sweep_ms"]
    ex = shard["exchange_model_ms"]["X_block"]
    assert ex["ring"] == pytest.approx(3 * ex["direct"]) and shard["predicted_iteration_ms"]["direct_no_overlap"] > ms["step_x"] + ms["step_y"]


def test_borrowed_lists_give_the_same_bench_objective():
    common = ["--config", "C5", "--rows", "60000", "--cols", "6000", "--obs-per-row", "200", "--steps", "3", "--warmup", "1"] + QUIET
    a = run(common + ["--borrow", "off"])
    b = run(common + ["--borrow", "on"])
    assert not a["setup_s"]["lists_borrowed_in_place"] and b["setup_s"]["lists_borrowed_in_place"]
    assert a["objective"] == b["objective"] and a["config"]["row_sweep"] == b["config"]["row_sweep"]
    assert a["step_model"]["within_peak"]


def test_inlib_host_equals_the_single_device_objective():
    """bench.py --host inlib: the fit a Julia fit!(glrm, HipProxGradParams(ngpus = N)) ccalls -- glrm_hip_multi_create / _fit, one process,
    N shards (here all on device 0) -- on the C4 recipe: same JSON shape as the torch.distributed host, the same objective bits as the
    one-GPU line after the same number of iterations (shards are bit-identical to the single handle), exchange diagnostics present."""
    common = ["--config", "C4", "--rows", "200000", "--cols", "10000", "--steps", "4", "--warmup", "2"]
    one = run(common + QUIET)
    lib = run(common + ["--host", "inlib", "--gpus", "4", "--shared-device"])
    assert lib["n_gpus"] == 4 and lib["config"]["host"] == "inlib" and lib["host"]["shared_device"]
    assert lib["objective"]["after_warmup_and_steps"] == one["objective"]["after_warmup_and_steps"]
    assert lib["value"] > 0 and lib["ms_per_step"] > 0 and lib["host"]["exchange_ms_per_step_exposed"] >= 0
    assert len(lib["host"]["row_bounds"]) == 5 and lib["host"]["row_bounds"][-1] == 200000
    m = lib["host"]["model_ms"]["X_block"]
    assert m["ring"] == pytest.approx(3 * m["direct"]) and m["link_GBps_one_direction"] == 76.8
