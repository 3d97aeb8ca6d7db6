"""bench.py's host-side pieces that need no GPU: the per-family roofline model (every fraction a ratio of an achieved rate to the
peak of the limiter it is priced against, never an on-chip-reuse artefact above 1 at the measured times), the config table, and the
CPU legs (cpu_baseline sample, J_ref leg) with the oracle standing in for the engine."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_algorithmic_bytes_match_the_survey():
    assert bench.algorithmic_bytes_per_update(32) == 536 and bench.algorithmic_bytes_per_update(64) == 1048


@pytest.mark.parametrize("family,kw,bound", [
    # measured half-step times of round 2 (profiles/r02_*_bench_final.json)
    ("blocked", dict(nnz=10**9, nseg=100_000, nopp=10_000_000, k=64, ld=64, ms=143.0), "infinity_cache"),   # C4 Y half-step: X = 5 GB, gathers out of the window kept on chip
    ("gather", dict(nnz=10**9, nseg=10_000_000, nopp=100_000, k=64, ld=64, ms=128.8), "l2"),     # C4 X half-step: Y = 51 MB, cache resident
    ("tiled", dict(nnz=5 * 10**8, nseg=10_000, nopp=1_000_000, k=32, ld=32, ms=9.7), "lds"),     # C2 Y half-step
    ("dense", dict(nnz=10**10, nseg=1_000_000, nopp=10_000, k=32, ld=32, ms=42.3, m=1_000_000, n=10_000), "mfma"),
    ("dense", dict(nnz=10**10, nseg=1_000_000, nopp=10_000, k=32, ld=32, ms=24.4, m=1_000_000, n=10_000, quad_gram=True), "mfma"),
    ("cached", dict(nnz=10**9, nseg=10_000_000, nopp=100_000, k=64, ld=64, ms=85.3), "infinity_cache"),  # C4 X half-step, one gather pass
    ("cached", dict(nnz=10**9, nseg=10_000_000, nopp=2_000_000, k=64, ld=64, ms=120.0), "hbm"),  # the same with Y = 1 GB: HBM gathers
])
def test_roofline_fractions_are_fractions(family, kw, bound):
    r = bench.kernel_roofline(family, **kw)
    assert r["best"]["bound"] == bound
    assert 0 < r["best"]["frac"] <= 1.0
    for c in r["candidates"]:
        assert 0 < c["frac"] <= 1.0 and c["frac"] == pytest.approx(c["achieved"] / c["peak"])
    assert r["best"]["frac"] == max(c["frac"] for c in r["candidates"])


def test_c4_gather_column_sweep_is_priced_at_the_survey_bytes():
    r = bench.kernel_roofline("gather", nnz=10**9, nseg=100_000, nopp=10_000_000, k=64, ld=64, ms=161.3)
    assert r["best"]["bound"] == "hbm" and r["best"]["per_launch"] == 1048 * 10**9
    assert r["best"]["frac"] == pytest.approx(0.812, abs=2e-3)  # profiles/r02_c4_bench.json


def test_the_roofline_headline_is_one_convention_and_a_fraction():
    """VERDICT r4 weak 7 / r5 item 4.  The C4 Y half-step (phase-aligned passes) at 130.8 and 131.7 ms printed frac 0.876 (PMC) and 0.9946
    (algorithmic) in round 4, 1.012 "of HBM" in round 5 -- bytes that demonstrably did not all come from HBM priced at the HBM spec, while the
    cached row sweep of the same line was priced at the measured Infinity-Cache gather ceiling.  Now one convention: every limiter where its
    bytes come from -- the gathers of the phase-aligned passes at the measured 8.2 TB/s like the cached row sweep's -- so `frac` = floor time /
    measured time <= 1; SURVEY 8(d)'s algorithmic fraction and the PMC traffic fraction stand beside it."""
    fr = {}
    for ms in (129.4, 130.8, 131.7):
        rl = bench.kernel_roofline("blocked", nnz=10**9, nseg=100_000, nopp=10_000_000, k=64, ld=64, ms=ms)
        r = bench.roofline_block(rl, "col passes", ms, 10**9, 9.17e11, "pmc", {"hit_rate": 0.189})
        assert r["bound"] == "infinity_cache" and r["per_launch"] == 2 * 512 * 10**9 and r["peak"] == 8200.0
        assert r["frac"] == pytest.approx(1.024e12 / (ms * 1e-3) / 8.2e12) and 0.9 < r["frac"] <= 1.0
        assert r["algorithmic_frac"] == pytest.approx(1.048e12 / (ms * 1e-3) / 8e12)
        assert r["cache_served"] is False  # (the flag is about HBM-priced fractions)
        assert r["traffic_frac"] == pytest.approx(9.17e11 / (ms * 1e-3) / 8e12) and r["traffic_frac"] < r["frac"]
        hbm = [c for c in r["candidates"] if c["bound"] == "hbm"][0]
        assert hbm["frac"] < r["frac"] and hbm["frac"] == pytest.approx((2 * 12e9 + 2 * 100_000 * 512 + 2 * 5.12e9) / (ms * 1e-3) / 8e12)
        fr[ms] = r["frac"]
    assert fr[129.4] > fr[130.8] > fr[131.7] and abs(fr[130.8] - fr[131.7]) < 0.01   # 1 % apart in time, 1 % apart in the headline
    slow = bench.roofline_block(bench.kernel_roofline("gather", nnz=10**9, nseg=100_000, nopp=10_000_000, k=64, ld=64, ms=161.3), "k", 161.3, 10**9, None, "off", None)
    assert slow["cache_served"] is False and slow["traffic_frac"] is None and slow["frac"] == pytest.approx(0.812, abs=2e-3)  # random gathers from HBM: HBM-priced


def test_every_family_of_the_committed_profile_lines_prints_a_fraction():
    """frac <= 1 for every half-step of every config the driver and the builder recorded (profiles/, BENCH_r05.json), recomputed from the
    recorded times under the one convention: the kernel did the work in that time, so no limiter's floor can exceed it."""
    import glob
    import json
    lines = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[56]_c*_bench*.json"))) + [os.path.join(ROOT, "BENCH_r05.json")]:
        try:
            txt = open(path).read()
        except OSError:
            continue
        cands = txt.splitlines()
        try:
            whole = json.loads(txt)
            cands = [txt] if "kernels" in whole else str(whole.get("tail", "")).splitlines()  # (a driver record keeps the line in its `tail`)
        except ValueError:
            pass
        for ln in cands:
            ln = ln.strip()
            if ln.startswith("{") and '"kernels"' in ln:
                try:
                    r = json.loads(ln)
                except ValueError:
                    continue
                if "config" in r and "kernels" in r:
                    lines.append((os.path.basename(path), r))
    assert len(lines) >= 3, [p for p, _ in lines]
    seen = set()
    for name, r in lines:
        c, kn = r["config"], r["kernels"]
        if c.get("degree") == "zipf":
            continue  # (the uniform-recipe byte model)
        m, n, k = c["m"], c["n"], c["k"]
        ld = 8 if k <= 8 else 16 if k <= 16 else 32 if k <= 32 else 64 if k <= 64 else 128
        qg = "quad_gram" in c["workload"]
        for side, fam, ms, nseg, nopp in (("row", c["row_sweep"], kn["row_sweep_ms"], m, n), ("col", c["col_sweep"], kn["col_sweep_ms"], n, m)):
            rl = bench.kernel_roofline(fam, nnz=c["observed"], nseg=nseg, nopp=nopp, k=k, ld=ld, ms=ms, m=m, n=n, quad_gram=qg)
            assert 0 < rl["best"]["frac"] <= 1.0, (name, side, fam, ms, rl["best"])
            seen.add(fam)
    assert {"cached", "blocked"} <= seen, seen


def test_config_table():
    c4 = bench.CONFIGS["C4"]
    assert (c4["rows"], c4["cols"], c4["k"], c4["q"], c4["reg"][0]) == (10_000_000, 100_000, 64, 100, 3) and bench.nonneg_start(c4)
    assert not bench.nonneg_start(bench.CONFIGS["C2"])
    for c in bench.CONFIGS.values():
        assert c["cols"] % c["q"] == 0
        if c["jref"]:
            assert c["jref"][1] % c["jref"][2] == 0


def test_cpu_legs_with_the_oracle_standing_in(monkeypatch):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O

    class Args:
        seed, cpu_sample_rows, rows = 20260926, 64, 10_000

    cfg = dict(bench.CONFIGS["C4"], jref=(600, 400, 20))
    api = O.oracle_api()
    orig = api.create
    monkeypatch.setattr(api, "create", lambda pa, device_id=0, **kw: orig(pa))

    class Dev:
        index = 0

    r = bench.jref_leg(Args, cfg, api, Dev)
    assert r["gpu_first_iteration_at_or_below_J_ref"] is not None and r["gpu_objective_there"] <= r["J_ref"] * (1 + 1e-5)
    assert r["gpu_first_iteration_at_or_below_J_ref"] <= r["cpu_iterations_to_own_stop"]
    monkeypatch.setattr(bench.time, "time", _fast_clock())
    Args.cpu_cols_sample, Args.cpu_target_obs = True, 40_000   # 2 000 rows x 20 per row, then 10 000 rows x 4 per row
    b = bench.cpu_baseline(Args, dict(cfg, q=20), 64, 20, 400, 10_000)
    assert b["kind"] == "port" and b["value"] > 0 and b["cores"] >= 1 and "first" in b["rows_sample"]["sample"]
    # the second sample keeps the columns at their full length: all rows x the first strata of the generator
    assert "first 2000 rows" in b["rows_sample"]["sample"] and "all 10000 rows x the first 80 columns" in b["columns_sample"]["sample"] and "columns sample" in b["sample"]
    rx, ry = b["rows_sample"]["x_halfstep_updates_per_s"], b["columns_sample"]["y_halfstep_updates_per_s"]
    assert b["value"] == pytest.approx(2 / (1 / rx + 1 / ry))


def _fast_clock():
    import time as _t
    t0, p0, real = _t.time(), _t.perf_counter(), _t.perf_counter  # perf_counter: bench.time.time itself is what gets patched
    return lambda: t0 + (real() - p0) * 50.0  # the 10-second sampling window of cpu_baseline in 0.2 s


def test_cached_family_prices_one_gather_pass():
    two = bench.kernel_roofline("gather", nnz=10**9, nseg=10_000_000, nopp=100_000, k=64, ld=64, ms=100.0)
    one = bench.kernel_roofline("cached", nnz=10**9, nseg=10_000_000, nopp=100_000, k=64, ld=64, ms=100.0)
    by = lambda r, b: next(c for c in r["candidates"] if c["bound"] == b)
    assert by(one, "infinity_cache")["per_launch"] * 2 == by(two, "l2")["per_launch"] == 2 * 10**9 * 8 * 64
    # priced against the MEASURED ceiling of such reads out of the Infinity Cache: 5.12e11 B in 85.5 ms = 0.73
    r = bench.kernel_roofline("cached", nnz=10**9, nseg=10_000_000, nopp=100_000, k=64, ld=64, ms=85.5)
    assert r["best"]["bound"] == "infinity_cache" and r["best"]["frac"] == pytest.approx(0.73, abs=5e-3) and "MEASURED" in r["best"]["peak_is"]


def test_step_model_reproduces_the_round_two_check():
    """BENCH_r02: 228.9 ms per iteration at C4.  SURVEY 8(d) (P = 2 on both sides) gives 9.16 TB/s -- above the peak; with the cached row
    sweep's single pass the step moves ~1.59e12 B = 6.9 TB/s = 0.87 of the peak."""
    sm = bench.step_model("cached", "blocked", 10**9, 10**9, 10_000_000, 100_000, 64, 64, 228.9, 1, m=10_000_000, n=100_000)
    assert sm["passes"] == {"x": 1, "y": 2} and sm["within_peak"]
    # HBM itself: Y (51 MB) lives in the Infinity Cache, so the X half-step's floor is its streams; X (5 GB) does not
    assert sm["hbm_floor"]["opposing_factor_cache_resident"] == {"x": True, "y": False} and sm["hbm_floor"]["GBps"] == pytest.approx(4677, abs=10)
    assert sm["survey_8d_P2_GBps"] == pytest.approx(9157, abs=5) and sm["GBps"] == pytest.approx(6913, abs=10)
    assert bench.step_model("gather", "gather", 10**9, 10**9, 10_000_000, 100_000, 64, 64, 290.0, 1, m=10_000_000, n=100_000)["passes"] == {"x": 2, "y": 2}
    # LDS-tiled sweeps (C5 at its stated size, 582.8 ms): the k-vectors are fetched per workgroup, not per update -- no 9.2 TB/s artefact
    t = bench.step_model("tiled", "tiled", 5 * 10**9, 5 * 10**9, 5_000_000, 50_000, 32, 32, 582.8, 1, m=5_000_000, n=50_000)
    assert t["within_peak"] and t["survey_8d_P2_GBps"] > 8000 and t["GBps"] == pytest.approx(420, abs=10)


def test_quad_gram_prices_the_flop_that_are_left():
    a = bench.kernel_roofline("dense", nnz=10**10, nseg=10**6, nopp=10**4, k=32, ld=32, ms=40.0, m=10**6, n=10**4)
    b = bench.kernel_roofline("dense", nnz=10**10, nseg=10**6, nopp=10**4, k=32, ld=32, ms=40.0, m=10**6, n=10**4, quad_gram=True)
    mf = lambda r: next(c for c in r["candidates"] if c["bound"] == "mfma")
    assert mf(a)["per_launch"] == 6.0 * 10**6 * 10**4 * 32 and mf(b)["per_launch"] == 4.0 * 10**6 * 10**4 * 32
    assert mf(a)["measured_ceiling"]["v_mfma_f64_16x16x4_f64"] == 50.3


def test_exchange_model_and_signature_combine():
    """xGMI model printed beside the shard measurements: direct = one block per link in ONE direction (76.8 GB/s: AMD's 153.6 GB/s per link
    counts both), ring = N - 1 hops, the optimistic variant and the equivalent all-gather bus bandwidth printed beside it; the whole
    problem's signature is the sum of the shards' observation counts and the max of everything else."""
    from lowrankmodels.jl_amd import _capi
    ex = bench.exchange_model_ms(640e6, 8)
    assert ex["direct"] == pytest.approx(640e6 / 76.8e9 * 1e3) and ex["ring"] == pytest.approx(7 * ex["direct"])
    assert ex["optimistic_direct_if_153GBps_were_per_direction"] == pytest.approx(ex["direct"] / 2) and ex["link_GBps_one_direction"] == 76.8
    assert ex["busbw_equivalent_GBps_direct"] == pytest.approx(7 * 76.8)
    assert bench.exchange_model_ms(1e9, 1) == {"direct": 0.0, "ring": 0.0}
    a, b = _capi.CSignature(10, 12, 3, 5, 0, 1), _capi.CSignature(7, 1, 9, 2, 1, 0)
    w = _capi.CSignature.combine([a, b.astuple()])
    assert w.astuple() == (17, 13, 9, 5, 1, 1)
    assert bench.passes_priced("cached") == 1 and bench.passes_priced("blocked") == 2
