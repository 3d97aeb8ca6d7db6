"""bench_legs.py -- everything `bench.py` reports BESIDE the timed region: the config table, the per-family roofline model
(`kernel_roofline`, `roofline_block`: ONE definition of `frac`), the byte model of a step, the CPU legs (`cpu_baseline`, `jref_leg`,
`cpu_full_leg`), the in-run PMC traffic passes, the other hosts and modes (`emulate_rank`, `inlib_host`, `other_config_line`,
`create_from_host_leg`).  `bench.py` itself holds the contract: argument parsing, set-up, the warm-up, the K timed steps between the two
fences, and the JSON line.  Split in round 5 (one 80 KB file hid the timed region among the legs); `from bench_legs
import *` keeps every name where the tests and tools look for it."""
import argparse
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))  # bench.py lives beside this file
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling); L2 ~34.5 TB/s aggregate; LDS ~150 TB/s aggregate for
# ds_read_b64/b128; Infinity Cache 256 MiB.  fp64 matrix peak: 78.6 TFLOP/s (datasheet; the guide's table has no fp64 row).
HBM_PEAK_GBS = 8000.0
L2_PEAK_GBS = 34500.0
LDS_PEAK_GBS = 150000.0
MFMA_F64_PEAK_TFLOPS = 78.6
MALL_BYTES = 256 * 2 ** 20
MALL_GATHER_GBS = 8200.0   # measured, not spec: random 512-byte reads out of the Infinity Cache (profiles/r02_ubench_gather.txt)
# roofline.frac is ALWAYS floor time / measured time of the kernel family's largest floor, each limiter priced where its bytes really come
# from (kernel_roofline): compulsory HBM bytes at the 8 TB/s spec, cache-served k-vector gathers at the measured 8.2 TB/s Infinity-Cache
# gather ceiling, LDS bytes at 150 TB/s, flops at the rating -- so frac <= 1 unless a measured ceiling is beaten.  A gather-family fraction
# above this mark of the HBM spec cannot come from HBM alone (MI355X_MICROARCH.md: 6.29 TB/s measured copy ceiling = 0.79 of the spec) and
# is flagged `cache_served`; `algorithmic_frac` (SURVEY 8(d) bytes / time / HBM peak) and `traffic_frac` (PMC bytes across the fabric / time /
# HBM peak) always stand beside it.
CACHE_SERVED_ABOVE = 0.9


def passes_priced(family):
    """Passes over the segment's observations the family has to make per half-step: gradient + first trial = 2,
    except the cached row sweep, which fetches a row once and runs every pass from registers."""
    return 1 if family == "cached" else 2

CONFIGS = {
    # name: rows, cols, rank, observations per row, value model, loss mix, regularizer descriptor (kind, wrap, scale)
    # jref: the scaled-down problem of the same recipe on which the CPU oracle runs to its own stop (to_ref_objective)
    "C2": dict(rows=1_000_000, cols=10_000, k=32, q=500, value_model=0, loss_mix=0, reg=(1, 0, 1.0), jref=(40_000, 2_000, 100),
               text="BASELINE configs[1] (C2): {m} x {n}, rank {k}, QuadLoss, 5% observed, QuadReg(1.0) on X and Y"),
    "C3": dict(rows=1_000_000, cols=10_000, k=32, q=10_000, value_model=0, loss_mix=0, reg=(0, 0, 1.0), jref=None,
               text="BASELINE configs[2] (C3): {m} x {n}, rank {k}, QuadLoss, fully observed (dense hand-over, fp64 MFMA path), ZeroReg"),
    "C4": dict(rows=10_000_000, cols=100_000, k=64, q=100, value_model=1, loss_mix=0, reg=(3, 0, 1.0), jref=(40_000, 4_000, 100),
               text="BASELINE configs[3] (C4, the north-star target): {m} x {n}, rank {k}, QuadLoss, 0.1% observed, "
                    "NonNegConstraint on X and Y (NNMF)"),
    "C5": dict(rows=5_000_000, cols=50_000, k=32, q=1000, value_model=0, loss_mix=1, reg=(1, 0, 1.0), jref=(20_000, 6_000, 200),
               text="BASELINE configs[4] (C5): {m} x {n}, rank {k}, Quad/Logistic/OrdinalHinge columns (f mod 3), 2% observed, QuadReg(1.0)"),
}
REG_NAME = {0: "ZeroReg", 1: "QuadReg(1.0)", 3: "NonNegConstraint"}
# Starting point.  SURVEY.md 8(d) asked for the reference default X0, Y0 ~ N(0,1) (src/glrm.jl:31) everywhere.  Under NonNegConstraint that
# start has objective Inf, the first trial of every row is accepted whatever its size, and at the C4 shape (100 observations per row,
# rank 64) the fit -- reference, oracle and engine alike -- collapses to X = 0 within two iterations and stays there (objective =
# sum of a^2): every later line search rejects its single trial.  The NNMF configs therefore start from |N(0,1)| / sqrt(k) (same
# streams, non-negative, x.y of the size of the data), on which the fit keeps descending for 100+ iterations.
INIT_NOTE = {True: "X0, Y0 = |N(0,1)| / sqrt(k) (the N(0,1) default collapses an NNMF of this shape to X = 0 in two iterations)",
             False: "X0, Y0 ~ N(0,1) (reference default, src/glrm.jl:31)"}


def nonneg_start(cfg):
    return cfg["reg"][0] == 3


def algorithmic_bytes_per_update(k):
    """P = 2 compulsory passes x (8 B value + 4 B index + k x 8 B factor slice)."""
    return 2 * (8 + 4 + 8 * k)


# ----------------------------------------------------------------------------- rooflines

def kernel_roofline(family, *, nnz, nseg, nopp, k, ld, ms, m=0, n=0, tile=560, segs_per_wg=256, quad_gram=False):
    """Candidate limiters of ONE half-step (all its launches) of the given kernel family, each as achieved/peak.

    family   'gather'  every update fetches the opposing k-vector from memory (csrc/glrm_hip.hip sweep_kernel)
             'tiled'   the opposing factor is staged tile by tile in LDS (csrc/glrm_tiled.hpp)
             'lane'    the same with ONE LANE per segment and 512 segments per staged tile (csrc/glrm_lane.hpp)
             'blocked' phase-aligned gather passes: the k-vector gathers are served by the L2 of the XCD (csrc/glrm_blocked.hip)
             'dense'   fully observed QuadLoss on the fp64 matrix cores (csrc/glrm_dense.hpp)
             'general' multi-dimensional losses (csrc/glrm_multi.hpp)
    nnz updates per launch, nseg own segments, nopp opposing vectors, ld padded rank, ms duration of the half-step.
    P = 2 passes over the segment per half-step (gradient + first trial) is the compulsory minimum."""
    t = ms * 1e-3
    if t <= 0:
        return None
    P = 2
    alg = nnz * P * (12 + 8 * k)                 # every update fetches its own k-vector
    stream = nnz * P * 12 + 2 * nseg * ld * 8   # what no design can avoid: the (index, value) stream per pass + own factor r/w
    opp = nopp * ld * 8
    cands = []
    if family == "dense":
        if quad_gram:  # glrm_options.quad_gram: the trial is O(k^2) per segment from the quadratic form -- one pass over A per half-step
            flops, passes = 4.0 * m * n * k, 1
            what = "4 m n k flop per half-step: residual and gradient products; the trial comes from J(x) + g.s + s'(YY')s (quad_gram)"
        else:
            flops, passes = 6.0 * m * n * k, P  # three m x n x k products per half-step (u, gradient, first trial): 12mnk per iteration
            what = "6 m n k flop per half-step (SURVEY 8(d): 12 m n k per outer iteration)"
        cands.append(dict(bound="mfma", achieved=flops / t / 1e12, peak=MFMA_F64_PEAK_TFLOPS, unit="TFLOP/s", per_launch=flops, what=what,
                          # tools/ubench_mfma.hip, profiles/r02_ubench_mfma.txt: what the instructions sustain with register operands
                          # and nothing else going on, at 2.37-2.40 GHz (v_fma_f64 throttles the clock to ~1.98 GHz)
                          measured_ceiling=dict(v_mfma_f64_16x16x4_f64=50.3, v_mfma_f64_4x4x4_4b_f64=73.1, v_fma_f64=61.1, unit="TFLOP/s",
                                                used="v_mfma_f64_16x16x4_f64", source="profiles/r02_ubench_mfma.txt")))
        cands.append(dict(bound="hbm", achieved=passes * m * n * 8 / t / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", per_launch=passes * m * n * 8,
                          what="A streamed once per pass, %d pass%s per half-step" % (passes, "es" if passes > 1 else "")))
    elif family in ("tiled", "lane"):
        if family == "lane":  # csrc/glrm_lane.hpp: 8 waves x 64 segments share a staged tile
            segs_per_wg = 512
        nwg = max(1, -(-nseg // segs_per_wg))
        staged = nwg * P * opp                   # every workgroup stages the whole opposing factor once per pass
        cands.append(dict(bound="hbm", achieved=(stream + opp) / t / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", per_launch=stream + opp,
                          what="compulsory HBM bytes: P x 12 B x |Omega| + own factor r/w + opposing factor once (tiles are re-read from L2)"))
        cands.append(dict(bound="l2", achieved=staged / t / 1e9, peak=L2_PEAK_GBS, unit="GB/s", per_launch=staged,
                          what="tile staging: workgroups x P x opposing factor bytes (L2 -> LDS)"))
        cands.append(dict(bound="lds", achieved=nnz * P * 8 * ld / t / 1e9, peak=LDS_PEAK_GBS, unit="GB/s", per_launch=nnz * P * 8 * ld,
                          what="LDS reads: one opposing vector (8 ld bytes) per update and pass",
                          # tools/ubench_lanerow.hip, profiles/r02_ubench_lanerow.txt: random 256-byte row reads by 4-lane groups and
                          # NOTHING else reach 63 TB/s with the padded rows (bank conflicts between the four groups of an LDS cycle)
                          # and 118 TB/s conflict-free; the sweeps turned out not to be bound by either (profiles/r02_rot_ab.txt)
                          measured_ceiling=dict(padded_rows=63000.0, conflict_free=118000.0, unit="GB/s",
                                                source="profiles/r02_ubench_lanerow.txt")))
    elif family == "cached":
        # csrc/glrm_cached.hip (regcached_sweep_kernel): the row's (index, value) list AND its opposing vectors are fetched ONCE per
        # half-step and kept in registers for the gradient pass and every line-search trial: P = 1 for this family.
        stream1 = nnz * 12 + 2 * nseg * ld * 8
        gathers = nnz * 8 * k
        if opp > MALL_BYTES:
            cands.append(dict(bound="hbm", achieved=(stream1 + gathers) / t / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", per_launch=stream1 + gathers,
                              what="12 B x |Omega| + own factor r/w + ONE k-vector gather per update from HBM (every pass reads it from registers)"))
        else:
            cands.append(dict(bound="hbm", achieved=(stream1 + opp) / t / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", per_launch=stream1 + opp,
                              what="compulsory HBM bytes: 12 B x |Omega| (one pass) + own factor r/w + opposing factor once"))
            cands.append(dict(bound="infinity_cache", achieved=gathers / t / 1e9, peak=MALL_GATHER_GBS, unit="GB/s", per_launch=gathers,
                              peak_is="MEASURED ceiling of random 8k-byte reads from a table that lives in the Infinity Cache (tools/ubench_gather.hip, "
                                      "profiles/r02_ubench_gather.txt: 8.2 TB/s at 512 B; MI355X_MICROARCH.md gives no spec bandwidth for that level)",
                              what="ONE k-vector gather per update served by the Infinity Cache (opposing factor %.0f MB <= 256 MiB)" % (opp / 1e6)))
    else:
        # 'gather' and 'blocked': every update fetches its k-vector from beyond the CU.  WHERE it comes from decides the price:
        #   gather, opposing factor beyond the Infinity Cache: random reads out of HBM -- SURVEY 8(d)'s algorithmic bytes over the HBM spec
        #   blocked (phase-aligned passes): all groups in flight read ONE window of the factor that was sized to stay in the Infinity
        #     Cache, so HBM sees the streams + the factor once per pass, and the gathers are cache traffic priced at the MEASURED ceiling
        #     of such reads (8.2 TB/s for random 512-byte reads out of the Infinity Cache against 6.5 TB/s from HBM,
        #     profiles/r02_ubench_gather.txt) -- the convention the cached row sweep already used (round 5 priced the
        #     same phenomenon at the HBM spec here and printed frac = 1.012 "of HBM")
        #   either, factor inside the Infinity Cache / L2: compulsory HBM bytes + gathers at the L2 peak
        mall_peak_is = ("MEASURED ceiling of random 8k-byte reads from a table that lives in the Infinity Cache (tools/ubench_gather.hip, "
                        "profiles/r02_ubench_gather.txt: 8.2 TB/s at 512 B; MI355X_MICROARCH.md gives no spec bandwidth for that level)")
        if family == "blocked" and opp > MALL_BYTES:
            comp = stream + P * opp
            cands.append(dict(bound="hbm", achieved=comp / t / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", per_launch=comp,
                              what="compulsory HBM bytes: P x 12 B x |Omega| + own factor r/w + the opposing factor streamed once per pass (window by window)"))
            cands.append(dict(bound="infinity_cache", achieved=nnz * P * 8 * k / t / 1e9, peak=MALL_GATHER_GBS, unit="GB/s", per_launch=nnz * P * 8 * k, peak_is=mall_peak_is,
                              what="k-vector gathers P x 8k per update, served by the Infinity Cache / L2 (phase-aligned passes: the window all groups read stays on chip)"))
        elif opp > MALL_BYTES or family == "general":  # the opposing factor cannot stay on chip: the gathers are HBM traffic
            cands.append(dict(bound="hbm", achieved=alg / t / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", per_launch=alg,
                              what="SURVEY 8(d) algorithmic bytes: P x (12 + 8k) per update (random k-vector gathers from HBM)"))
        else:  # the opposing factor fits the Infinity Cache / L2: HBM sees the streams, the gathers are cache traffic
            cands.append(dict(bound="hbm", achieved=(stream + opp) / t / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", per_launch=stream + opp,
                              what="compulsory HBM bytes: P x 12 B x |Omega| + own factor r/w + opposing factor once"))
            cands.append(dict(bound="l2", achieved=nnz * P * 8 * k / t / 1e9, peak=L2_PEAK_GBS, unit="GB/s", per_launch=nnz * P * 8 * k,
                              what="k-vector gathers served by L2 / Infinity Cache (opposing factor %.0f MB <= 256 MiB); priced at the "
                                   "L2 peak, the Infinity-Cache path behind it is slower" % (opp / 1e6)))
    for c in cands:
        c["frac"] = c["achieved"] / c["peak"]
    best = max(cands, key=lambda c: c["frac"])
    return dict(best=best, candidates=cands, algorithmic_GBps=alg / t / 1e9)


def side_summary(rl):
    """The best-priced limiter of one half-step for the `kernels` block: `frac` = achieved / peak of that limiter, always the same kind of
    number (never swapped for another definition on a threshold); `cache_served` marks a fraction that HBM alone could not deliver
    (> 0.9 of the 8 TB/s spec, whose measured copy ceiling is 6.3 TB/s): the caches served part of the bytes that were priced."""
    if not rl:
        return None
    b = rl["best"]
    out = {kk: b[kk] for kk in ("bound", "achieved", "peak", "unit", "frac")}
    out["cache_served"] = bool(b["bound"] == "hbm" and b["frac"] > CACHE_SERVED_ABOVE)
    return out


def roofline_block(rl, kernel_name, dom_ms, dom_nnz, traffic, traffic_src, l2_hits):
    """The `roofline` object of the JSON line for the dominant kernel.  ONE definition, whatever the timing noise does (VERDICT r4 weak 7:
    rounds 3-4 replaced `frac` by the PMC fraction whenever the algorithmic one reached 1.0, so two runs 1 % apart printed 0.995 and 0.876
    for the same kernel):
      frac          achieved / peak of the best-priced limiter (kernel_roofline) = that limiter's floor time / the measured time; every
                    limiter is priced where its bytes come from (round 6: the phase-aligned passes' gathers at the measured Infinity-Cache
                    gather ceiling like the cached row sweep's, no longer at the HBM spec), so it stays <= 1 by construction.
      algorithmic_frac  SURVEY 8(d) bytes P x (12 + 8k) x updates / launch time / 8 TB/s, kept beside it (may pass 1 when caches serve gathers)
      traffic_frac  what crossed the fabric (PMC: 2 x FETCH_SIZE + WRITE_SIZE of the kernel's launches) / launch time / 8 TB/s; None when
                    no PMC pass ran."""
    if not rl:
        return None
    best = rl["best"]
    gbps = traffic / (dom_ms * 1e-3) / 1e9 if traffic and dom_ms > 0 else None
    return {"bound": best["bound"], "kernel": kernel_name, "achieved": best["achieved"], "peak": best["peak"], "unit": best["unit"],
            "frac": best["frac"], "cache_served": bool(best["bound"] == "hbm" and best["frac"] > CACHE_SERVED_ABOVE),
            "traffic": traffic, "traffic_frac": gbps / HBM_PEAK_GBS if gbps is not None else None, "traffic_GBps": gbps, "traffic_source": traffic_src,
            "per_launch": best["per_launch"], "per_launch_is": best["what"], "updates_per_launch": dom_nnz, "avg_launch_ms": dom_ms,
            "candidates": [{kk: c[kk] for kk in ("bound", "achieved", "peak", "unit", "frac", "what")} for c in rl["candidates"]],
            "survey_8d_algorithmic_GBps": rl["algorithmic_GBps"], "algorithmic_frac": rl["algorithmic_GBps"] / HBM_PEAK_GBS, "l2": l2_hits,
            "frac_is": ("achieved / peak of the limiter named in `bound` (per_launch_is; the largest fraction among `candidates`) = that limiter's floor "
                        "time / measured time, the same definition in every run and family: compulsory HBM bytes at the 8 TB/s spec, cache-served "
                        "gathers at the measured 8.2 TB/s Infinity-Cache gather ceiling, LDS at 150 TB/s, flops at the rating; cache_served = an HBM-priced "
                        "fraction above %.1f of the spec, which HBM alone cannot deliver (6.29 TB/s measured copy ceiling); traffic_frac = PMC bytes across the fabric / launch "
                        "time / HBM peak; algorithmic_frac = SURVEY 8(d) bytes at P = 2 / launch time / HBM peak (above 1 for families that "
                        "re-use the opposing vectors on chip); durations are HIP events on the launch stream around every sweep of the timed "
                        "region" % CACHE_SERVED_ABOVE)}


def family_step_bytes(family, nnz, nseg, nopp, k, ld, hbm_floor=False):
    """Bytes one half-step of the family has to bring in from beyond the CU: the (index, value) stream once per pass, the own factor
    read and written, and the opposing k-vectors -- once per update and pass for the families that gather them (gather, phase-aligned
    passes, general sweeps), once per update for the cached row sweep, once per half-step for the LDS-tiled sweeps (the tiles are
    shared by the 256 segments of a workgroup and re-read from L2 by the others).
    hbm_floor=True: what of that HBM itself has to deliver -- an opposing factor that fits the 256 MiB Infinity Cache is counted once."""
    P = passes_priced(family)
    own = 2 * nseg * ld * 8
    opp = nopp * ld * 8
    if family in ("tiled", "lane") or (hbm_floor and opp <= MALL_BYTES):
        return P * 12 * nnz + own + opp
    return P * (12 + 8 * k) * nnz + own


def step_model(fam_r, fam_c, nnz_r, nnz_c, nseg_r, nseg_c, k, ld, ms_per_step, world, m=0, n=0):
    """The bytes ONE outer iteration of one rank has to move under the kernel families that ran it, over the measured time of the step.
    SURVEY.md 8(d) prices every half-step at P = 2 passes x (12 + 8k) B per update; the cached row sweep makes ONE pass (the row's list
    and vectors stay in registers for every trial) and the LDS-tiled sweeps fetch a k-vector once per workgroup instead of once per
    update, so with them the 8(d) figure is no lower bound -- the model the step actually obeys is published here so that the check can
    be redone from the JSON alone.  `GBps` counts every k-vector a kernel fetches from beyond its CU (cache-served ones too: at C4 the X
    half-step's gathers come out of the Infinity Cache); `hbm_floor` counts a cache-resident opposing factor once -- `within_peak` is the
    self-check on that floor."""
    bx = family_step_bytes(fam_r, nnz_r, nseg_r, n, k, ld)
    by = family_step_bytes(fam_c, nnz_c, nseg_c, m, k, ld)
    fx = family_step_bytes(fam_r, nnz_r, nseg_r, n, k, ld, hbm_floor=True)
    fy = family_step_bytes(fam_c, nnz_c, nseg_c, m, k, ld, hbm_floor=True)
    t = ms_per_step * 1e-3
    gbps = (bx + by) / t / 1e9
    floor = (fx + fy) / t / 1e9
    survey = (nnz_r + nnz_c) * 2 * (12 + 8 * k) / t / 1e9
    return {"passes": {"x": passes_priced(fam_r), "y": passes_priced(fam_c)}, "families": {"x": fam_r, "y": fam_c},
            "bytes_per_step_per_rank": {"x": bx, "y": by, "total": bx + by},
            "bytes_are": "per family (bench.py: family_step_bytes): stream + own factor r/w + opposing vectors per update and pass (gather / "
                         "phase-aligned / general), per update (cached rows) or per half-step (LDS-tiled); rank 0's shard",
            "GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBS,
            "hbm_floor": {"bytes": fx + fy, "GBps": floor, "frac_of_hbm_peak": floor / HBM_PEAK_GBS,
                          "is": "the same with an opposing factor that fits the 256 MiB Infinity Cache counted once",
                          "opposing_factor_cache_resident": {"x": bool(n * ld * 8 <= MALL_BYTES), "y": bool(m * ld * 8 <= MALL_BYTES)}},
            "within_peak": bool(floor <= HBM_PEAK_GBS),
            "survey_8d_P2_GBps": survey,
            "note": ("ms_per_step includes host round trips, the objective sum and (N > 1) the exchange; GBps above the HBM peak means the "
                     "gathers were served by the caches (small problems); survey_8d_P2_GBps (every update priced at 2 x (12 + 8k) B) "
                     "exceeds the peak whenever a family re-uses the opposing vectors on chip -- kept for comparison with earlier rounds only")}


# ----------------------------------------------------------------------------- CPU legs (rank 0, N = 1 only)

def _oracle_problem(ms, n, k, q, cfg, seed):
    import numpy as np
    import oracle as O
    from lowrankmodels.jl_amd import _capi, synth
    rowptr, colidx, rowvals, colptr, rowidx, colvals, X0, Y0 = O.synth_cpu(ms, n, k, q, seed=seed, value_model=cfg["value_model"],
                                                                            loss_mix=cfg["loss_mix"], transpose=True)
    if nonneg_start(cfg):
        X0, Y0 = np.asfortranarray(np.abs(X0) * (1.0 / k ** 0.5)), np.asfortranarray(np.abs(Y0) * (1.0 / k ** 0.5))
    reg = np.array([cfg["reg"]], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(ms, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, synth.loss_table(n, cfg["loss_mix"]), reg, reg)
    return pa, X0, Y0


def _time_oracle(pa, X0, Y0, cores, budget_s):
    """Outer iterations of the oracle on `pa`, X and Y half-steps timed apart.  Returns (iterations, seconds_x, seconds_y)."""
    import oracle as O
    api = O.oracle_api()
    O.set_threads(cores)
    h = api.create(pa)
    api.set_factors(h, X0, Y0)
    api.reset_stepsizes(h, 1.0)
    for _ in range(2):  # warm-up iterations: past the first line searches from the random start, like the GPU's warm-up
        api.step_x(h, 0.01); api.step_y(h, 0.01)
    iters, tx, ty, t0 = 0, 0.0, 0.0, time.time()
    while iters < 3 or (time.time() - t0 < budget_s and iters < 500):
        a = time.time(); api.step_x(h, 0.01)
        b = time.time(); api.step_y(h, 0.01)
        c = time.time()
        tx += b - a; ty += c - b
        iters += 1
    api.destroy(h)
    return iters, tx, ty


def cpu_baseline(args, cfg, k, q, n, m_full):
    """The oracle (CPU restatement of the reference, oracle/) timed on the host cores on TWO bounded samples of the same recipe, because no
    single affordable sample has both the rows and the columns of the benchmark problem:
      rows sample     the first `ms` rows x all n columns: rows as long as in the full problem (q observations, Y as large), columns ms/m as long
      columns sample  all m rows x the first `ns` columns (the first ns / (n / q) strata of the generator): columns as long as in the full
                      problem and X at its full size (5 GB at C4: not cache resident), rows ns / n as long
    `value` combines the X half-step rate of the rows sample with the Y half-step rate of the columns sample (harmonic: one update of
    each per observed entry and iteration); both samples' own rates are reported beside it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    cores = O.usable_cores()  # affinity mask and cgroup CPU quota, not the hardware thread count of the host
    target_obs = getattr(args, "cpu_target_obs", 0) or (2.5e6 if k <= 32 else 1.2e6) * cores  # ~10 s of CPU work per sample at the oracle's rate
    ms = int(min(max(args.cpu_sample_rows, target_obs / q), args.rows))
    ms = max(ms - ms % 8, 8)
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    pa, X0, Y0 = _oracle_problem(ms, n, k, q, cfg, args.seed)
    it_a, tx_a, ty_a = _time_oracle(pa, X0, Y0, cores, 8.0)
    obs_a = int(pa.rowptr[-1])
    del pa, X0, Y0
    rate_x_a, rate_y_a = it_a * obs_a / tx_a, it_a * obs_a / ty_a
    out = {"unit": "observed-entry updates/s", "cores": cores, "kind": "port",
           "rows_sample": {"sample": f"first {ms} rows x all {n} columns ({obs_a} observations, rank {k}), {it_a} outer iterations after 2 warm-up",
                           "x_halfstep_updates_per_s": rate_x_a, "y_halfstep_updates_per_s": rate_y_a,
                           "updates_per_s": it_a * 2 * obs_a / (tx_a + ty_a)}}
    rate_y = rate_y_a
    S = n // q                                   # stratum width of the generator: row e observes one column per stratum
    qs = min(q, max(1, int(round(target_obs / m_full))))
    if args.cpu_cols_sample and ms < m_full and m_full * k * 8 * 4 < 64e9:  # (ms == m_full: the rows sample already is the whole problem)
        ns = qs * S
        pb, Xb, Yb = _oracle_problem(m_full, ns, k, qs, cfg, args.seed)   # same Omega / values / start as the full problem on these columns
        it_b, tx_b, ty_b = _time_oracle(pb, Xb, Yb, cores, 8.0)
        obs_b = int(pb.rowptr[-1])
        del pb, Xb, Yb
        rate_y = it_b * obs_b / ty_b
        out["columns_sample"] = {"sample": f"all {m_full} rows x the first {ns} columns ({obs_b} observations, {m_full * q // n} per column as in the full "
                                           f"problem, X at full size {m_full * k * 8 / 1e9:.2f} GB), {it_b} outer iterations after 2 warm-up",
                                 "x_halfstep_updates_per_s": it_b * obs_b / tx_b, "y_halfstep_updates_per_s": rate_y,
                                 "updates_per_s": it_b * 2 * obs_b / (tx_b + ty_b)}
    out["value"] = 2.0 / (1.0 / rate_x_a + 1.0 / rate_y)
    out["sample"] = ("X half-step rate of the rows sample combined with the Y half-step rate of the "
                     + ("columns sample" if "columns_sample" in out else "rows sample (no columns sample was run)")
                     + " (2 / (1/rate_x + 1/rate_y)); OpenMP over rows then columns; the reference itself is Julia and cannot run here (no oracle/_ref)")
    return out


def jref_leg(args, cfg, api, device, fixture=None):
    """SURVEY.md 8(d), second leg of the metric: iterations and wall-clock until the GPU's recorded objective is <= J_ref (1 + 1e-5), where
    J_ref = ch.objective[end] of the CPU oracle running default ProxGradParams() to its OWN stop (src/algorithms/proxgrad.jl:210-213) on
    the same problem from the same X0, Y0; the GPU runs with the stop rule off.
    With a committed fixture (tests/golden/jref_<config>.json: a problem of the recipe with >= 1e8 observations, minutes of CPU time,
    run once by tools/make_jref.py) the problem is regenerated on the device from the same counter-based generator, and the whole
    trajectory and the stored factor samples are compared as well (`parity`); without one the oracle runs a small problem of the recipe
    here (cfg["jref"])."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from jref_tools import jref_device_problem, jref_parity, trajectory_deviation  # the checker's helpers (tests/jref_tools.py)
    from lowrankmodels.jl_amd.params import ProxGradParams
    k = cfg["k"]
    parity = None
    if fixture is not None:
        ms, n, q = fixture["m"], fixture["n"], fixture["q"]
        j_ref, it_cpu = float(fixture["J_ref"]), int(fixture["iterations_to_own_stop"])
        h, Xg, Yg = jref_device_problem(fixture, cfg, args.seed, api, device)
        cpu = {"cpu_iterations_to_own_stop": it_cpu, "cpu_seconds": fixture["cpu_seconds"], "cpu_cores": fixture["cpu_cores"],
               "cpu_where": fixture["cpu_where"], "fixture": f"tests/golden/jref_{fixture['config']}.json (tools/make_jref.py)",
               "cpu_objective_initial": fixture["objective"][0], "cpu_on_gpu_box": fixture.get("cpu_on_gpu_box")}
        nobs = fixture["observations"]
        try:
            parity = jref_parity(fixture, api, h, Xg, Yg)
        except Exception as e:  # the line must survive
            parity = {"error": repr(e)}
    else:
        import oracle as O
        ms, n, q = cfg["jref"]
        pa, X0, Y0 = _oracle_problem(ms, n, k, q, cfg, args.seed)
        cores = O.usable_cores()
        O.set_threads(cores)
        oapi = O.oracle_api()
        Xc, Yc = X0.copy(order="F"), Y0.copy(order="F")
        h = oapi.create(pa)
        t0 = time.time()
        obj_cpu, _ = oapi.fit(h, ProxGradParams(), Xc, Yc)
        t_cpu = time.time() - t0
        oapi.destroy(h)
        j_ref, it_cpu = float(obj_cpu[-1]), len(obj_cpu) - 1
        Xg, Yg = X0.copy(order="F"), Y0.copy(order="F")
        h = api.create(pa, device_id=device.index or 0)
        cpu = {"cpu_iterations_to_own_stop": it_cpu, "cpu_seconds": t_cpu, "cpu_cores": cores, "cpu_where": "this host, in this run"}
        nobs = int(pa.rowptr[-1])
        parity = {"vs_oracle_in_reference_order": {"cpu_iterations_to_own_stop": it_cpu}}
    prm_gpu = ProxGradParams(max_iter=max(it_cpu + 20, 30), abs_tol=-1e300, rel_tol=-1e300)  # stop rule off (no decrease is ever below these)
    t0 = time.time()
    obj_gpu, sec_gpu = api.fit(h, prm_gpu, Xg, Yg)
    t_gpu = time.time() - t0
    api.destroy(h)
    if fixture is None:
        parity["vs_oracle_in_reference_order"]["trajectory"] = trajectory_deviation(obj_gpu[: it_cpu + 1], obj_cpu)
    hit = np.flatnonzero(obj_gpu <= j_ref * (1 + 1e-5))
    it = int(hit[0]) if len(hit) else None
    out = {"problem": f"{ms} x {n}, rank {k}, {q} observations per row ({nobs} observed), same generator / losses / regularizers / start",
           "J_ref": j_ref, **cpu,
           "gpu_first_iteration_at_or_below_J_ref": it, "gpu_seconds_to_J_ref": float(sec_gpu[it]) if it is not None else None,
           "gpu_objective_there": float(obj_gpu[it]) if it is not None else None, "gpu_objective_initial": float(obj_gpu[0]),
           "gpu_objective_at_cpu_stop_iteration": float(obj_gpu[min(it_cpu, len(obj_gpu) - 1)]),
           "gpu_ms_per_iteration": 1e3 * float(sec_gpu[-1]) / max(len(sec_gpu) - 1, 1), "gpu_fit_wall_s_incl_transfers": t_gpu,
           "rule": "first GPU iteration with objective <= J_ref (1 + 1e-5); J_ref = oracle, default ProxGradParams(), own stop rule",
           "max_rel_dev_over_trajectory": (parity or {}).get("vs_oracle_in_reference_order", {}).get("trajectory", {}).get("max_rel"),
           "parity": parity}
    if cpu.get("cpu_seconds") and it is not None and sec_gpu[it] > 0:
        same_box = cpu.get("cpu_where", "").startswith("this host")
        out["speedup_to_J_ref_vs_cpu" if same_box else "cross_box_ratio_cpu_seconds_over_gpu_seconds"] = cpu["cpu_seconds"] / float(sec_gpu[it])
        gb = cpu.get("cpu_on_gpu_box")
        if not same_box and gb:
            # the same CPU run, timed on the cores of a GPU box of this pool in an earlier session (bit-identical trajectory): the CPU leg of
            # "wall-clock to reference convergence" on the box class the GPU number comes from
            out["speedup_to_J_ref_vs_cpu_on_a_gpu_box"] = {"ratio": gb["cpu_seconds"] / float(sec_gpu[it]), "cpu_seconds": gb["cpu_seconds"], "cores": gb["cores"],
                                                           "gpu_seconds": float(sec_gpu[it]), "cpu_measured": gb["where"]}
        if not same_box:
            out["cross_box_note"] = ("the CPU seconds were measured on the 8 cores of the build container when the fixture was made, the GPU seconds on this box: "
                                     "not a same-box speed-up (cpu_baseline is the same-box CPU rate)")
    return out


# ----------------------------------------------------------------------------- PMC traffic (rank 0, N = 1 only)

def pmc_traffic(args, kernel_re, per_halfstep=False, child_env=None, eval_pass=False):
    """HBM-side bytes per launch of the dominant kernel from rocprofv3 PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
    WRITE_SIZE in SEPARATE passes (TCC slots), FETCH_SIZE doubled on gfx950 (128-B requests tallied at 64 B); both counters are in KiB.
    Each pass re-runs this script as a child (same config, 2 timed steps) under `rocprofv3 --pmc <counter>`; the mean over the
    dispatches of the kernel in that child is used.  A third pass collects TCC_HIT_sum / TCC_MISS_sum: the L2 hit rate of the kernel's
    requests (where the bytes that did not cross the fabric came from).  Returns (bytes per launch or None, note, L2 dict or None)."""
    rp = shutil.which("rocprofv3")
    if rp is None:
        return None, "rocprofv3 not found", None
    out = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum TCC_MISS_sum"):
        d = tempfile.mkdtemp(prefix="glrm_pmc_", dir="/tmp")
        cmd = [rp, "--pmc", *ctr.split(), "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.join(ROOT, "bench.py"),
               "--config", args.config, "--rows", str(args.rows), "--steps", "2", "--warmup", str(max(args.warmup, 1)), "--tiled", str(args.tiled), *(["--quad-gram"] if args.quad_gram else []),
               "--no-cpu-baseline", "--no-convergence-run", "--no-jref", "--no-other-configs", "--no-create-from-host", "--pmc", "off", "--seed", str(args.seed), "--borrow", args.borrow,
               "--cols", str(args.cols), "--obs-per-row", str(args.obs_per_row), "--rank", str(args.k), "--degree", args.degree, "--zipf-s", str(args.zipf_s)]
        env = dict(os.environ, TMPDIR="/tmp", **(child_env or {}))
        try:
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
            try:
                p.communicate(timeout=args.pmc_timeout)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, 9)
                p.communicate()
                if ctr.startswith("TCC_HIT"):
                    out["L2"] = None
                    continue
                return None, f"PMC pass {ctr} timed out after {args.pmc_timeout} s", None
            vals, hits, misses = [], 0.0, 0.0
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(path)):
                    if not re.search(kernel_re, r.get("Kernel_Name", "")):
                        continue
                    if r.get("Counter_Name") == ctr:
                        vals.append(float(r["Counter_Value"]))
                    elif r.get("Counter_Name") == "TCC_HIT_sum":
                        hits += float(r["Counter_Value"])
                    elif r.get("Counter_Name") == "TCC_MISS_sum":
                        misses += float(r["Counter_Value"])
            if ctr.startswith("TCC_HIT"):  # L2 hit rate of the kernel's requests (MI355X_MICROARCH.md: TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum))
                out["L2"] = {"hits": hits, "misses": misses, "hit_rate": hits / (hits + misses) if hits + misses > 0 else None}
                continue
            if not vals:
                return None, f"PMC pass {ctr}: no dispatch matching /{kernel_re}/ in the counter CSV (exit {p.returncode})", None
            # one-kernel sweeps: mean over the dispatches; pass families (several launches per half-step): total over the child's
            # dispatches / its half-steps (warm-up + 2 timed; the initial objective evaluation adds one gradient-type pass for columns)
            # The column side also runs ONE gradient-type pass outside the half-steps (the evaluation of the initial objective, col_losses):
            # half of a two-pass half-step's traffic.  Round 3 divided by the half-steps alone and over-stated the per-half-step traffic
            # of the column pass families by 10 % (5 half-steps) -- the C4 line's "0.94 of the HBM peak" was 0.86.
            halfsteps = max(args.warmup, 1) + 2 + (0.5 if eval_pass else 0.0)
            out[ctr] = (sum(vals) / (halfsteps if per_halfstep else len(vals)), len(vals))
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch, write = out["FETCH_SIZE"][0] * 1024.0 * 2.0, out["WRITE_SIZE"][0] * 1024.0
    return fetch + write, _pmc_note(out, per_halfstep, kernel_re), out.get("L2")


def _pmc_note(out, per_halfstep, kernel_re):
    return (f"in-run rocprofv3 --pmc passes of this config: 2 x FETCH_SIZE ({out['FETCH_SIZE'][0]:.4g} KiB, gfx950 correction) + "
                           f"WRITE_SIZE ({out['WRITE_SIZE'][0]:.4g} KiB), {'total of' if per_halfstep else 'mean over'} {out['FETCH_SIZE'][1]} dispatches matching /{kernel_re}/{' divided by the half-steps of the child run (+ 0.5 for the one-pass evaluation of the initial objective on the column side)' if per_halfstep else ''}")


# ----------------------------------------------------------------------------- one rank's shard geometry on one GPU

# xGMI on an 8-GPU MI355X node: every GPU has 7 links, one per peer.  MI355X_MICROARCH.md gives no xGMI figure; the task brief quotes
# "7 links x ~153 GB/s per GPU", which is AMD's per-link number counted in BOTH directions (153.6 GB/s = 2 x 76.8 GB/s).  A block that one
# GPU pushes to a peer travels ONE direction of one link, so the model's default is the per-direction figure; the bidirectional number is
# printed beside it as the optimistic bound round 3 used.  Nothing here is measured: no multi-GPU node has been available to this build.
XGMI_LINK_GBS_PER_DIRECTION = 76.8
XGMI_LINK_GBS_BIDIRECTIONAL = 153.6


def exchange_model_ms(block_bytes, n, link_gbs=XGMI_LINK_GBS_PER_DIRECTION):
    """Time for every rank to publish its block to its n - 1 peers.  direct: the owner pushes the block over all its links at once (one
    link per peer carries one block in one direction).  ring: the block travels n - 1 hops, one link busy per step (what a ring
    all-gather does).  busbw_equivalent: the all-gather bus bandwidth (n - 1) / n x total bytes / time an RCCL test would print for `direct`."""
    if n <= 1:
        return {"direct": 0.0, "ring": 0.0}
    one = block_bytes / (link_gbs * 1e9) * 1e3
    return {"direct": one, "ring": one * (n - 1), "link_GBps_one_direction": link_gbs,
            "busbw_equivalent_GBps_direct": (n - 1) * block_bytes / (one * 1e-3) / 1e9,
            "optimistic_direct_if_153GBps_were_per_direction": block_bytes / (XGMI_LINK_GBS_BIDIRECTIONAL * 1e9) * 1e3}


def emulate_rank(args):
    """No multi-GPU node is needed to know what ONE rank of the sharded fit computes per half-step: this builds rank r's shard of the
    N-way problem (strong scaling: m/N rows and n/N columns of the full problem, X and Y fully replicated, the kernel families chosen
    from the signature of the WHOLE problem exactly as the N-rank job would) on one GPU and times step_x / step_y with HIP events.
    What it cannot measure is the exchange: that is modelled (exchange_model_ms) and printed beside the measurement.  The other
    ranks' blocks of X and Y are never updated here, so the objective is not the job's -- only the kernels' work is."""
    import torch
    from lowrankmodels.jl_amd import _capi, synth
    from lowrankmodels.jl_amd.fit import ShardedFit
    N, r = args.of, args.emulate_rank
    if N < 1 or not 0 <= r < N:
        raise SystemExit("--emulate-rank r needs --of N with 0 <= r < N")
    if args.config == "C3":
        raise SystemExit("--emulate-rank covers the list configs (C2, C4, C5)")
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    cfg = dict(CONFIGS[args.config])
    if args.cols or args.obs_per_row or args.k:
        cfg.update(cols=args.cols or cfg["cols"], q=args.obs_per_row or cfg["q"], k=args.k or cfg["k"])
    k, q, n, reg = cfg["k"], cfg["q"], cfg["cols"], cfg["reg"]
    m = args.rows or cfg["rows"]
    if m % N or n % N or n % q:
        raise SystemExit("rows and cols must be divisible by the number of shards, cols by the observations per row")
    rbs = [m // N * i for i in range(N + 1)]
    cbs = [n // N * i for i in range(N + 1)]
    api = _capi.hip_api()
    w = synth.DeviceWorkload(m, n, k, q, rows=(rbs[r], rbs[r + 1]), cols=(cbs[r], cbs[r + 1]), seed=args.seed,
                             value_model=cfg["value_model"], loss_mix=cfg["loss_mix"], rx=reg, ry=reg, device=device)
    whole = w.whole_signature()
    sf = ShardedFit(api, w.problem(), rbs, cbs, device=device, stream=torch.cuda.current_stream().cuda_stream,
                    opts=dict(profile=1, waves_row=args.waves_row, waves_col=args.waves_col, tiled=args.tiled), x_chunks=1,
                    whole_signature=whole)
    nnz_r, nnz_c = w.nnz_rows, w.nnz_cols
    w.free_sources()
    X0, Y0 = w.init_factors(sf.ld)
    if nonneg_start(cfg):
        X0.abs_().mul_(1.0 / k ** 0.5); Y0.abs_().mul_(1.0 / k ** 0.5)
    sf.dX.copy_(X0); sf.dY.copy_(Y0)
    del X0, Y0
    api.reset_stepsizes(sf.h, 1.0)

    class P:
        stepsize, inner_iter_X, inner_iter_Y, min_stepsize = 1.0, 1, 1, 0.01

    for _ in range(args.warmup):
        sf.iteration(P)
    api.kernel_stats(sf.h, reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sf.iteration(P)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.steps * 1e3
    st = api.kernel_stats(sf.h)
    sf.close()
    ms_x, ms_y = st["ms_x"] / args.steps, st["ms_y"] / args.steps
    flags, ld = st["tiled"], st["ld"]
    fam_r = "lane" if flags & 256 else "tiled" if flags & 1 else "cached" if flags & 64 else "blocked" if flags & 16 else "gather"
    fam_c = "lane" if flags & 512 else "tiled" if flags & 2 else "blocked" if flags & 32 else "gather"
    ex_x = exchange_model_ms((m // N) * ld * 8, N)
    ex_y = exchange_model_ms((n // N) * ld * 8 + (n // N) * 8, N)
    bytes_x = nnz_r * (12 + 8 * k) * (1 if fam_r == "cached" else 2)
    bytes_y = nnz_c * (12 + 8 * k) * 2
    out = {"mode": "emulate-rank", "rank": r, "of": N, "config": args.config, "m": m, "n": n, "k": k, "shard_rows": m // N, "shard_cols": n // N,
           "shard_observations": {"rows": nnz_r, "cols": nnz_c}, "steps": args.steps, "warmup": args.warmup,
           "measured_ms": {"step_x": ms_x, "step_y": ms_y, "iteration_wall_incl_host": wall},
           "families": {"row_sweep": fam_r, "col_sweep": fam_c, "flags": flags, "waves_row": st["waves_row"], "waves_col": st["waves_col"]},
           "mean_trials": {"per_row": st["trials_x"] / max(args.steps * (m // N), 1), "per_col": st["trials_y"] / max(args.steps * (n // N), 1)},
           "algorithmic_GBps": {"step_x": bytes_x / (ms_x * 1e-3) / 1e9 if ms_x > 0 else None, "step_y": bytes_y / (ms_y * 1e-3) / 1e9 if ms_y > 0 else None,
                                "passes_priced": {"step_x": 1 if fam_r == "cached" else 2, "step_y": 2}},
           "exchange_model_ms": {"X_block": ex_x, "Y_block_and_objectives": ex_y, "link_GBps_one_direction": XGMI_LINK_GBS_PER_DIRECTION,
                                 "note": "MODELLED, not measured: no multi-GPU node was available; direct = the owner pushes its block over its 7 "
                                         "links at once (one direction of each: 76.8 GB/s = half of AMD's bidirectional 153.6 GB/s per link), "
                                         "ring = n - 1 hops over one link"},
           "predicted_iteration_ms": {"direct_no_overlap": ms_x + ms_y + ex_x["direct"] + ex_y["direct"],
                                      "ring_no_overlap": ms_x + ms_y + ex_x["ring"] + ex_y["ring"]},
           "predicted_updates_per_s_all_ranks": {"direct_no_overlap": 2.0 * m * q / ((ms_x + ms_y + ex_x["direct"] + ex_y["direct"]) * 1e-3),
                                                 "ring_no_overlap": 2.0 * m * q / ((ms_x + ms_y + ex_x["ring"] + ex_y["ring"]) * 1e-3)},
           "whole_signature": dict(zip(("nnz_rows", "nnz_cols", "max_row_len", "max_col_len", "rows_unordered", "cols_unordered"), whole.astuple()))}
    print(json.dumps(out), flush=True)


# ----------------------------------------------------------------------------- the host a Julia fit! binds: ONE process, N devices

def host_arrays_from_device(w):
    """The generated Omega views copied to host numpy arrays (what glrm_hip_multi_create and the CPU oracle take)."""
    import numpy as np
    f = lambda t, dt: np.ascontiguousarray(t.cpu().numpy().astype(dt, copy=False))
    return (f(w.rowptr, np.int64), f(w.colidx[: w.nnz_rows], np.int32), f(w.rowvals[: w.nnz_rows], np.float64),
            f(w.colptr, np.int64), f(w.rowidx[: w.nnz_cols], np.int32), f(w.colvals[: w.nnz_cols], np.float64))


def host_problem(args, cfg, m, n, k, q, device):
    """The whole bench problem as host arrays + start: generated on `device` (seconds) and copied over PCIe once."""
    import numpy as np
    from lowrankmodels.jl_amd import _capi, synth
    reg = cfg["reg"]
    w = synth.DeviceWorkload(m, n, k, q, seed=args.seed, value_model=cfg["value_model"], loss_mix=cfg["loss_mix"], rx=reg, ry=reg, device=device)
    arrs = host_arrays_from_device(w)
    dX, dY = w.init_factors(k)  # ld = k: the host layout of the C ABI (k x m, k x n, column-major)
    w.free_sources()
    if nonneg_start(cfg):
        dX.abs_().mul_(1.0 / k ** 0.5); dY.abs_().mul_(1.0 / k ** 0.5)
    X0 = np.asfortranarray(dX.cpu().numpy().reshape(m, k).T)
    Y0 = np.asfortranarray(dY.cpu().numpy().reshape(n, k).T)
    del dX, dY
    r = np.array([reg], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(m, n, k, *arrs, synth.loss_table(n, cfg["loss_mix"]), r, r)
    return pa, X0, Y0


def inlib_host(args):
    """bench.py --host inlib --gpus N: the fit a Julia `fit!(glrm, HipProxGradParams(ngpus = N))` ccalls -- glrm_hip_multi_create /
    glrm_hip_multi_fit (csrc/glrm_multigpu.hip), ONE host process driving N devices, the library sharding the host problem, one host
    thread per shard, blocks exchanged by direct peer pushes or RCCL -- on the same problem as the torch.distributed host.  ONE
    multi_fit call of warmup + steps outer iterations with the stop rule off; the K timed steps are iterations warmup+1 .. warmup+K on the
    library's own per-iteration clock (the `seconds` array = ch.times, src/convergence.jl:22-26: every iteration ends with a device
    synchronisation on every shard), so the factors' trip over PCIe at the start and end of the call is outside the timed region like
    in the other host.  --shared-device: all N shards on device 0 (a box with one GPU: the code path, not the speed)."""
    if args.shared_device:
        # N compute streams + N link streams (+ the engine's side streams) on ONE device: the runtime maps streams onto GPU_MAX_HW_QUEUES
        # hardware queues (default 4) and a queue standing in an emulated link wait would hold up every stream that shares it
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(min(32, 2 * args.gpus + 8)))
    import numpy as np
    import torch
    from lowrankmodels.jl_amd import _capi
    from lowrankmodels.jl_amd.params import ProxGradParams
    N = args.gpus
    cfg = dict(CONFIGS[args.config])
    if args.config == "C3":
        raise SystemExit("--host inlib covers the list configs")
    # the link emulator exists in the TEST BUILD of the engine only (libglrm_hip_testing.so): a line with --emulate-link-gbps is a schedule
    # experiment and says which library ran it; every other line runs the product library
    get_api = _capi.hip_testing_api if args.emulate_link_gbps > 0 else _capi.hip_api
    if args.cols or args.obs_per_row or args.k:
        cfg.update(cols=args.cols or cfg["cols"], q=args.obs_per_row or cfg["q"], k=args.k or cfg["k"])
    k, q, n = cfg["k"], cfg["q"], cfg["cols"]
    m = args.rows or cfg["rows"]
    ndev = torch.cuda.device_count()
    if not args.shared_device and ndev < N:
        raise SystemExit(f"--host inlib --gpus {N}: {ndev} device(s) visible (use --shared-device to put every shard on device 0)")
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    api = get_api()
    t0 = time.time()
    pa, X0, Y0 = host_problem(args, cfg, m, n, k, q, device)
    t_gen = time.time() - t0
    torch.cuda.empty_cache()
    ids = [0] * N if args.shared_device else list(range(N))
    prm = ProxGradParams(max_iter=args.warmup + args.steps, abs_tol=-1e300, rel_tol=-1e300)
    nnz = int(pa.rowptr[-1])

    def one_fit(link_gbps):
        """ONE create + multi_fit with the link emulator at link_gbps GB/s per direction (0 = off: copies take what they take)."""
        if link_gbps > 0:
            os.environ["GLRM_EXCHANGE_EMULATE_GBPS"] = repr(float(link_gbps))
        else:
            os.environ.pop("GLRM_EXCHANGE_EMULATE_GBPS", None)
        t0 = time.time()
        mh = api.multi_create(pa, N, device_ids=ids, exchange=args.inlib_exchange, x_chunks=args.x_chunks if N > 1 else 0, profile=1,
                              waves_row=args.waves_row, waves_col=args.waves_col, tiled=args.tiled, arrival=args.arrival)
        t_create = time.time() - t0
        try:
            X, Y = np.array(X0, order="F"), np.array(Y0, order="F")
            t0 = time.perf_counter()
            obj, sec = api.multi_fit(mh, prm, X, Y)
            wall = time.perf_counter() - t0
            info = api.multi_info(mh, N)
        finally:
            api.multi_destroy(mh)
            os.environ.pop("GLRM_EXCHANGE_EMULATE_GBPS", None)
        assert len(sec) == args.warmup + args.steps + 1, len(sec)
        return {"ms_per_step": 1e3 * float(sec[-1] - sec[args.warmup]) / args.steps, "obj": obj, "info": info, "wall": wall, "create_s": t_create}

    run = one_fit(args.emulate_link_gbps)
    base = one_fit(0.0) if args.emulate_link_gbps > 0 else None  # the same fit with free copies: what the link adds is the difference
    obj, info, elapsed = run["obj"], run["info"], run["ms_per_step"] * 1e-3 * args.steps
    host = {"kind": "in-library (glrm_hip_multi_create / glrm_hip_multi_fit): what julia/HipGLRM.jl ccalls for HipProxGradParams(ngpus = N)",
            "library": "libglrm_hip_testing.so (test build: link emulator)" if args.emulate_link_gbps > 0 else "libglrm_hip.so",
            "exchange_used": {0: "direct peer pushes (hipMemcpyPeerAsync, one copy stream per (source, destination) pair)", 1: "RCCL ncclAllGather / grouped broadcasts"}.get(info["exchange"], info["exchange"]),
            "arrival_order": {0: "on (default)", 1: "on", 2: "off: the Y half-step waits for the whole X exchange"}[args.arrival] + (
                "" if args.arrival == 2 else ": the Y half-step consumes the peers' row chunks of X as their copy events fire (glrm_hip_step_y_arrival)"),
            "exchange_ms_per_step_exposed": info["exchange_ms"] / max(args.warmup + args.steps, 1),
            "exchange_ms_is": "time the compute streams stood waiting for blocks that had not arrived (Y-block exchange: end of the sweeps to the last "
                              "arrival; X: the waits inside the arrival-ordered Y half-step), max over shards, summed over the call, per iteration.  In TRUE arrival "
                              "order (the default on the phase-aligned column passes) a super-tile is only enqueued once its blocks have arrived: the stream "
                              "never stands in a wait, the host thread polls instead; since round 6 the time it polled with NO super-tile ready is booked in this figure too "
                              "(an upper bound on the device's idle time) -- link_emulation's A/B stays the direct measure",
            "row_bounds": info["row_bounds"], "col_bounds": info["col_bounds"], "x_chunks": args.x_chunks if N > 1 else 0,
            "shared_device": bool(args.shared_device),
            "timed_region": "iterations warmup+1 .. warmup+steps of ONE glrm_hip_multi_fit call on the library's per-iteration clock (ch.times)",
            "whole_call_wall_s_incl_factor_transfers_and_prologue": run["wall"],
            "model_ms": {"X_block": exchange_model_ms((m // N) * k * 8, N), "Y_block": exchange_model_ms((n // N) * k * 8, N)}}
    if base is not None:
        share = max(ids.count(d) for d in set(ids))
        dil = int(os.environ.get("GLRM_EXCHANGE_EMULATE_DILATE", share))
        host["link_emulation"] = {
            "GBps_per_direction_and_link": args.emulate_link_gbps, "dilation": dil,
            "is": "every direct push occupies its link for bytes / (rate / dilation) from the moment its source rows were complete "
                  "(csrc/glrm_multigpu.hip: LinkEmu); dilation = shards per device: they time-share one GPU, so compute is that many times "
                  "slower than on as many GPUs and a transfer is slowed alike to keep its proportion to the compute",
            "ms_per_step_with_link": run["ms_per_step"], "ms_per_step_with_free_copies": base["ms_per_step"],
            "exchange_exposed_ms_per_step": run["ms_per_step"] - base["ms_per_step"],
            "exchange_exposed_frac_of_step": (run["ms_per_step"] - base["ms_per_step"]) / run["ms_per_step"],
            "same_objective_bits_both_runs": bool(np.array_equal(run["obj"], base["obj"])),
            "X_block_link_ms_undilated": exchange_model_ms((m // N) * k * 8, N, args.emulate_link_gbps)["direct"],
            "what_it_cannot_show": "N shards on ONE device do not run in lockstep like N devices do (the hardware interleaves their kernels), and "
                                   "their copies share one HBM: the figure is the exchange the schedule leaves exposed, not a prediction of an "
                                   "8-GPU iteration time"}
    out = {"metric": "observed-entry updates/sec", "value": args.steps * 2.0 * nnz / elapsed, "unit": "updates/s", "n_gpus": N, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "ranks_seen": {"world_size": N, "backend": {0: "direct peer pushes (hipMemcpyPeerAsync)", 1: "RCCL (in-library communicators)"}.get(info["exchange"], info["exchange"]),
                          "devices": [{"rank": r_, "device": d_} for r_, d_ in enumerate(ids)], "distinct_devices": len(set(ids)),
                          "visible_devices_per_rank": ndev, "host": "one process, N devices (glrm_hip_multi_fit: one host thread per shard)"},
           "config": {"workload": cfg["text"].format(m=m, n=n, k=k, pct=100.0 * q / n) + f" ({nnz} observations), ProxGradParams defaults, stop rule off",
                      "name": args.config, "m": m, "n": n, "k": k, "observed": nnz, "host": "inlib",
                      "parallelism": f"glrm_hip_multi_fit: ONE host process, rows/cols in {N} nnz-balanced blocks on devices {ids}, X,Y replicated"},
           "host": host,
           "objective": {"initial": float(obj[0]), "after_warmup_and_steps": float(obj[-1])},
           "setup_s": {"generate_and_copy_to_host": t_gen, "multi_create": run["create_s"]}}
    print(json.dumps(out), flush=True)
    return out


def visible_devices():
    """HIP devices this process would see, without importing torch (the self-launching parent never touches a device)."""
    import ctypes
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            n = ctypes.c_int(0)
            if ctypes.CDLL(name).hipGetDeviceCount(ctypes.byref(n)) == 0:
                return int(n.value)
            return 0
        except OSError:
            continue
    return 0


def ranks_seen_block(dist, torch, rank, dev_index):
    """`ranks_seen` of the JSON line: what the process group itself reports -- world size and backend from
    torch.distributed, and for every rank the device it computes on (index and PCI address, all-gathered), so that a line claiming N GPUs
    shows N distinct devices.  dist = None: one rank, no process group."""
    p = torch.cuda.get_device_properties(dev_index)
    mine = [rank, dev_index, int(getattr(p, "pci_domain_id", -1)), int(getattr(p, "pci_bus_id", -1)), int(getattr(p, "pci_device_id", -1))]
    if dist is None:
        rows, world, backend = [mine], 1, None
    else:
        world, backend = dist.get_world_size(), dist.get_backend()
        t = torch.tensor(mine, dtype=torch.int64, device=torch.device("cuda", dev_index))
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        rows = [[int(v) for v in o.tolist()] for o in out]
    devs = [{"rank": r[0], "device": r[1], "pci": "%04x:%02x:%02x" % (r[2] & 0xffff, r[3] & 0xff, r[4] & 0xff)} for r in rows]
    return {"world_size": world, "backend": {"nccl": "nccl (RCCL)"}.get(backend, backend), "devices": devs,
            "distinct_devices": len({d["pci"] for d in devs}), "device_name": p.name, "visible_devices_per_rank": torch.cuda.device_count(),
            "host": "one process per GPU (torch.distributed)"}


def self_launch(args, argv):
    """`python bench.py --gpus N` WITHOUT a launcher (the driver's plain command died with SystemExit before touching a
    device).  The parent spawns N ranks of this very command under torch.distributed.run on 127.0.0.1 (one process per GPU over RCCL; the
    ranks take the launcher path and rank 0 prints the line), relays that line with how it was launched, and exits 0.  If the N-rank job
    cannot run -- fewer devices than ranks on the RCCL backend, a rendezvous or RCCL failure, a hang (--launch-timeout) -- the in-library host
    (ONE process, N devices, direct peer pushes: bench.py --host inlib, what julia/HipGLRM.jl ccalls) runs the same problem instead and its
    line is printed with `fell_back_from`.  Only if both fail the line carries `error` and value null, and the exit code is 1."""
    import socket
    N = args.gpus
    backend = os.environ.get("GLRM_BENCH_BACKEND", "nccl")
    ndev = visible_devices()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={N}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), *argv]
    launch = {"how": "self-launched: `bench.py --gpus %d` without WORLD_SIZE in the environment re-ran itself under torch.distributed.run" % N,
              "command": "python -m torch.distributed.run --nnodes=1 --nproc-per-node=%d --master-addr 127.0.0.1 --master-port %d bench.py %s" % (N, port, " ".join(argv)),
              "visible_devices": ndev, "backend": backend}
    why = None
    if backend == "nccl" and ndev < N:
        why = {"error": f"{ndev} device(s) visible, {N} ranks asked for: RCCL needs one device per rank (GLRM_BENCH_BACKEND=gloo lets ranks share devices: plumbing only)"}
    else:
        t0 = time.time()
        env = dict(os.environ, GLRM_BENCH_SELF_LAUNCHED="1")
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            so, se = p.communicate(timeout=args.launch_timeout)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, 9)
            so, se = p.communicate()
            why = {"error": f"the {N}-rank job was killed after --launch-timeout {args.launch_timeout} s", "stderr_tail": (se or "")[-1500:]}
        lines = [ln for ln in (so or "").splitlines() if ln.startswith("{")]
        if why is None and p.returncode == 0 and lines:
            out = json.loads(lines[-1])
            out["launch"] = dict(launch, child_wall_s=time.time() - t0)
            print(json.dumps(out), flush=True)
            return out
        if why is None:
            why = {"error": f"the {N}-rank job exited with {p.returncode}" + ("" if lines else " and printed no line"), "stderr_tail": (se or "")[-1500:]}
    sys.stderr.write("bench.py: %s -- falling back to the in-library host\n" % why["error"])
    shared = ndev < N and ndev >= 1 and backend != "nccl"  # the plumbing mode of a box with fewer GPUs than ranks
    if ndev >= N or shared:
        r = inlib_child(args, N, timeout_s=max(420, args.launch_timeout // 2), shared_device=shared)
        if "error" not in r:
            r["launch"] = dict(launch, how=launch["how"] + "; the N-rank job did not produce a line, the in-library host ran instead")
            r["fell_back_from"] = why
            print(json.dumps(r), flush=True)
            return r
        why = {"n_rank_job": why, "in_library_host": r}
    out = {"metric": "observed-entry updates/sec", "value": None, "unit": "updates/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
           "higher_is_better": True, "error": why, "launch": launch}
    print(json.dumps(out), flush=True)
    raise SystemExit(1)


def inlib_child(args, n_gpus, timeout_s=420, shared_device=False):
    """`bench.py --host inlib --gpus N` as a child process with a time limit (the first multi-GPU run of a path that has only ever run
    with its shards on one device must not take the job's line with it): returns the child's JSON line, or what went wrong."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--host", "inlib", "--gpus", str(n_gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--config", args.config, "--rows", str(args.rows), "--cols", str(args.cols), "--obs-per-row", str(args.obs_per_row), "--rank", str(args.k),
           "--seed", str(args.seed), "--tiled", str(args.tiled), "--x-chunks", str(args.x_chunks), "--waves-row", str(args.waves_row), "--waves-col", str(args.waves_col),
           "--arrival", str(args.arrival), "--emulate-link-gbps", str(args.emulate_link_gbps)]
    if shared_device:
        cmd.append("--shared-device")
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                                                               "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    t0 = time.time()
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {timeout_s} s", "command": " ".join(cmd[1:])}
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": f"exit {p.returncode}", "stderr_tail": p.stderr[-1500:], "command": " ".join(cmd[1:])}
    r = json.loads(lines[-1])
    r["child_wall_s"] = time.time() - t0
    return r


def other_config_line(args, name, extra=(), timeout_s=600):
    """One of the other BASELINE configs under the same clock as the default line (C2 / C3 / C5 were builder-run
    profiles only): `bench.py --config <name>` as a child process (its own device memory: C5 at its stated size needs ~200 GB), compacted
    to what the tables quote -- ms per iteration, updates/s, kernel families, the roofline of its dominant kernel with PMC traffic."""
    steps = min(args.steps, 10)
    # the second leg of the metric (iterations / seconds to J_ref) rides along for the list configs whose fixture is committed (C2, C5:
    # tests/golden/jref_<config>.json, ~1e8 observations, a few seconds of GPU time after the timed region has released its memory)
    jref = [] if name in ("C2", "C5") and not args.no_jref else ["--no-jref"]
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", name, "--steps", str(steps), "--warmup", str(args.warmup), "--seed", str(args.seed),
           "--no-cpu-baseline", *jref, "--no-convergence-run", "--no-other-configs", "--pmc", args.pmc, "--pmc-timeout", str(args.pmc_timeout), *extra]
    t0 = time.time()
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {timeout_s} s", "command": " ".join(cmd[1:])}
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": f"exit {p.returncode}", "stderr_tail": p.stderr[-800:], "command": " ".join(cmd[1:])}
    r = json.loads(lines[-1])
    rf, kn = r.get("roofline") or {}, r.get("kernels") or {}
    tr = r.get("to_ref_objective")
    if isinstance(tr, dict) and "error" not in tr:
        par = tr.get("parity") or {}
        tr = {kk: tr.get(kk) for kk in ("problem", "J_ref", "fixture", "cpu_iterations_to_own_stop", "cpu_seconds", "cpu_cores", "gpu_first_iteration_at_or_below_J_ref",
                                        "gpu_seconds_to_J_ref", "gpu_objective_there", "gpu_ms_per_iteration", "max_rel_dev_over_trajectory", "rule")}
        tr["gpu_iterations_to_own_stop"] = par.get("gpu_iterations_to_own_stop")
        tr["factor_samples_vs_reference_order"] = {kk: (par.get("vs_oracle_in_reference_order") or {}).get(kk) for kk in ("X_sample_rel_fro", "Y_sample_rel_fro")}
    return {"workload": r["config"]["workload"], "to_ref_objective": tr, "full_size": r["config"].get("full_size"), "ms_per_step": r["ms_per_step"], "updates_per_s": r["value"],
            "steps": r["steps"], "warmup": r["warmup"], "families": {"row_sweep": r["config"].get("row_sweep"), "col_sweep": r["config"].get("col_sweep")},
            "row_sweep_ms": kn.get("row_sweep_ms"), "col_sweep_ms": kn.get("col_sweep_ms"),
            "mean_trials": {"per_row": kn.get("mean_trials_per_row"), "per_col": kn.get("mean_trials_per_col")},
            "roofline": {kk: rf.get(kk) for kk in ("bound", "kernel", "achieved", "peak", "unit", "frac", "cache_served", "traffic", "traffic_frac", "per_launch",
                                                   "avg_launch_ms", "algorithmic_frac")} | {"l2_hit_rate": (rf.get("l2") or {}).get("hit_rate")},
            "objective": r.get("objective"), "setup_s": r.get("setup_s"), "child_wall_s": time.time() - t0,
            "command": "bench.py " + " ".join(cmd[2:])}


def create_from_host_leg(args, cfg, api, m, n, k, q, device):
    """The boundary at north-star scale from HOST memory (`setup_s.create` times a device-to-device hand-over of generator
    output).  What a host binding pays before the first iteration when Omega lives in host memory as a sparse matrix:
      csr_from_csc_s  the row view from the CSC arrays by ONE counting transpose on the host (scipy's csc -> csr; the column view IS
                      colptr / rowval / nzval after an index shift) -- what a host without GLRM_PROBLEM_ROWS_FROM_COLS has to do
      create_s        glrm_hip_create on pageable host arrays: both lists' trip over PCIe + set-up (family choice, buffers)
      column_view_only.create_s   GLRM_PROBLEM_ROWS_FROM_COLS: the column view alone crosses PCIe, the row view is derived on the device
                      (what julia/HipGLRM.jl hands over for a SparseMatrixCSC)
    The lists are generated on the device and copied to the host first (d2h_s: not part of what a host would pay)."""
    import numpy as np
    import scipy.sparse as sp
    import torch
    from lowrankmodels.jl_amd import _capi, synth
    reg = cfg["reg"]
    t0 = time.time()
    w = synth.DeviceWorkload(m, n, k, q, seed=args.seed, value_model=cfg["value_model"], loss_mix=cfg["loss_mix"], rx=reg, ry=reg, device=device)
    colptr = w.colptr.cpu().numpy().astype(np.int64, copy=False)
    rowidx = w.rowidx[: w.nnz_cols].cpu().numpy()
    colvals = w.colvals[: w.nnz_cols].cpu().numpy()
    ref_rowptr = w.rowptr.cpu().numpy()
    ref_colidx_head = w.colidx[: 1 << 20].cpu().numpy()
    w.free_sources()
    del w
    torch.cuda.empty_cache()
    t_d2h = time.time() - t0
    t0 = time.time()
    A = sp.csc_matrix((colvals, rowidx, colptr), shape=(m, n)).tocsr()   # counting transpose, O(nnz); columns ascending inside every row
    rowptr, colidx, rowvals = A.indptr.astype(np.int64, copy=False), A.indices.astype(np.int32, copy=False), A.data
    t_csr = time.time() - t0
    same = bool(np.array_equal(rowptr, ref_rowptr) and np.array_equal(colidx[: 1 << 20], ref_colidx_head))  # = the generator's own row view
    r = np.array([reg], dtype=_capi.REG_DTYPE)
    pa = _capi.ProblemArrays(m, n, k, rowptr, colidx, rowvals, colptr, rowidx, colvals, synth.loss_table(n, cfg["loss_mix"]), r, r)
    nbytes = int(rowptr.nbytes + colidx.nbytes + rowvals.nbytes + colptr.nbytes + rowidx.nbytes + colvals.nbytes)
    t0 = time.time()
    h = api.create(pa, device_id=device.index or 0)
    t_create = time.time() - t0
    st = api.kernel_stats(h)
    api.destroy(h)
    # GLRM_PROBLEM_ROWS_FROM_COLS: the column view alone crosses PCIe, the engine derives the row view on the device (one stable radix
    # sort of the column-major stream by row id) -- no host transpose, half the upload
    pc = _capi.ProblemArrays(m, n, k, None, None, None, colptr, rowidx, colvals, synth.loss_table(n, cfg["loss_mix"]), r, r, flags=_capi.PROBLEM_ROWS_FROM_COLS)
    t0 = time.time()
    h = api.create(pc, device_id=device.index or 0)
    t_cols = time.time() - t0
    st2 = api.kernel_stats(h)
    api.destroy(h)
    return {"observations": int(rowptr[-1]), "list_bytes": nbytes, "csr_from_csc_s": t_csr, "csr_equals_the_generators_row_view": same,
            "create_s": t_create, "create_GBps": nbytes / t_create / 1e9, "kernel_flags": st["tiled"], "d2h_s_not_a_host_cost": t_d2h,
            "column_view_only": {"create_s": t_cols, "uploaded_bytes": int(colptr.nbytes + rowidx.nbytes + colvals.nbytes), "same_kernel_families": st2["tiled"] == st["tiled"],
                                 "nnz_rows": int(st2["nnz_rows"]),
                                 "is": "GLRM_PROBLEM_ROWS_FROM_COLS: colptr / rowidx / colvals only; the row view is derived on the device -- what "
                                       "julia/HipGLRM.jl hands over for a SparseMatrixCSC (no host transpose at all)"},
            "is": "host-resident Omega as CSC arrays -> row view by one counting transpose on the host (scipy csc -> csr) -> glrm_hip_create from "
                  "pageable host memory (PCIe upload of both views + set-up); column_view_only = the same Omega handed over as its column view alone; "
                  "the PCIe-inclusive figures are never part of `value`"}


def cpu_full_leg(args):
    """bench.py --cpu-full: ONE warm-up + ONE timed outer iteration of the CPU oracle on the FULL lists of the config (C4: 1e9 observations,
    24 GB of lists + 5 GB of factors on the host, twice while the handle copies them) on every core this box grants -- the check of the
    composite cpu_baseline (two bounded samples) VERDICT r3 asked for.  The problem is generated on the GPU and copied to the host once."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import oracle as O
    cfg = dict(CONFIGS[args.config])
    if args.config == "C3":
        raise SystemExit("--cpu-full covers the list configs")
    if args.cols or args.obs_per_row or args.k:
        cfg.update(cols=args.cols or cfg["cols"], q=args.obs_per_row or cfg["q"], k=args.k or cfg["k"])
    k, q, n = cfg["k"], cfg["q"], cfg["cols"]
    m = args.rows or cfg["rows"]
    need = (m * q * 12 * 2) * 2 + (m + n) * k * 8 * 3
    try:
        avail = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1]) * 1024
    except Exception:
        avail = None
    if avail is not None and avail < 1.3 * need:
        print(json.dumps({"mode": "cpu-full", "skipped": f"host memory: {avail / 1e9:.0f} GB available, {need / 1e9:.0f} GB needed (x 1.3)"}), flush=True)
        return
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    t0 = time.time()
    pa, X0, Y0 = host_problem(args, cfg, m, n, k, q, device)
    t_gen = time.time() - t0
    cores = O.usable_cores()
    O.set_threads(cores)
    api = O.oracle_api()
    h = api.create(pa)
    del pa
    api.set_factors(h, X0, Y0)
    api.reset_stepsizes(h, 1.0)
    t = []
    for _ in range(2):  # warm-up, timed
        a = time.time(); api.step_x(h, 0.01)
        b = time.time(); api.step_y(h, 0.01)
        t.append((b - a, time.time() - b))
    st = api.kernel_stats(h)
    api.destroy(h)
    nnz = m * q
    (wx, wy), (tx, ty) = t
    print(json.dumps({"mode": "cpu-full", "config": args.config, "m": m, "n": n, "k": k, "observations": nnz, "cores": cores, "kind": "port",
                      "timed": "the SECOND outer iteration of the oracle on the full lists (the first is the warm-up: from the random start, rows take more trials)",
                      "seconds": {"warmup_x": wx, "warmup_y": wy, "x_halfstep": tx, "y_halfstep": ty},
                      "x_halfstep_updates_per_s": nnz / tx, "y_halfstep_updates_per_s": nnz / ty, "updates_per_s": 2.0 * nnz / (tx + ty),
                      "trials": {"x": st["trials_x"], "y": st["trials_y"]}, "generate_and_copy_s": t_gen,
                      "where": "this box, this run"}), flush=True)


# ----------------------------------------------------------------------------- main

