#!/usr/bin/env python
"""bench.py -- observed-entry updates/sec of the GLRM proximal-gradient hot path on MI355X.

A "step" is one outer iteration of fit!(glrm, ProxGradParams) (src/algorithms/proxgrad.jl:107-217): the X half-step over
every observed entry of the rank's rows, the Y half-step over every observed entry of its columns, the recorded objective.
One observed-entry update = one (i,j) consumed by one factor half-step, so an outer iteration performs
|Omega_rows| + |Omega_cols| updates.

Default workload = the configuration BASELINE.json's north star is quoted on (configs[3], "C4"): 10M x 100k, rank 64, QuadLoss,
1e9 observed entries, NonNegConstraint on X and Y, fp64, generated in HBM by the counter-based generator.  It fits one MI355X
(24 GB of index lists + 5.2 GB of factors).  --gpus N shards THAT problem (strong scaling: m/N rows and n/N columns per rank,
X and Y replicated, one all-gather of the updated factor after each half-step; --scaling weak keeps --rows rows per rank).
Other configs: --config C2 | C3 | C5.

    python bench.py                                  # C4 on one GPU
    python bench.py --gpus N                         # no launcher: spawns its own N ranks (torch.distributed.run on 127.0.0.1), one per GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W       # the driver's launcher form: used as it is

The JSON line carries
  roofline      the bound of the dominant kernel (the longest half-step): every candidate limiter of that kernel family is
                priced and the largest fraction is `frac`
  cpu_baseline  the CPU oracle (oracle/, test infrastructure) on the host cores, on a bounded sample of the same recipe
  to_ref_objective   iterations / seconds until the GPU's recorded objective is <= J_ref (1 + 1e-5), where J_ref is what the CPU
                oracle reaches with default ProxGradParams() and its own stop rule on a scaled-down problem of the same recipe
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench_legs import *  # noqa: E402,F401,F403  (config table, rooflines, CPU / PMC / host legs: bench_legs.py)
from bench_legs import _oracle_problem, _time_oracle, _pmc_note  # noqa: E402,F401  (underscore names the tests reach for)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C4", choices=sorted(CONFIGS))
    ap.add_argument("--rows", type=int, default=0, help="rows of the whole problem (strong scaling) or per rank (--scaling weak); 0 = the config's")
    ap.add_argument("--rows-per-gpu", type=int, default=0, help="alias: rows per rank with --scaling weak")
    ap.add_argument("--cols", type=int, default=0, help="override the config's column count (scaled-down runs and tests)")
    ap.add_argument("--obs-per-row", type=int, default=0, help="override the config's observations per row")
    ap.add_argument("--rank", dest="k", type=int, default=0, help="override the config's rank")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--seed", type=int, default=20260926)
    ap.add_argument("--waves-row", type=int, default=0)
    ap.add_argument("--waves-col", type=int, default=0)
    ap.add_argument("--tiled", type=int, default=0, help="0 auto, 1 gather sweeps only, 2 LDS-tiled sweeps")
    ap.add_argument("--quad-gram", action="store_true", help="C3 only: glrm_options.quad_gram (line-search trials from the quadratic form, no pass over A per trial)")
    ap.add_argument("--x-chunks", type=int, default=4, help="N > 1: row chunks of the X half-step whose all-gather overlaps the next chunk")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-convergence-run", action="store_true")
    ap.add_argument("--no-jref", action="store_true")
    ap.add_argument("--jref-small", action="store_true", help="to_ref_objective on the small in-run problem even when a fixture is committed")
    ap.add_argument("--pmc", default="auto", choices=["auto", "on", "off"], help="HBM traffic of the dominant kernel from rocprofv3 PMC child passes")
    ap.add_argument("--pmc-timeout", type=int, default=240)
    ap.add_argument("--cpu-sample-rows", type=int, default=20_000)
    ap.add_argument("--borrow", default="auto", choices=["auto", "on", "off"], help="hand the generated lists to the engine in place (no second copy in HBM)")
    ap.add_argument("--no-cpu-cols-sample", dest="cpu_cols_sample", action="store_false", help="cpu_baseline: skip the all-rows x few-columns sample")
    ap.add_argument("--emulate-rank", type=int, default=-1, help="with --of N: ONE GPU runs rank r's shard of the N-way sharded problem "
                    "(m/N rows, n/N columns, full replicas of X and Y, kernel families chosen from the whole problem) and times its half-steps")
    ap.add_argument("--of", type=int, default=0, help="number of shards emulated by --emulate-rank")
    ap.add_argument("--host", default="torch", choices=["torch", "inlib"], help="torch: one process per GPU under torch.distributed.run (RCCL); inlib: ONE "
                    "process drives --gpus N devices through glrm_hip_multi_create / glrm_hip_multi_fit (what the Julia shim ccalls)")
    ap.add_argument("--shared-device", action="store_true", help="--host inlib: every shard on device 0 (one-GPU box: the code path, not the speed)")
    ap.add_argument("--inlib-exchange", type=int, default=0, choices=[0, 1], help="--host inlib: 0 direct peer pushes, 1 RCCL")
    ap.add_argument("--arrival", type=int, default=0, choices=[0, 1, 2], help="--host inlib: glrm_multi_options.arrival (0 / 1: the Y half-step consumes "
                    "the X chunks in arrival order; 2: it waits for the whole exchange, the round-4 schedule)")
    ap.add_argument("--emulate-link-gbps", type=float, default=0.0, help="--host inlib: emulate an xGMI link of this many GB/s per direction on the "
                    "direct pushes (GLRM_EXCHANGE_EMULATE_GBPS) and run the same fit once more with free copies: the difference is the exposed exchange")
    ap.add_argument("--no-inlib-leg", action="store_true", help="N > 1 under torch.distributed.run: skip the in-library host's run that rank 0 adds to the line")
    ap.add_argument("--degree", default="uniform", choices=["uniform", "zipf"], help="uniform: exactly q observations per row (SURVEY 8(d) recipe); zipf: "
                    "power-law row degrees and column popularities with about the same |Omega| (lowrankmodels.jl_amd/synth.py: ZipfWorkload; N = 1, list configs)")
    ap.add_argument("--zipf-s", type=float, default=0.5, help="--degree zipf: the exponent of both laws, w(rank) = (rank + 1)^-s")
    ap.add_argument("--no-create-from-host", action="store_true", help="default C4 line: skip the create-from-host-arrays leg (24 GB over PCIe, ~1 minute)")
    ap.add_argument("--no-other-configs", action="store_true", help="default C4 line at N = 1: skip the compact C2 / C3 / C5 lines (child runs, ~5 minutes)")
    ap.add_argument("--launch-timeout", type=int, default=1500, help="--gpus N without a launcher: seconds the self-launched N-rank job may take before it is "
                    "killed and the in-library host (one process, N devices) runs instead")
    ap.add_argument("--cpu-full", action="store_true", help="one warm-up + one timed iteration of the CPU oracle on the FULL lists of the config (minutes, tens of GB of host memory)")
    args = ap.parse_args()
    if args.emulate_rank >= 0:
        return emulate_rank(args)
    if args.cpu_full:
        return cpu_full_leg(args)
    if args.host == "inlib":
        return inlib_host(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        return self_launch(args, sys.argv[1:])  # plain `python bench.py --gpus N`: no launcher around us -- start the ranks ourselves

    import torch
    import torch.distributed as dist
    from lowrankmodels.jl_amd import _capi, synth
    from lowrankmodels.jl_amd.fit import ShardedFit

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # GLRM_BENCH_BACKEND=gloo is a plumbing check of the multi-rank path on a box with fewer GPUs than ranks (ranks then share
    # devices); the measured configuration is one rank per GPU over RCCL.
    backend = os.environ.get("GLRM_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    ranks_seen = ranks_seen_block(dist if world > 1 else None, torch, rank, dev_index)  # what the process group saw: ranks, backend, a device per rank

    cfg = dict(CONFIGS[args.config])
    if args.cols or args.obs_per_row or args.k:
        cfg.update(cols=args.cols or cfg["cols"], q=args.obs_per_row or cfg["q"], k=args.k or cfg["k"])
        cfg["text"] = "scaled-down " + cfg["text"].replace("5% observed", "{pct:.3g}% observed").replace("0.1% observed", "{pct:.3g}% observed").replace("2% observed", "{pct:.3g}% observed")
    k, q, n, reg = cfg["k"], cfg["q"], cfg["cols"], cfg["reg"]
    if args.rows_per_gpu:
        args.rows, args.scaling = args.rows_per_gpu, "weak"
    if not args.rows:
        args.rows = cfg["rows"]
    m = args.rows * world if args.scaling == "weak" else args.rows
    if args.config == "C3" and world != 1:
        raise SystemExit("--config C3 runs on one GPU (the dense hand-over takes the whole matrix)")
    if n % world or n % q or m % world:
        raise SystemExit("rows and cols must be divisible by the number of GPUs, cols by the observations per row")
    rbs = [m // world * r for r in range(world + 1)]
    cbs = [n // world * r for r in range(world + 1)]

    api = _capi.hip_api()
    t_gen = time.time()
    zipf = args.degree == "zipf"
    if zipf:
        if world != 1 or args.config == "C3" or cfg["loss_mix"]:
            raise SystemExit("--degree zipf covers the single-loss list configs (C2, C4) on one GPU")
        w = synth.ZipfWorkload(m, n, k, m * q, s_rows=args.zipf_s, s_cols=args.zipf_s, seed=args.seed, value_model=cfg["value_model"], rx=reg, ry=reg, device=device)
        args.no_cpu_baseline = args.no_jref = args.no_other_configs = True  # those legs are defined on the uniform recipe
        cfg["text"] = ("POWER-LAW Omega (Zipf s = %g on row degrees and column popularities) of " % args.zipf_s) + cfg["text"]
    elif args.config == "C3":
        w = synth.DenseDeviceWorkload(m, n, k, seed=args.seed, rx=reg, ry=reg, device=device)
    else:
        w = synth.DeviceWorkload(m, n, k, q, rows=(rbs[rank], rbs[rank + 1]), cols=(cbs[rank], cbs[rank + 1]), seed=args.seed,
                                 value_model=cfg["value_model"], loss_mix=cfg["loss_mix"], rx=reg, ry=reg, device=device)
    t_gen = time.time() - t_gen
    t_create = time.time()
    # Lists too large to hold twice in HBM (C5 at its stated size: 120 GB) are handed over in place (GLRM_PROBLEM_BORROW_DEVICE_ARRAYS):
    # the engine reads the generator's arrays and only makes the private copies its kernels need (the row view regrouped by loss kind)
    borrow = args.config != "C3" and (args.borrow == "on" or (args.borrow == "auto" and w.list_bytes() > 90e9))
    sf = ShardedFit(api, w.problem(borrow=borrow) if args.config != "C3" else w.problem(), rbs, cbs, device=device, stream=torch.cuda.current_stream().cuda_stream,
                    opts=dict(profile=1, waves_row=args.waves_row, waves_col=args.waves_col, tiled=args.tiled, quad_gram=1 if args.quad_gram else 0),
                    x_chunks=args.x_chunks if args.config != "C3" else 1)
    nnz_r, nnz_c = w.nnz_rows, w.nnz_cols
    degrees = w.degree_summary() if zipf else None
    if not borrow:
        w.free_sources()
    t_create = time.time() - t_create
    def load_start():
        X0, Y0 = w.init_factors(sf.ld)
        if nonneg_start(cfg):
            X0.abs_().mul_(1.0 / k ** 0.5); Y0.abs_().mul_(1.0 / k ** 0.5)
        sf.dX.copy_(X0); sf.dY.copy_(Y0)
        del X0, Y0
        api.reset_stepsizes(sf.h, 1.0)

    load_start()

    class P:  # reference defaults, stop rule disabled (fixed number of outer iterations)
        stepsize, inner_iter_X, inner_iter_Y, min_stepsize = 1.0, 1, 1, 0.01

    obj0 = sf.initial_objective()
    objs = []
    for _ in range(args.warmup):
        objs.append(sf.iteration(P))
    api.kernel_stats(sf.h, reset=True)
    if world > 1:
        sf.exchange_times()  # drop the warm-up's exchange time

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        objs.append(sf.iteration(P))
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([nnz_r, nnz_c], dtype=torch.int64, device=device)
        dist.all_reduce(tot)
        tot_r, tot_c = int(tot[0].item()), int(tot[1].item())
    else:
        tot_r, tot_c = nnz_r, nnz_c
    st = api.kernel_stats(sf.h)
    # what the exchanges added to the iterations of the timed region (events on the rank's stream around every exchange; with row
    # chunks: the wait for the chunks the pipeline did not hide), and the set-up probe that picked the exchange (N > 2 on RCCL)
    exch = sf.exchange_times() if world > 1 else None
    if exch is not None:  # arrival order: the X exchange is waited for block by block inside the Y half-step (glrm_hip_step_y_arrival)
        exch["x"] += st.get("ms_wait_y", 0.0)

    # The reference's own run at full size -- default ProxGradParams(), stop rule of src/algorithms/proxgrad.jl:210-213, from the same
    # X0, Y0 -- timed end to end on the GPU (the CPU-derived J_ref leg is `to_ref_objective`).
    conv = None
    if not args.no_convergence_run:
        load_start()
        fence()
        tc = time.perf_counter()
        hist = [sf.initial_objective()]
        scaled_abs_tol = 1e-5 * float(tot_r)
        for i in range(1, 101):
            obj = sf.iteration(P)
            dec = hist[-1] - obj
            hist.append(obj)
            if i > 10 and (dec < scaled_abs_tol or dec / obj < 1e-4):
                break
        fence()
        conv = {"iterations": len(hist) - 1, "seconds": time.perf_counter() - tc, "objective_initial": hist[0],
                "objective_final": hist[-1], "stop_rule": "reference defaults: abs_tol=1e-5*|Omega|, rel_tol=1e-4, max_iter=100"}

    nseg_r, nseg_c = rbs[1] - rbs[0], cbs[1] - cbs[0]
    flags = st["tiled"]
    sf_p2p, sf_chunks, sf_probe = sf._p2p, sf.x_chunks, sf.exchange_probe_ms
    sf.close()
    del sf
    if borrow:
        w.free_sources()  # the handle is gone: the borrowed lists may go too (the PMC child runs need the memory)
    torch.cuda.empty_cache()

    if rank == 0:
        updates_per_step = tot_r + tot_c
        value = args.steps * updates_per_step / elapsed
        ms_x = st["ms_x"] / max(args.steps, 1)  # per outer iteration (the X half-step may run as several chunk launches)
        ms_y = st["ms_y"] / max(args.steps, 1)
        fam_r = "general" if flags & 8 else "dense" if flags & 4 else "lane" if flags & 256 else "tiled" if flags & 1 else "cached" if flags & 64 else "blocked" if flags & 16 else "gather"
        fam_c = "general" if flags & 8 else "dense" if flags & 4 else "lane" if flags & 512 else "tiled" if flags & 2 else "blocked" if flags & 32 else "gather"
        ld = st["ld"]
        tile = 150 * 1024 // (ld * 8 + 16) // 16 * 16  # csrc/glrm_tiled.hip: tile_rows_c
        rl_r = kernel_roofline(fam_r, nnz=nnz_r, nseg=nseg_r, nopp=n, k=k, ld=ld, ms=ms_x, m=nseg_r, n=n, tile=tile, quad_gram=args.quad_gram)
        rl_c = kernel_roofline(fam_c, nnz=nnz_c, nseg=nseg_c, nopp=m, k=k, ld=ld, ms=ms_y, m=m, n=nseg_c, tile=tile, quad_gram=args.quad_gram)
        # dominant kernel = the longer half-step
        dom = "row" if ms_x >= ms_y else "col"
        rl, dom_ms, dom_nnz, dom_fam = (rl_r, ms_x, nnz_r, fam_r) if dom == "row" else (rl_c, ms_y, nnz_c, fam_c)
        kname = {("row", "gather"): ("sweep_kernel (X half-step, k-vector gather)", "sweep_kernel"),
                 ("col", "gather"): ("sweep_kernel (Y half-step, k-vector gather)", "sweep_kernel"),
                 ("row", "tiled"): ("tiled_sweep_kernel<ROUNDS> (gradient pass + first trial) + tiled_col_pass_kernel<ROWS> rounds over the still-searching rows "
                                    "(X half-step, LDS-tiled)", r"tiled_sweep_kernel|tiled_col_pass_kernel<[^>]*, true>"),
                 ("col", "tiled"): ("tiled_col_pass_kernel gradient pass + trial rounds + col_reduce/col_decide (Y half-step, LDS-tiled)",
                                    r"tiled_col_pass_kernel<[^>]*, false>"),
                 ("row", "lane"): ("lane_pass_kernel gradient pass + trial rounds + col_reduce/col_decide (X half-step, LDS tiles, one lane per row)", "lane_pass_kernel"),
                 ("col", "lane"): ("lane_pass_kernel gradient pass + trial rounds + col_reduce/col_decide (Y half-step, LDS tiles, one lane per column)", "lane_pass_kernel"),
                 ("row", "blocked"): ("tiled_col_pass_kernel<L2> passes + col_reduce/col_decide (X half-step, phase-aligned L2 gathers)", "tiled_col_pass_kernel"),
                 ("col", "blocked"): ("tiled_col_pass_kernel<L2> passes + col_reduce/col_decide (Y half-step, phase-aligned L2 gathers)", "tiled_col_pass_kernel"),
                 ("row", "cached"): ("regcached_persist_kernel / regcached_sweep_kernel<G, R, LOSS, 7, 2> (X half-step: the row's list and opposing vectors fetched "
                                     "once into registers, two waves per row; rows beyond 13 trips run sweep_kernel)", r"regcached_(persist|sweep)_kernel"),
                 ("row", "dense"): ("dense_pass_kernel (X half-step, fp64 MFMA)", "dense_pass_kernel"),
                 ("col", "dense"): ("dense_pass_kernel (Y half-step, fp64 MFMA)", "dense_pass_kernel"),
                 ("row", "general"): ("multi_sweep_kernel (X half-step)", "multi_sweep_kernel"),
                 ("col", "general"): ("multi_colpass_kernel (Y half-step)", "multi_colpass_kernel")}[(dom, dom_fam)]
        traffic, traffic_src, l2_hits = None, "not collected", None
        if world == 1 and args.pmc != "off":
            try:
                kre = kname[1]
                if dom_fam == "gather":  # row and column sweeps are instantiations of one template: tell them apart by WAVES
                    wv = st["waves_row"] if dom == "row" else st["waves_col"]
                    kre = r"(?<!tiled_)sweep_kernel<\d+, \d+, %d, \d+, \d+, false>" % wv
                # the row and the column pass of the blocked family are the same kernel instantiation: the child runs only the dominant
                # side on it (the other side on the one-kernel gather sweep), so its dispatches can be told apart by name
                cenv = {"GLRM_HIP_BLOCKED": "1" if dom == "row" else "2"} if dom_fam == "blocked" else None
                if dom_fam == "lane":  # rows and columns run ONE kernel: the child keeps only the dominant side on it (the other on the four-lane kernels)
                    cenv = {"GLRM_HIP_LANE": "1" if dom == "row" else "2"}
                traffic, traffic_src, l2_hits = pmc_traffic(args, kre, per_halfstep=dom_fam in ("blocked", "tiled", "lane"), child_env=cenv,
                                                            eval_pass=dom == "col" and dom_fam in ("blocked", "tiled", "lane"))
                if traffic is not None and dom_fam == "gather" and st["waves_row"] == st["waves_col"]:
                    traffic_src += " (row and column sweeps run the same instantiation here: the mean is over both)"
            except Exception as e:  # the bench line must survive a profiler problem
                traffic, traffic_src = None, f"PMC pass failed: {e!r}"
        roof = roofline_block(rl, kname[0], dom_ms, dom_nnz, traffic, traffic_src, l2_hits)
        out = {
            "metric": "observed-entry updates/sec", "value": value, "unit": "updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic", "ranks_seen": ranks_seen,
            "config": {"workload": cfg["text"].format(m=m, n=n, k=k, pct=100.0 * q / n) + f" ({tot_r} observations), ProxGradParams defaults, stop rule off"
                                   + (", glrm_options.quad_gram = 1" if args.quad_gram else ""),
                       "name": args.config, "m": m, "n": n, "k": k, "observed": tot_r,
                       "full_size": m == CONFIGS[args.config]["rows"] and n == CONFIGS[args.config]["cols"] and k == CONFIGS[args.config]["k"], "start": INIT_NOTE[nonneg_start(cfg)], "regularizer": REG_NAME.get(reg[0], str(reg)),
                       "parallelism": f"rows/cols sharded over {world} GPU(s) ({args.scaling} scaling), X,Y replicated",
                       "waves_row": st["waves_row"], "waves_col": st["waves_col"], "row_sweep": fam_r, "col_sweep": fam_c,
                       **({"degree": "zipf", "zipf_s": args.zipf_s, "degrees": degrees} if zipf else {})},
            "roofline": roof,
            "kernels": {"row_sweep_ms": ms_x, "col_sweep_ms": ms_y,
                        "row_sweep": side_summary(rl_r), "col_sweep": side_summary(rl_c),
                        "mean_trials_per_row": st["trials_x"] / max(args.steps * nseg_r, 1),
                        "mean_trials_per_col": st["trials_y"] / max(args.steps * nseg_c, 1)},
            "exchange": None if exch is None else {
                "ms_per_step_on_rank0_stream": {kk: v / max(args.steps, 1) for kk, v in exch.items()},
                "mode": "p2p" if sf_p2p else "all-gather / broadcast", "x_chunks": sf_chunks, "probe_ms": sf_probe,
                "model_ms": {"X_block": exchange_model_ms((m // world) * st["ld"] * 8, world), "Y_block": exchange_model_ms((n // world) * st["ld"] * 8, world)},
                "note": "measured on rank 0's stream (HIP events around the exchanges; warm-up iterations are included in the sum only if "
                        "they ran after the last read-out); model: direct = one block per xGMI link at 76.8 GB/s in one direction (AMD's 153.6 GB/s per "
                        "link counts both directions), ring = N - 1 hops; unmeasured priors"},
            "step_model": step_model(fam_r, fam_c, nnz_r, nnz_c, nseg_r, nseg_c, k, ld, 1e3 * elapsed / args.steps, world, m=m, n=n) if args.config != "C3" else None,
            "objective": {"initial": obj0, "after_warmup_and_steps": objs[-1] if objs else None},
            "to_reference_stop": conv,
            "setup_s": {"generate": t_gen, "create": t_create, "lists_borrowed_in_place": bool(borrow)},
        }
        if world == 1 and not args.no_jref and args.config != "C3":
            try:
                scaled = bool(args.cols or args.obs_per_row or args.k)
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                from jref_tools import load_jref_fixture
                out["to_ref_objective"] = jref_leg(args, cfg, api, device, fixture=None if scaled or args.jref_small else load_jref_fixture(args.config, args.seed, CONFIGS[args.config]))
            except Exception as e:
                out["to_ref_objective"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            c3 = args.config == "C3"  # the oracle has no dense hand-over: its list path on the same fully observed recipe
            out["cpu_baseline"] = cpu_baseline(args, cfg, k, q if not c3 else n, n, m)
        # the same problem once more through the host the reference would bind (one process, N devices); every rank of this job has
        # released its shard, and waits at the barrier below while rank 0's child runs.  GLRM_BENCH_INLIB=shared: all shards on device 0
        # (the gloo plumbing mode of a box with fewer GPUs than ranks: tests/test_gpu_multirank.py drives this very code path)
        # the other BASELINE configs under the same clock (the default line is C4 at full size on one GPU; scaled-down / sharded runs skip them)
        full = m == CONFIGS[args.config]["rows"] and n == CONFIGS[args.config]["cols"] and k == CONFIGS[args.config]["k"]
        if world == 1 and args.config == "C4" and full and not args.no_other_configs and args.tiled == 0:
            out["other_configs"] = {"C2": other_config_line(args, "C2"), "C3": other_config_line(args, "C3"),
                                    "C3_quad_gram": other_config_line(args, "C3", extra=("--quad-gram",)),
                                    "C5": other_config_line(args, "C5", timeout_s=900),
                                    "note": "each a child run of this script (`command`), this box, this session; C5 at its stated size (5e9 observations, lists "
                                            "borrowed in place); roofline as in the main line: frac = achieved / peak of the best-priced limiter, traffic = PMC"}
        if world == 1 and args.config == "C4" and full and not args.no_create_from_host and not zipf:
            try:
                out["setup_s"]["create_from_host"] = create_from_host_leg(args, cfg, api, m, n, k, q, device)
            except Exception as e:  # the line must survive (host memory, scipy)
                out["setup_s"]["create_from_host"] = {"error": repr(e)}
        inlib_shared = os.environ.get("GLRM_BENCH_INLIB") == "shared"
        if world > 1 and (backend == "nccl" or inlib_shared) and not args.no_inlib_leg and args.config != "C3" and args.scaling == "strong":
            out["inlib_host"] = inlib_child(args, world, shared_device=inlib_shared)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
