"""-m gpu: the multi-rank path of bench.py with the HIP engine on ONE GPU: two ranks (gloo instead of RCCL, both on cuda:0) run the
sharded fit -- row / column shard handles, pipelined X exchange on a side stream, all-gather of Y and of the per-column
objectives -- and record the same objective, bit for bit, as a single rank on the same problem."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run(cmd, env):
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


QUIET = ["--no-convergence-run", "--no-cpu-baseline", "--no-jref", "--pmc", "off"]


def torchrun(n):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(free_port()), "bench.py", "--gpus", str(n)]


@pytest.mark.parametrize("x_chunks,gather", [(4, "allgather"), (1, "broadcast"), (1, "p2p")])
def test_two_ranks_on_one_gpu_equal_one_rank(x_chunks, gather):
    """Weak-scaling flags (--rows-per-gpu): two ranks x 30 000 rows = one rank x 60 000 rows of the C2 recipe."""
    common = ["--config", "C2", "--steps", "3", "--warmup", "2", "--cols", "2000", "--obs-per-row", "100", "--tiled", "2"] + QUIET
    env = dict(os.environ, GLRM_BENCH_BACKEND="gloo", GLRM_GATHER=gather)
    two = run(torchrun(2) + ["--rows-per-gpu", "30000", "--x-chunks", str(x_chunks)] + common, env)
    one = run([sys.executable, "bench.py", "--rows", "60000"] + common, dict(os.environ))
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1 and two["scaling"] == "weak"
    assert two["config"]["observed"] == one["config"]["observed"] == 60000 * 100
    assert two["objective"]["initial"] == one["objective"]["initial"]
    assert two["objective"]["after_warmup_and_steps"] == one["objective"]["after_warmup_and_steps"]


@pytest.mark.parametrize("nranks", [2, 8])
def test_c4_recipe_strong_scaling_equals_one_rank(nranks):
    """bench.py's default mode: the C4 recipe (rank 64, NonNegConstraint, non-negative start) as ONE fixed problem whose rows and
    columns are cut into `nranks` shards (strong scaling) -- same recorded objectives, bit for bit, as the single-rank run."""
    common = ["--config", "C4", "--rows", "48000", "--cols", "4000", "--obs-per-row", "100", "--steps", "3", "--warmup", "2"] + QUIET
    env = dict(os.environ, GLRM_BENCH_BACKEND="gloo", GLRM_GATHER="allgather", GLRM_BENCH_INLIB="shared")
    many = run(torchrun(nranks) + common, env)
    one = run([sys.executable, "bench.py"] + common, dict(os.environ))
    assert many["n_gpus"] == nranks and many["scaling"] == "strong" and many["config"]["k"] == 64
    # rank 0 ran the in-library host (glrm_hip_multi_fit: what the Julia shim ccalls) on the same problem as a child process and
    # embedded its line: same shard count, same objective bits as both other hosts
    lib = many["inlib_host"]
    assert "error" not in lib, lib
    assert lib["n_gpus"] == nranks and lib["config"]["host"] == "inlib" and lib["host"]["shared_device"]
    assert lib["objective"]["after_warmup_and_steps"] == one["objective"]["after_warmup_and_steps"]
    assert many["config"]["observed"] == one["config"]["observed"] == 48000 * 100
    assert many["objective"] == one["objective"]
    rf = one["roofline"]  # one definition: achieved / peak of the best-priced limiter; above 0.9 of the HBM spec it is flagged cache_served
    assert rf["frac"] > 0 and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"]) and rf["bound"] in ("hbm", "l2", "lds", "mfma", "infinity_cache")
    assert rf["cache_served"] == (rf["bound"] == "hbm" and rf["frac"] > 0.9) and "traffic_frac" in rf
    # the multi-rank line says what the exchanges cost on rank 0's stream and what the xGMI model expects; the single-rank line has none
    ex = many["exchange"]
    assert one["exchange"] is None and set(ex["ms_per_step_on_rank0_stream"]) == {"x", "y", "objective"}
    assert all(v >= 0 for v in ex["ms_per_step_on_rank0_stream"].values()) and ex["model_ms"]["X_block"]["direct"] > 0
    assert one["step_model"]["within_peak"] and many["step_model"]["passes"] == one["step_model"]["passes"]


def test_eight_ranks_on_one_gpu_equal_one_rank():
    """The world size the scaling bench ends at: eight row / column shards (default auto kernel choice, pipelined X exchange)."""
    common = ["--config", "C2", "--steps", "2", "--warmup", "2", "--cols", "4000", "--obs-per-row", "200"] + QUIET
    env = dict(os.environ, GLRM_BENCH_BACKEND="gloo", GLRM_GATHER="allgather")
    eight = run(torchrun(8) + ["--rows-per-gpu", "25000"] + common, env)
    one = run([sys.executable, "bench.py", "--rows", "200000"] + common, dict(os.environ))
    assert eight["n_gpus"] == 8 and eight["config"]["observed"] == one["config"]["observed"] == 200000 * 200
    assert eight["objective"] == one["objective"]


def test_phase_aligned_passes_two_ranks_equal_one_rank():
    """The C4 recipe at 2M rows (2e8 observations): large enough for the automatic choices of the full-size problem -- the row sweep
    with the row's vectors held in registers (csrc/glrm_cached.hip), the phase-aligned gather passes on the column side
    (csrc/glrm_blocked.hip) -- made from the GLOBAL problem, so two shards run the same families as one and record the same objectives
    bit for bit (strong scaling, pipelined X exchange in row chunks on the cached row sweep)."""
    common = ["--config", "C4", "--rows", "2000000", "--steps", "2", "--warmup", "2"] + QUIET
    env = dict(os.environ, GLRM_BENCH_BACKEND="gloo", GLRM_GATHER="allgather")
    two = run(torchrun(2) + common, env)
    one = run([sys.executable, "bench.py"] + common, dict(os.environ))
    for r in (one, two):
        assert r["config"]["row_sweep"] == "cached" and r["config"]["col_sweep"] == "blocked"
    assert two["objective"] == one["objective"]
    # both sides on the phase-aligned passes (the round-2 default before the cached row sweep)
    env_b = dict(GLRM_HIP_CACHED="0")
    two = run(torchrun(2) + common, dict(env, **env_b))
    one = run([sys.executable, "bench.py"] + common, dict(os.environ, **env_b))
    for r in (one, two):
        assert r["config"]["row_sweep"] == r["config"]["col_sweep"] == "blocked"
    assert two["objective"] == one["objective"]


def test_plain_command_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus N` -- the driver's plain command, no torch.distributed.run around it -- spawns its own N
    ranks (here gloo: both ranks share the box's one GPU), prints ONE line with exit code 0, and the line says what the process group saw:
    `ranks_seen` (world size, backend, a device per rank) and `launch` (the command it re-ran itself under).  Same objective bits as one rank;
    the one-rank command is unchanged apart from carrying `ranks_seen` too."""
    common = ["--config", "C4", "--rows", "48000", "--cols", "4000", "--obs-per-row", "100", "--steps", "3", "--warmup", "2", "--no-inlib-leg"] + QUIET
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + common, env=dict(env, GLRM_BENCH_BACKEND="gloo", GLRM_GATHER="allgather"),
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    two = json.loads(lines[0])
    one = run([sys.executable, "bench.py"] + common, env)
    assert two["n_gpus"] == 2 and two["objective"] == one["objective"] and two["value"] > 0
    rs = two["ranks_seen"]
    assert rs["world_size"] == 2 and rs["backend"] == "gloo" and [d["rank"] for d in rs["devices"]] == [0, 1]
    assert rs["distinct_devices"] == 1  # one GPU on this box: the plumbing, not the speed -- a real run shows N distinct PCI addresses
    assert "torch.distributed.run" in two["launch"]["command"] and two["launch"]["visible_devices"] >= 1
    assert one["ranks_seen"]["world_size"] == 1 and one["ranks_seen"]["backend"] is None and "launch" not in one


def test_plain_command_falls_back_and_says_so():
    """The RCCL backend needs a device per rank: on a one-GPU box `bench.py --gpus 2` cannot run its ranks and says so in ONE line (value null,
    `error`, exit code 1) instead of a traceback; with shared devices allowed (gloo) and the N-rank job failing to start -- an unusable
    rendezvous address -- the in-library host (one process, N devices) runs the same problem and its line is printed with `fell_back_from`."""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("written for the one-GPU box")
    common = ["--config", "C4", "--rows", "48000", "--cols", "4000", "--obs-per-row", "100", "--steps", "2", "--warmup", "1"] + QUIET
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GLRM_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + common, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert r.returncode == 1 and line["value"] is None and "device" in line["error"]["error"] and line["n_gpus"] == 2
    # the N-rank job dies at once (the backend name is not one torch.distributed knows): the in-library host takes over
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + common, env=dict(env, GLRM_BENCH_BACKEND="no-such-backend"), capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["config"]["host"] == "inlib" and line["n_gpus"] == 2 and line["value"] > 0 and "fell_back_from" in line
    assert line["ranks_seen"]["world_size"] == 2 and line["host"]["shared_device"]
