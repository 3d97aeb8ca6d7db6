#!/bin/bash
# Phase-aligned L2 gather passes against the LDS-tiled sweeps where the opposing factor is a few L2s large (1M x 50k: Y = 12.8 MB) and at C2.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
Q="--rows 1000000 --cols 50000 --obs-per-row 1000 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 10"
run() { # name, env..., -- args
  local name=$1; shift
  env "$@" timeout 600 python bench.py $ARGS > gpurun_out/blk_$name.json 2> gpurun_out/blk_$name.err; echo "$name exit $?"
}
ARGS="--config C2 $Q"; run quad_tiled GLRM_HIP_BLOCKED=0
ARGS="--config C2 $Q"; run quad_rowsblk GLRM_HIP_TILED=2 GLRM_HIP_BLOCKED=1
ARGS="--config C2 $Q"; run quad_bothblk GLRM_HIP_TILED=0 GLRM_HIP_BLOCKED=3
ARGS="--config C5 $Q"; run mix_rowsblk GLRM_HIP_TILED=2 GLRM_HIP_BLOCKED=1
ARGS="--config C5 $Q"; run mix_bothblk GLRM_HIP_TILED=0 GLRM_HIP_BLOCKED=3
ARGS="--config C2 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 10"; run c2_rowsblk GLRM_HIP_TILED=2 GLRM_HIP_BLOCKED=1
ARGS="--config C2 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 10"; run c2_bothblk GLRM_HIP_TILED=0 GLRM_HIP_BLOCKED=3
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/blk_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "ms/iter %.2f" % d["ms_per_step"], "row %.2f col %.2f" % (d["kernels"]["row_sweep_ms"], d["kernels"]["col_sweep_ms"]), d["config"]["row_sweep"], d["config"]["col_sweep"], "trials %.3f %.3f" % (d["kernels"]["mean_trials_per_row"], d["kernels"]["mean_trials_per_col"]))
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
