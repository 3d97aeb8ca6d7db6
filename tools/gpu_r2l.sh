#!/bin/bash
# Timing-only experiments on the LDS-tiled sweeps: what does a tile cost without staging (NOSTAGE) and without compute (NOCOMPUTE)?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
QS="--config C2 --rows 1000000 --cols 50000 --obs-per-row 1000 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 6 --warmup 2"
QC="--config C2 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 6 --warmup 2"
for L in libglrm_hip.so libglrm_hip_NOSTAGE.so libglrm_hip_NOCOMPUTE.so; do
  timeout 300 python tests/perf/ab_lib.py $L $QC > gpurun_out/exp_c2_$L.json 2> gpurun_out/exp_c2_$L.err; echo "c2 $L exit $?"
  timeout 300 python tests/perf/ab_lib.py $L $QS > gpurun_out/exp_s_$L.json 2> gpurun_out/exp_s_$L.err; echo "sparse $L exit $?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/exp_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "row %.2f col %.2f" % (d["kernels"]["row_sweep_ms"], d["kernels"]["col_sweep_ms"]), "trials %.3f %.3f" % (d["kernels"]["mean_trials_per_row"], d["kernels"]["mean_trials_per_col"]))
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
