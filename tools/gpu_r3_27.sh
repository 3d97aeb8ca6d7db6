#!/bin/bash
# round 3, call 27: where do the heterogeneous LDS-tiled sweeps spend their time?  timing builds on the C5-family shape (1M x 50k, 1e9 observations):
# product / staging compiled out after the first tile (GLRM_EXP_NOSTAGE) / compute compiled out (GLRM_EXP_NOCOMPUTE); times per PASS (trial counts differ)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
Q="--config C5 --rows 1000000 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 4 --warmup 1"
for L in libglrm_hip.so libglrm_hip_nostage.so libglrm_hip_nocompute.so; do
  timeout 400 python tests/perf/ab_lib.py $L $Q > gpurun_out/r3_27_$L.json 2> gpurun_out/r3_27_$L.err
  python - "$L" <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r3_27_{sys.argv[1]}.json").read().strip().splitlines()[-1]); k=d["kernels"]
tr,tc=k["mean_trials_per_row"],k["mean_trials_per_col"]
print(sys.argv[1], "ms/step %.1f row %.2f col %.2f trials %.3f %.3f -> per pass: row %.2f col %.2f" % (d["ms_per_step"],k["row_sweep_ms"],k["col_sweep_ms"],tr,tc,k["row_sweep_ms"]/(1+tr),k["col_sweep_ms"]/(1+tc)))
PY
done 2>&1 | tee gpurun_out/r3_27_timing.txt
