#!/bin/bash
# round 3, call 15: resident grid of the persistent row sweep (percent of what the occupancy query admits)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
Q="--steps 8 --warmup 2 --pmc off --no-jref --no-cpu-baseline --no-convergence-run"
for F in 100 75 50 150 125; do
  GLRM_HIP_CACHED_PERSIST_FILL=$F timeout 300 python bench.py $Q > gpurun_out/r3_15_tmp.json 2> gpurun_out/r3_15_tmp.err
  python - "$F" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r3_15_tmp.json").read().strip().splitlines()[-1]); k=d["kernels"]
print("persistent grid", sys.argv[1], "percent: ms/step", round(d["ms_per_step"],1), "row", round(k["row_sweep_ms"],2), "col", round(k["col_sweep_ms"],2), "obj", repr(d["objective"]["after_warmup_and_steps"]))
PY
done 2>&1 | tee gpurun_out/r3_15_persist_grid.txt
