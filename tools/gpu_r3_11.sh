#!/bin/bash
# round 3, call 11: knobs of the phase-aligned passes at C4 (super-tile size, launch slice) with the persistent row sweep in place
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
Q="--steps 8 --warmup 2 --pmc off --no-jref --no-cpu-baseline --no-convergence-run"
for V in "910 25" "910 33" "910 66" "455 50" "455 33" "228 50" "455 25"; do
  set -- $V
  GLRM_HIP_BLOCKED_TPS=$1 GLRM_HIP_BLOCKED_FILL=$2 timeout 300 python bench.py $Q > gpurun_out/r3_11_tmp.json 2> gpurun_out/r3_11_tmp.err
  python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r3_11_tmp.json").read().strip().splitlines()[-1]); k=d["kernels"]
print("tps", sys.argv[1], "fill", sys.argv[2], "ms/step %.1f row %.2f col %.2f obj %.12g" % (d["ms_per_step"],k["row_sweep_ms"],k["col_sweep_ms"],d["objective"]["after_warmup_and_steps"]))
PY
done 2>&1 | tee gpurun_out/r3_11_blocked_knobs2.txt
