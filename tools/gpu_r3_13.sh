#!/bin/bash
# round 3, call 13: waves per workgroup of the phase-aligned passes (2 / 4 / 8) x launch slice
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
Q="--steps 8 --warmup 2 --pmc off --no-jref --no-cpu-baseline --no-convergence-run"
for V in "libglrm_hip_bnw1.so 25" "libglrm_hip_bnw1.so 33" "libglrm_hip_bnw1.so 40" "libglrm_hip_bnw1.so 50" "libglrm_hip_bnw1.so 60" "libglrm_hip_bnw1.so 75" "libglrm_hip_bnw2.so 40"; do
  set -- $V
  GLRM_HIP_BLOCKED_FILL=$2 timeout 300 python tests/perf/ab_lib.py $1 $Q > gpurun_out/r3_13_tmp.json 2> gpurun_out/r3_13_tmp.err
  python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r3_13_tmp.json").read().strip().splitlines()[-1]); k=d["kernels"]
print(sys.argv[1], "fill", sys.argv[2], "ms/step %.1f row %.2f col %.2f obj %.12g" % (d["ms_per_step"],k["row_sweep_ms"],k["col_sweep_ms"],d["objective"]["after_warmup_and_steps"]))
PY
done 2>&1 | tee gpurun_out/r3_13_blocked_nw3.txt
