#!/bin/bash
# round 3, call 16: super-tile size of the phase-aligned passes with one-wave workgroups and half-residency slices
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
Q="--steps 8 --warmup 2 --pmc off --no-jref --no-cpu-baseline --no-convergence-run"
for V in "455 50" "228 50" "1820 50" "455 66" "114 50"; do
  set -- $V
  GLRM_HIP_BLOCKED_TPS=$1 GLRM_HIP_BLOCKED_FILL=$2 timeout 300 python bench.py $Q > gpurun_out/r3_16_tmp.json 2> gpurun_out/r3_16_tmp.err
  python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r3_16_tmp.json").read().strip().splitlines()[-1]); k=d["kernels"]
print("1 wave per workgroup, tps", sys.argv[1], "fill", sys.argv[2], "ms/step", round(d["ms_per_step"],1), "row", round(k["row_sweep_ms"],2), "col", round(k["col_sweep_ms"],2))
PY
done 2>&1 | tee gpurun_out/r3_16_tps.txt
