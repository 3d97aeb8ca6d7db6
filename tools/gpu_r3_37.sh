#!/bin/bash
# round 3, call 37: persistent cached row sweep with non-temporal loads of the observation lists (-DGLRM_CACHED_NT) against the product, C4
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
Q="--steps 6 --warmup 2 --pmc off --no-jref --no-cpu-baseline --no-convergence-run"
for L in libglrm_hip.so libglrm_hip_cachednt.so libglrm_hip.so libglrm_hip_cachednt.so; do
  timeout 400 python tests/perf/ab_lib.py $L $Q > gpurun_out/r3_37_tmp.json 2> gpurun_out/r3_37_tmp.err
  python - "$L" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r3_37_tmp.json").read().strip().splitlines()[-1]); k=d["kernels"]
print("C4", sys.argv[1], "ms/step %.1f row %.2f col %.2f obj %.12g" % (d["ms_per_step"],k["row_sweep_ms"],k["col_sweep_ms"],d["objective"]["after_warmup_and_steps"]))
PY
done 2>&1 | tee gpurun_out/r3_37_nt.txt
