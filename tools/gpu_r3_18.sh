#!/bin/bash
# round 3, call 18: persistent cached row sweep on four waves per row (experiment) against the default two
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
Q="--steps 8 --warmup 2 --pmc off --no-jref --no-cpu-baseline --no-convergence-run"
for W in 2 4; do
  GLRM_HIP_CACHED_WAVES=$W timeout 300 python bench.py $Q > gpurun_out/r3_18_tmp.json 2> gpurun_out/r3_18_tmp.err
  python - "$W" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r3_18_tmp.json").read().strip().splitlines()[-1]); k=d["kernels"]
print("persistent cached sweep,", sys.argv[1], "waves per row: ms/step", round(d["ms_per_step"],1), "row", round(k["row_sweep_ms"],2), "col", round(k["col_sweep_ms"],2), "obj", repr(d["objective"]["after_warmup_and_steps"]))
PY
done 2>&1 | tee gpurun_out/r3_18_waves.txt
