#!/bin/bash
# round 3, call 34: lockstep windows on the column view (persistent kernel, per-XCD meeting after every ~2 MB window of X): parity, then C4 A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_families.py -m gpu -q -k "lockstep" > gpurun_out/r3_34_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_34_pytest.log; tail -5 gpurun_out/r3_34_pytest.log
Q="--steps 6 --warmup 2 --pmc off --no-jref --no-cpu-baseline --no-convergence-run"
run() { local label=$1; shift
  timeout 400 python bench.py $Q > gpurun_out/r3_34_tmp.json 2> gpurun_out/r3_34_tmp.err
  python - "$label" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r3_34_tmp.json").read().strip().splitlines()[-1]); k=d["kernels"]
print(sys.argv[1], "ms/step %.1f row %.2f col %.2f trials %.3f %.3f obj %.12g" % (d["ms_per_step"],k["row_sweep_ms"],k["col_sweep_ms"],k["mean_trials_per_row"],k["mean_trials_per_col"],d["objective"]["after_warmup_and_steps"]))
PY
}
{
GLRM_HIP_LOCKSTEP=0 run "C4 phase-aligned launches      "
GLRM_HIP_LOCKSTEP=1 run "C4 lockstep, 2 MB windows      "
GLRM_HIP_LOCKSTEP=1 GLRM_HIP_LOCKSTEP_WT=28 run "C4 lockstep, 4 MB windows      "
GLRM_HIP_LOCKSTEP=1 GLRM_HIP_LOCKSTEP_SPIN=1 run "C4 lockstep, 2 MB, no meeting  "
} 2>&1 | tee gpurun_out/r3_34_ab.txt
