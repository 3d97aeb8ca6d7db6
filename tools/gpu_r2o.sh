#!/bin/bash
# Loader waves on the row sweep only (GLRM_HIP_TILE_LW=2, sides=1): parity incl. heterogeneous models, A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
GLRM_HIP_TILE_LW=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_reference_scripts.py tests/test_gpu_crossval.py tests/test_multi_in_process.py -m gpu -q --timeout 600 > gpurun_out/pytest_lw2rows.log 2>&1; echo "== pytest LW=2 rows: $(tail -1 gpurun_out/pytest_lw2rows.log)"
grep -E "FAILED|ERROR" gpurun_out/pytest_lw2rows.log | head -5
QS="--rows 1000000 --cols 50000 --obs-per-row 1000 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 10"
QC="--pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 10"
for LW in 0 2; do
  GLRM_HIP_TILE_LW=$LW timeout 300 python bench.py --config C2 $QC > gpurun_out/lwr${LW}_c2.json 2> gpurun_out/lwr${LW}_c2.err; echo "c2 LW=$LW exit $?"
  GLRM_HIP_TILE_LW=$LW timeout 300 python bench.py --config C5 $QS > gpurun_out/lwr${LW}_mix.json 2> gpurun_out/lwr${LW}_mix.err; echo "mix LW=$LW exit $?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/lwr*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "ms/iter %.2f" % d["ms_per_step"], "row %.2f col %.2f" % (d["kernels"]["row_sweep_ms"], d["kernels"]["col_sweep_ms"]), "trials %.3f %.3f" % (d["kernels"]["mean_trials_per_row"], d["kernels"]["mean_trials_per_col"]), "obj", d["objective"]["after_warmup_and_steps"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
