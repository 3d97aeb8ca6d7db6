#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dense.py -m gpu -q --maxfail=20 --timeout 600 > gpurun_out/pytest_dense.log 2>&1; echo "== dense: $(tail -1 gpurun_out/pytest_dense.log)"
grep -E "^E  |^FAILED" gpurun_out/pytest_dense.log | head -10
timeout 900 python bench.py --config C3 --steps 4 --warmup 2 --no-convergence-run > gpurun_out/cfg_C3.json 2> gpurun_out/cfg_C3.err
python - <<PY
import json
d=json.load(open("gpurun_out/cfg_C3.json")); k=d["kernels"]
print("C3: %.3g upd/s ms/step %.2f row %.2f col %.2f trials %.3f/%.3f obj %.8g" % (d["value"], d["ms_per_step"], k["row_sweep_ms"], k["col_sweep_ms"], k["mean_trials_per_row"], k["mean_trials_per_col"], d["objective"]["after_warmup_and_steps"]))
PY
