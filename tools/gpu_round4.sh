#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
GLRM_HIP_TILED=3 GLRM_HIP_TILED_HALF_LANES=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=6 --timeout 600 > gpurun_out/pytest_half.log 2>&1; echo "== half lanes: $(tail -1 gpurun_out/pytest_half.log)"
for cfg in "1 0" "0 1" "1 1"; do
  set -- $cfg
  GLRM_HIP_TILE_CFG=$1 GLRM_HIP_TILED_HALF_LANES=$2 timeout 600 python bench.py --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_x.json")); k=d["kernels"]
    print("cfg=$1 half=$2 ms/step %.2f row %.2f col %.2f trials %.3f/%.3f obj %.8g" % (d["ms_per_step"], k["row_sweep_ms"], k["col_sweep_ms"], k["mean_trials_per_row"], k["mean_trials_per_col"], d["objective"]["after_warmup_and_steps"]))
except Exception as e:
    print("cfg=$1 half=$2 FAILED", e); print(open("gpurun_out/bench_x.err").read()[-600:])
PY
done
