#!/bin/bash
# round 3, call 29: LDS-DMA staging from all waves as the default (padded and ROT tiles): parity suites of the tiled sweeps, soak seeds on the
# tiled family, then C5-family / C2 A/B against -DGLRM_TILE_DMA_ALL=0 (libglrm_hip_stagevgpr.so) and C2 with the loader waves off
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_families.py tests/test_gpu_crossval.py tests/test_reference_scripts.py tests/test_reference_notebook.py -m gpu -q > gpurun_out/r3_29_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_29_pytest.log; tail -4 gpurun_out/r3_29_pytest.log
timeout 300 python tests/perf/soak_fuzz.py 3000 3600 2>&1 | grep -v Warning | grep -v "return float" | tail -14
Q5="--config C5 --rows 1000000 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 6 --warmup 2"
Q2="--config C2 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 10 --warmup 3"
run() { # label lib args...
  local label=$1 lib=$2; shift 2
  timeout 400 python tests/perf/ab_lib.py $lib "$@" > gpurun_out/r3_29_tmp.json 2> gpurun_out/r3_29_tmp.err
  python - "$label" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r3_29_tmp.json").read().strip().splitlines()[-1]); k=d["kernels"]
print(sys.argv[1], "ms/step %.2f row %.2f col %.2f trials %.3f %.3f obj %.12g" % (d["ms_per_step"],k["row_sweep_ms"],k["col_sweep_ms"],k["mean_trials_per_row"],k["mean_trials_per_col"],d["objective"]["after_warmup_and_steps"]))
PY
}
{
run "C5-family product(dma)  " libglrm_hip.so $Q5
run "C5-family stagevgpr     " libglrm_hip_stagevgpr.so $Q5
run "C5-family product(dma)  " libglrm_hip.so $Q5
run "C2 product(dma)         " libglrm_hip.so $Q2
run "C2 stagevgpr            " libglrm_hip_stagevgpr.so $Q2
GLRM_HIP_TILE_LW=0 run "C2 product(dma), LW=0   " libglrm_hip.so $Q2
GLRM_HIP_TILE_LW=1 run "C2 product(dma), LW=1   " libglrm_hip.so $Q2
run "C2 product(dma)         " libglrm_hip.so $Q2
} 2>&1 | tee gpurun_out/r3_29_ab.txt
