#!/bin/bash
# round 3, call 28: single-buffer LDS-tiled sweeps staged by LDS-DMA from all waves (-DGLRM_TILE_DMA_ALL=1) against the product's load / ds_write
# staging: parity of the variant, then C5-family and C2 A/B on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
V=$PWD/lowrankmodels.jl_amd/libglrm_hip_dmaall.so
GLRM_HIP_LIB_PATH=$V timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_tiled.py -m gpu -q -x > gpurun_out/r3_28_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_28_pytest.log; tail -4 gpurun_out/r3_28_pytest.log
Q5="--config C5 --rows 1000000 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 6 --warmup 2"
Q2="--config C2 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 10 --warmup 3"
for W in C5 C2; do
for L in libglrm_hip.so libglrm_hip_dmaall.so libglrm_hip.so libglrm_hip_dmaall.so; do
  if [ $W = C5 ]; then Q=$Q5; else Q=$Q2; fi
  timeout 400 python tests/perf/ab_lib.py $L $Q > gpurun_out/r3_28_tmp.json 2> gpurun_out/r3_28_tmp.err
  python - "$L" "$W" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r3_28_tmp.json").read().strip().splitlines()[-1]); k=d["kernels"]
tr,tc=k["mean_trials_per_row"],k["mean_trials_per_col"]
print(sys.argv[2], sys.argv[1], "ms/step %.2f row %.2f col %.2f trials %.3f %.3f obj %.12g" % (d["ms_per_step"],k["row_sweep_ms"],k["col_sweep_ms"],tr,tc,d["objective"]["after_warmup_and_steps"]))
PY
done; done 2>&1 | tee gpurun_out/r3_28_ab.txt
