#!/bin/bash
# GPU session: tiled sweeps -- parity under forced/auto modes, then bench per mode.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "3 0" "3 1"; do
  set -- $cfg
  GLRM_HIP_TILED=$1 GLRM_HIP_TILE_CFG=$2 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=6 --timeout 600 > gpurun_out/pytest_tiled_$1_$2.log 2>&1
  echo "== tiled=$1 cfg=$2: $(tail -1 gpurun_out/pytest_tiled_$1_$2.log)"
done
timeout 600 python -m pytest tests -m gpu -q --maxfail=6 --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "== auto: $(tail -1 gpurun_out/pytest_gpu.log)"
for cfg in "0 0" "3 0" "3 1"; do
  set -- $cfg
  GLRM_HIP_TILED=$1 GLRM_HIP_TILE_CFG=$2 timeout 600 python bench.py --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/bench_t$1_c$2.json 2> gpurun_out/bench_t$1_c$2.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_t$1_c$2.json")); k=d["kernels"]
    print("tiled=$1 cfg=$2 ms/step %.2f row %.2f col %.2f trials %.3f/%.3f obj %.8g" % (d["ms_per_step"], k["row_sweep_ms"], k["col_sweep_ms"], k["mean_trials_per_row"], k["mean_trials_per_col"], d["objective"]["after_warmup_and_steps"]))
except Exception as e:
    print("tiled=$1 cfg=$2 FAILED", e); print(open("gpurun_out/bench_t$1_c$2.err").read()[-600:])
PY
done
R=$PWD
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-24)
  rm -rf gpurun_out/pmct_$tag
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmct_$tag -o pmc -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmct_$tag.json 2> $R/gpurun_out/pmct_$tag.err); echo "pmc $tag exit $?"
done
find gpurun_out -name "*kernel_trace*" -size +8M -delete
