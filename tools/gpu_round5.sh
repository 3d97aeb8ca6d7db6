#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "== auto: $(tail -1 gpurun_out/pytest_gpu.log)"
grep -E "^FAILED" gpurun_out/pytest_gpu.log | head
timeout 600 python bench.py --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_x.json")); k=d["kernels"]
print("ms/step %.2f row %.2f col %.2f trials %.3f/%.3f obj %.8g conv %s" % (d["ms_per_step"], k["row_sweep_ms"], k["col_sweep_ms"], k["mean_trials_per_row"], k["mean_trials_per_col"], d["objective"]["after_warmup_and_steps"], d["to_reference_stop"]))
PY
rm -rf gpurun_out/pmct_*
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-24)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmct_$tag -o pmc -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-convergence-run > $R/gpurun_out/pmct_$tag.json 2> $R/gpurun_out/pmct_$tag.err); echo "pmc $tag exit $?"
done
find gpurun_out -name "*kernel_trace*" -size +8M -delete
