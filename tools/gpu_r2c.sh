#!/bin/bash
# Round 2, session C: heterogeneous tiled row sweep with one loss kind per wave step -- parity tests, then same-box A/B on the C5-family
# line (libglrm_hip_base.so = the build before the change) and SQ counters of the new kernel.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
TAG=${1:-r02c}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_reference_scripts.py tests/test_gpu_crossval.py -m gpu -q -x --timeout 600 > gpurun_out/pytest_$TAG.log 2>&1; echo "== pytest: $(tail -1 gpurun_out/pytest_$TAG.log)"
grep -E "FAILED|ERROR|assert" gpurun_out/pytest_$TAG.log | head -20
Q="--config C5 --rows 1000000 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 10"
timeout 600 python tests/perf/ab_lib.py libglrm_hip_base.so $Q > gpurun_out/c5_base_$TAG.json 2> gpurun_out/c5_base_$TAG.err; echo "base exit $?"
timeout 600 python bench.py $Q > gpurun_out/c5_new_$TAG.json 2> gpurun_out/c5_new_$TAG.err; echo "new exit $?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c5_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "ms/iter %.2f" % d["ms_per_step"], "row %.2f col %.2f" % (d["kernels"]["row_sweep_ms"], d["kernels"]["col_sweep_ms"]), "trials/row %.3f" % d["kernels"]["mean_trials_per_row"], "obj", d["objective"]["after_warmup_and_steps"])
    except Exception as e: print(f, "ERR", e)
PY
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES" "WRITE_SIZE"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-24)
  rm -rf gpurun_out/pmc_$tag
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$tag -o pmc -- python $R/bench.py --config C5 --rows 1000000 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 2 --warmup 2 > $R/gpurun_out/pmc_$tag.json 2> $R/gpurun_out/pmc_$tag.err); echo "pmc $tag exit $?"
done
python tools/pmc_summary.py gpurun_out > gpurun_out/pmc_summary_$TAG.md; grep "tiled_sweep" gpurun_out/pmc_summary_$TAG.md
find gpurun_out -name "*kernel_trace*" -size +8M -delete; find gpurun_out -name "*counter_collection.csv" -size +8M -delete
