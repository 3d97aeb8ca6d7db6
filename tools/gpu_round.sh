#!/bin/bash
# One GPU-box session: smoke, -m gpu tests, bench, rocprofv3 kernel trace.  Everything is logged under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
STAGE=${1:-all}
echo "== rocminfo ==" > gpurun_out/env.log
(rocminfo | grep -E "Marketing Name|gfx9" | head -4; nproc; free -g | head -2) >> gpurun_out/env.log 2>&1
if [[ "$STAGE" == "all" || "$STAGE" == *smoke* ]]; then
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
  tail -3 gpurun_out/smoke.log
fi
if [[ "$STAGE" == "all" || "$STAGE" == *test* ]]; then
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 --timeout 900 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
  tail -15 gpurun_out/pytest_gpu.log
fi
if [[ "$STAGE" == "all" || "$STAGE" == *bench* ]]; then
  timeout 900 python bench.py --steps ${BENCH_STEPS:-10} --warmup ${BENCH_WARMUP:-3} > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"
  tail -2 gpurun_out/bench.err; cat gpurun_out/bench.json
fi
if [[ "$STAGE" == "all" || "$STAGE" == *prof* ]]; then
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof_bench.json" 2> "$OLDPWD/gpurun_out/prof.err"); echo "prof exit $?"
  find gpurun_out/prof -name "*stats*" | head; find gpurun_out/prof -name "*kernel_stats*" -exec head -12 {} \;
  # keep the big per-dispatch trace out of the merge-back budget
  find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete
fi
