#!/bin/bash
# round 2, session w: dense pass with double-buffered staged tiles -- parity, then C3 A/B against the previous build
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_dense.py -m gpu -q --timeout 900 > gpurun_out/pytest_w.log 2>&1; echo "== pytest: $(tail -1 gpurun_out/pytest_w.log)"
grep -E "FAILED|ERROR" gpurun_out/pytest_w.log | head
for lib in libglrm_hip.so libglrm_hip_prev.so; do
 for flag in "" "--quad-gram"; do
  echo "== C3 $lib $flag"
  timeout 900 python tests/perf/ab_lib.py $lib --config C3 --steps 10 --warmup 3 --no-jref --no-cpu-baseline --no-convergence-run --pmc off $flag 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']; print(d['ms_per_step'], k['row_sweep_ms'], k['col_sweep_ms'], d['roofline']['frac'], d['objective'])"
 done
done 2>&1 | tee gpurun_out/c3_dbuf.txt
