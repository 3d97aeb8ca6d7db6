#!/bin/bash
# round 2, session s: conflict-free tile reads (unpadded rows, chunk walk i ^ p) -- parity tests, then A/B against the padded build
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py tests/test_reference_notebook.py tests/test_gpu_fuzz.py tests/test_multi_in_process.py -m gpu -q -x --timeout 900 > gpurun_out/pytest_s.log 2>&1; echo "== pytest: $(tail -1 gpurun_out/pytest_s.log)"
grep -E "FAILED|ERROR" gpurun_out/pytest_s.log | head
for cfg in C2 C5; do
  for lib in libglrm_hip.so libglrm_hip_norot.so; do
    echo "== $cfg $lib"
    timeout 600 python tests/perf/ab_lib.py $lib --config $cfg --rows 1000000 --steps 10 --warmup 3 --no-jref --no-cpu-baseline --no-convergence-run --pmc off 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']; print(d['ms_per_step'], k['row_sweep_ms'], k['col_sweep_ms'], k.get('mean_trials_per_row'), k.get('mean_trials_per_col'), d['objective'])"
  done
done 2>&1 | tee gpurun_out/ab_rot.txt
