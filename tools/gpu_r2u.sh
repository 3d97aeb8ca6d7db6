#!/bin/bash
# round 2, session u: half-lane tiled layouts (GLRM_HIP_TILE_HALF=1: two lanes per segment at kp = 32) -- parity, then timing
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
GLRM_HIP_TILE_HALF=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x --timeout 900 > gpurun_out/pytest_u.log 2>&1; echo "== pytest (half): $(tail -1 gpurun_out/pytest_u.log)"
grep -E "FAILED|ERROR" gpurun_out/pytest_u.log | head
for cfg in C2 C5; do
  for half in 0 1; do
    echo "== $cfg half=$half"
    GLRM_HIP_TILE_HALF=$half timeout 600 python bench.py --config $cfg --rows 1000000 --steps 10 --warmup 3 --no-jref --no-cpu-baseline --no-convergence-run --pmc off 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']; print(d['ms_per_step'], k['row_sweep_ms'], k['col_sweep_ms'], k.get('mean_trials_per_row'), k.get('mean_trials_per_col'), d['objective'])"
  done
done 2>&1 | tee gpurun_out/ab_half.txt
