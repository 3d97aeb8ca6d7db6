#!/bin/bash
# round 2, session 3d: whole -m gpu suite and the C2 / default lines with the loader waves on by default for uniform QuadLoss row sweeps
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 --timeout 900 > gpurun_out/pytest_gpu_r02s3.log 2>&1; echo "== pytest: $(tail -1 gpurun_out/pytest_gpu_r02s3.log)"
grep -E "FAILED|ERROR" gpurun_out/pytest_gpu_r02s3.log | head -20
timeout 600 python bench.py --config C2 > gpurun_out/bench_c2_r02s3.json 2> gpurun_out/bench_c2_r02s3.err; echo "bench C2 exit $?"; cut -c1-400 gpurun_out/bench_c2_r02s3.json
python -c "
import json
d=json.load(open('gpurun_out/bench_c2_r02s3.json')); k=d['kernels']; print(d['ms_per_step'], k['row_sweep_ms'], k['col_sweep_ms'], d['roofline']['bound'], d['roofline']['frac'], d['to_reference_stop'], d['to_ref_objective']['gpu_first_iteration_at_or_below_J_ref'])"
