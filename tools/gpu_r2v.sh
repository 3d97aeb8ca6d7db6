#!/bin/bash
# round 2, session v: glrm_options.quad_gram on the dense path -- parity tests, then the C3 line with and without it
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_dense.py tests/test_multi_in_process.py tests/test_c_abi_example.py -m gpu -q --timeout 900 > gpurun_out/pytest_v.log 2>&1; echo "== pytest: $(tail -1 gpurun_out/pytest_v.log)"
grep -E "FAILED|ERROR" gpurun_out/pytest_v.log | head
for flag in "" "--quad-gram"; do
  echo "== C3 $flag"
  timeout 900 python bench.py --config C3 --steps 10 --warmup 3 --no-jref --no-cpu-baseline --pmc off $flag 2>&1 | tail -1 > gpurun_out/c3_gram_${flag#--}.json
  python -c "
import json,sys
d=json.load(open(sys.argv[1])); k=d['kernels']; print(d['ms_per_step'], k['row_sweep_ms'], k['col_sweep_ms'], d['roofline']['bound'], d['roofline']['frac'], d['objective'], d.get('to_reference_stop'))" gpurun_out/c3_gram_${flag#--}.json
done 2>&1 | tee gpurun_out/c3_gram.txt
