#!/bin/bash
# GPU session 2: re-run -m gpu tests, variant grid, PMC passes (FETCH/WRITE/TCC hit) and csv kernel stats.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 --timeout 900 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log
timeout 1500 python tools/exp_variants.py gpurun_out/variants.jsonl; cat gpurun_out/variants.jsonl
(cd /tmp && rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1)
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rm -rf gpurun_out/pmc_$tag
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$tag -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$tag.json 2> $R/gpurun_out/pmc_$tag.err); echo "pmc $tag exit $?"
done
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err); echo "prof exit $?"
find gpurun_out -name "*.csv" | head -30
find gpurun_out -name "*kernel_trace*" -size +8M -delete
