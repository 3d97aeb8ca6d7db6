#!/bin/bash
# Full GPU session: smoke, all -m gpu tests, default bench, rocprofv3 kernel stats (csv) and PMC passes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
TAG=${1:-run}
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -2 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 --timeout 900 > gpurun_out/pytest_gpu.log 2>&1; echo "== pytest: $(tail -1 gpurun_out/pytest_gpu.log)"
timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench exit $?"; cat gpurun_out/bench_$TAG.json
rm -rf gpurun_out/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_$TAG.err); echo "prof exit $?"
head -12 gpurun_out/prof_$TAG/bench_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-24)
  rm -rf gpurun_out/pmc_$tag
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$tag -o pmc -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_$tag.json 2> $R/gpurun_out/pmc_$tag.err); echo "pmc $tag exit $?"
done
find gpurun_out -name "*kernel_trace*" -size +8M -delete
