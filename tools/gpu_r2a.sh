#!/bin/bash
# Round 2, session A: new parity tests, the C4 default bench line (with in-run PMC), rocprofv3 kernel stats + TCC/SQ counters of the
# C4 gather sweeps, and the gather microbenchmark (HBM vs Infinity Cache vs per-XCD L2 windows).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
TAG=${1:-r02a}
timeout 120 build/ubench_gather > gpurun_out/ubench_gather_$TAG.txt 2>&1; echo "ubench exit $?"; cat gpurun_out/ubench_gather_$TAG.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -m gpu -q -k "c4 or ranks" --timeout 600 > gpurun_out/pytest_c4_$TAG.log 2>&1; echo "== pytest c4: $(tail -1 gpurun_out/pytest_c4_$TAG.log)"
timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench exit $?"; cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
rm -rf gpurun_out/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o bench -- python $R/bench.py --steps 10 --no-cpu-baseline --no-jref --no-convergence-run --pmc off > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_$TAG.err); echo "prof exit $?"
head -8 gpurun_out/prof_$TAG/bench_kernel_stats.csv
for C in "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-24)
  rm -rf gpurun_out/pmc_$tag
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$tag -o pmc -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-jref --no-convergence-run --pmc off > $R/gpurun_out/pmc_$tag.json 2> $R/gpurun_out/pmc_$tag.err); echo "pmc $tag exit $?"
done
python tools/pmc_summary.py gpurun_out > gpurun_out/pmc_summary_$TAG.md; cat gpurun_out/pmc_summary_$TAG.md
find gpurun_out -name "*kernel_trace*" -size +8M -delete
find gpurun_out -name "*counter_collection.csv" -size +8M -delete
