#!/bin/bash
# rocprofv3 kernel stats of the C4 and C5-family single-GPU runs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
for C in "C5 1000000" "C4 10000000"; do
  set -- $C
  rm -rf gpurun_out/prof_$1
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$1 -o bench -- python $R/bench.py --config $1 --rows-per-gpu $2 --steps 5 --warmup 3 --no-cpu-baseline --no-convergence-run > $R/gpurun_out/prof_$1.json 2> $R/gpurun_out/prof_$1.err); echo "$1 exit $?"
  cat gpurun_out/prof_$1.json | cut -c1-400
  find gpurun_out/prof_$1 -name "*kernel_stats.csv" | head -1 | xargs head -8
done
find gpurun_out -name "*kernel_trace*" -size +8M -delete
