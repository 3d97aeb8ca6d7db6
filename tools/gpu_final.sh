#!/bin/bash
# Round-end check on the GPU box: smoke, all -m gpu tests, the default bench line, rocprofv3 kernel stats of the same command,
# and the C5-family line on both kernel families.  (PMC passes: tools/gpu_full.sh.)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
TAG=${1:-final}
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -2 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 --timeout 900 > gpurun_out/pytest_gpu.log 2>&1; echo "== pytest: $(tail -1 gpurun_out/pytest_gpu.log)"
timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench exit $?"; cat gpurun_out/bench_$TAG.json
rm -rf gpurun_out/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_$TAG.err); echo "prof exit $?"
head -8 gpurun_out/prof_$TAG/bench_kernel_stats.csv
for T in 0 1; do
  timeout 600 python bench.py --config C5 --rows-per-gpu 1000000 --tiled $T --no-cpu-baseline --no-convergence-run > gpurun_out/c5_tiled$T.json 2>/dev/null
  python -c "import json; d=json.loads(open('gpurun_out/c5_tiled$T.json').readlines()[-1]); print('C5 tiled=$T', round(d['ms_per_step'],2), d['kernels'], d['value'])"
done
rm -rf gpurun_out/prof_C5
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_C5 -o bench -- python $R/bench.py --config C5 --rows-per-gpu 1000000 --steps 5 --warmup 3 --no-cpu-baseline --no-convergence-run > $R/gpurun_out/prof_C5.json 2> $R/gpurun_out/prof_C5.err); echo "C5 prof exit $?"
head -6 gpurun_out/prof_C5/bench_kernel_stats.csv
find gpurun_out -name "*kernel_trace*" -size +8M -delete
