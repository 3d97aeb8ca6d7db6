#!/bin/bash
# Round 2 round-end session: smoke, the whole -m gpu suite, the default bench line (C4, with in-run PMC / J_ref / cpu_baseline), rocprofv3
# kernel stats of the same config, and the C2 / C3 / C5-family lines.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
TAG=${1:-r02final}
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke_$TAG.log
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 --timeout 900 > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "== pytest: $(tail -1 gpurun_out/pytest_gpu_$TAG.log)"
grep -E "FAILED|ERROR" gpurun_out/pytest_gpu_$TAG.log | head -20
timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench (default = C4) exit $?"; cut -c1-2500 gpurun_out/bench_$TAG.json; tail -2 gpurun_out/bench_$TAG.err
rm -rf gpurun_out/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o bench -- python $R/bench.py --steps 10 --no-cpu-baseline --no-jref --no-convergence-run --pmc off > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_$TAG.err); echo "prof exit $?"
head -8 gpurun_out/prof_$TAG/bench_kernel_stats.csv | cut -c1-200
timeout 600 python bench.py --config C2 > gpurun_out/bench_c2_$TAG.json 2> gpurun_out/bench_c2_$TAG.err; echo "bench C2 exit $?"; cut -c1-600 gpurun_out/bench_c2_$TAG.json
timeout 600 python bench.py --config C5 --rows 1000000 --no-jref > gpurun_out/bench_c5_$TAG.json 2> gpurun_out/bench_c5_$TAG.err; echo "bench C5-family exit $?"; cut -c1-600 gpurun_out/bench_c5_$TAG.json
timeout 900 python bench.py --config C3 --steps 10 > gpurun_out/bench_c3_$TAG.json 2> gpurun_out/bench_c3_$TAG.err; echo "bench C3 exit $?"; cut -c1-600 gpurun_out/bench_c3_$TAG.json; tail -2 gpurun_out/bench_c3_$TAG.err
timeout 900 python bench.py --config C3 --steps 10 --quad-gram > gpurun_out/bench_c3gram_$TAG.json 2> gpurun_out/bench_c3gram_$TAG.err; echo "bench C3 quad_gram exit $?"; cut -c1-600 gpurun_out/bench_c3gram_$TAG.json; tail -2 gpurun_out/bench_c3gram_$TAG.err
find gpurun_out -name "*kernel_trace*" -size +8M -delete
